/*
 * lf_reader.c -- the data formats either side of the hot path: FASTA / FASTQ (plain or gzip) in, SAM out.
 *
 * lf_reads_*    record grammar of the reference's reader (src/Reads.cpp:43-131 = kseq over gzFile): a record starts
 *               at '>' or '@'; the name ends at the first white space, the rest of the header line is a comment and
 *               is dropped; sequence lines are concatenated until a line starts with '>', '@' or '+'; after '+' the
 *               quality string is read until it is as long as the sequence.  FASTA records get QUAL "*" (:104-108).
 * lf_map_file   reads batches on a reader thread while the previous batch is on the GPU, writes the SAM header
 *               (src/BWT.cpp:668-681) and the records of every batch in input order: the `--search` loop of
 *               src/baseFAST.cpp:56-81 as one call.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <zlib.h>
#include <unistd.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <time.h>
#include "lf_internal.h"

struct lf_reads {
    gzFile fp;
    unsigned char *buf; int beg, end, eof;
    int last_char;                     /* header character already consumed ('>' / '@'), 0 if none */
    char path[1024];
    /* plain (uncompressed) FASTA in a regular file: the file is mapped and a batch is parsed by several threads at once
     * (lf_reads_next_mapped).  Anything else -- gzip, FASTQ, pipes -- goes through the sequential parser below. */
    const unsigned char *map; size_t map_size, map_pos; int mapped;
};
#define LF_RBUF (4 << 20)

static inline int rd_getc(struct lf_reads *r)
{
    if (r->beg >= r->end) {
        if (r->eof) return -1;
        r->beg = 0; r->end = gzread(r->fp, r->buf, LF_RBUF);
        if (r->end <= 0) { r->eof = 1; r->end = 0; return -1; }
    }
    return r->buf[r->beg++];
}

typedef struct { char *s; size_t n, cap; } gstr_t;
static inline void gs_putc(gstr_t *g, char c) { if (g->n + 2 > g->cap) { g->cap = g->cap ? g->cap * 2 : 256; g->s = (char *)realloc(g->s, g->cap); } g->s[g->n++] = c; }
static inline void gs_putn(gstr_t *g, const unsigned char *p, size_t n)
{
    if (g->n + n + 2 > g->cap) { while (g->n + n + 2 > g->cap) g->cap = g->cap ? g->cap * 2 : 256; g->s = (char *)realloc(g->s, g->cap); }
    memcpy(g->s + g->n, p, n); g->n += n;
}
/* append the rest of the current line (without the newline); returns the terminating character (-1 at EOF) */
static int gs_getline(struct lf_reads *r, gstr_t *g, int stop_at_space)
{
    for (;;) {
        /* bulk scan of the buffered bytes: sequence / quality lines are found with memchr and appended in one piece (a 15 kbp
         * read is one or a few hundred lines; byte-at-a-time parsing ran at ~0.3 GB/s, a hundredth of what the GPU maps) */
        if (!stop_at_space && r->beg < r->end) {
            const unsigned char *p0 = r->buf + r->beg;
            const unsigned char *nl = (const unsigned char *)memchr(p0, '\n', (size_t)(r->end - r->beg));
            const size_t len = nl ? (size_t)(nl - p0) : (size_t)(r->end - r->beg);
            gs_putn(g, p0, len);
            r->beg += (int)len;
            if (nl) { r->beg++; return '\n'; }
        }
        while (stop_at_space && r->beg < r->end) {
            const unsigned char c = r->buf[r->beg];
            if (c == '\n' || c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f') { r->beg++; return c; }
            r->beg++;
            gs_putc(g, (char)c);
        }
        const int c = rd_getc(r);
        if (c < 0) return -1;
        r->beg--;                                  /* re-scan it in the loop above */
    }
}

int lf_reads_open(const char *path, lf_reads_t **out)
{
    if (!path || !out) { lf_set_error("lf_reads_open: bad argument"); return LF_ERR_ARG; }
    gzFile fp = gzopen(path, "r");
    if (!fp) { lf_set_error("lf_reads_open: cannot open %s", path); return LF_ERR_IO; }
    (void)gzbuffer(fp, 1 << 20);
    struct lf_reads *r = (struct lf_reads *)calloc(1, sizeof *r);
    r->fp = fp; r->buf = (unsigned char *)malloc(LF_RBUF);
    snprintf(r->path, sizeof r->path, "%s", path);
    if (!getenv("LF_READER_SEQUENTIAL")) {
        /* a regular file whose first byte is '>' (not gzip's 0x1f, not FASTQ's '@'): map it */
        const int fd = open(path, O_RDONLY);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            void *m = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                if (((const unsigned char *)m)[0] == '>') { r->map = (const unsigned char *)m; r->map_size = (size_t)sb.st_size; r->map_pos = 0; r->mapped = 1; (void)madvise(m, (size_t)sb.st_size, MADV_SEQUENTIAL); }
                else munmap(m, (size_t)sb.st_size);
            }
        }
        if (fd >= 0) close(fd);
    }
    *out = r;
    return LF_OK;
}
static void blob_pool_trim(void);
void lf_reads_close(lf_reads_t *r)
{
    if (!r) return;
    blob_pool_trim();                           /* the blobs cached for this file's batches (batches freed later are cached again) */
    if (r->map) munmap((void *)r->map, r->map_size);
    gzclose(r->fp); free(r->buf); free(r);
}

/* one record into (name, seq, qual); returns 1 = record, 0 = end of file, < 0 = error (truncated quality) */
static int read_record(struct lf_reads *r, gstr_t *name, gstr_t *seq, gstr_t *qual)
{
    int c;
    name->n = seq->n = qual->n = 0;
    if (r->last_char == 0) {                       /* jump to the next header line */
        while ((c = rd_getc(r)) != -1 && c != '>' && c != '@') {}
        if (c == -1) return 0;
        r->last_char = c;
    }
    c = gs_getline(r, name, 1);
    if (c == -1 && name->n == 0) { r->last_char = 0; return 0; }
    if (c != '\n' && c != -1) { static __thread gstr_t skip; skip.n = 0; (void)gs_getline(r, &skip, 0); }      /* comment */
    while ((c = rd_getc(r)) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;                   /* empty line */
        gs_putc(seq, (char)c);
        (void)gs_getline(r, seq, 0);
        if (seq->n > 1 && seq->s[seq->n - 1] == '\r') seq->n--;          /* "\r\n" line ends (kseq.h:140) */
    }
    if (c == '>' || c == '@') r->last_char = c; else r->last_char = 0;
    gs_putc(name, 0); name->n--; gs_putc(seq, 0); seq->n--;
    if (c != '+') { gs_putc(qual, 0); qual->n--; return 1; }                 /* FASTA */
    { static __thread gstr_t skip; skip.n = 0; c = gs_getline(r, &skip, 0); }  /* rest of the '+' line */
    if (c == -1) return -2;
    while (qual->n < seq->n) {
        const size_t before = qual->n;
        c = gs_getline(r, qual, 0);
        if (qual->n > 1 && qual->s[qual->n - 1] == '\r') qual->n--;
        if (c == -1 && qual->n == before) break;
    }
    r->last_char = 0;
    gs_putc(qual, 0); qual->n--;
    if (qual->n != seq->n) return -2;
    return 1;
}

/* Blobs of the mapped-FASTA batches are recycled: a batch is ~0.75 GB in eight blobs, and memory fresh from malloc (= mmap
 * at that size) costs a page fault and a zeroed page per 4 KiB on first touch and an munmap on free -- more than the parse
 * itself (1 M reads: the reader took 3.0 s, the mapper 2.6 s).  Up to 32 freed blobs wait here for the next batch. */
#define LF_BLOB_POOL 32
static struct { char *p; size_t cap; } g_blob_pool[LF_BLOB_POOL];
static pthread_mutex_t g_blob_mu = PTHREAD_MUTEX_INITIALIZER;
static char *blob_get(size_t need, size_t *cap_out)
{
    pthread_mutex_lock(&g_blob_mu);
    int best = -1;
    for (int i = 0; i < LF_BLOB_POOL; i++) if (g_blob_pool[i].p && g_blob_pool[i].cap >= need && (best < 0 || g_blob_pool[i].cap < g_blob_pool[best].cap)) best = i;
    char *p = NULL; size_t cap = 0;
    if (best >= 0) { p = g_blob_pool[best].p; cap = g_blob_pool[best].cap; g_blob_pool[best].p = NULL; }
    pthread_mutex_unlock(&g_blob_mu);
    if (!p) { cap = need + need / 8 + 4096; p = (char *)malloc(cap); }      /* the pieces of later batches differ by a few per cent */
    *cap_out = cap;
    return p;
}
static void blob_pool_trim(void)
{
    pthread_mutex_lock(&g_blob_mu);
    for (int i = 0; i < LF_BLOB_POOL; i++) { free(g_blob_pool[i].p); g_blob_pool[i].p = NULL; g_blob_pool[i].cap = 0; }
    pthread_mutex_unlock(&g_blob_mu);
}
static void blob_put(char *p, size_t cap)
{
    if (!p) return;
    pthread_mutex_lock(&g_blob_mu);
    int slot = -1, smallest = -1;
    for (int i = 0; i < LF_BLOB_POOL; i++) {
        if (!g_blob_pool[i].p) { slot = i; break; }
        if (smallest < 0 || g_blob_pool[i].cap < g_blob_pool[smallest].cap) smallest = i;
    }
    char *drop = NULL;
    if (slot < 0 && smallest >= 0 && g_blob_pool[smallest].cap < cap) { drop = g_blob_pool[smallest].p; slot = smallest; }      /* keep the larger ones */
    if (slot >= 0) { g_blob_pool[slot].p = p; g_blob_pool[slot].cap = cap; p = NULL; }
    pthread_mutex_unlock(&g_blob_mu);
    free(drop); free(p);
}

struct lf_read_batch {
    int n; uint64_t bases;
    const char **names, **seqs, **quals; uint32_t *lens;
    char *blobs[8]; size_t blob_caps[8]; int nblobs;                    /* mapped FASTA: one blob per parser thread */
    char *blob; size_t blob_n, blob_cap;
    size_t *off; int cap;                          /* 3 offsets per record into blob */
};
void lf_read_batch_free(lf_read_batch_t *b)
{
    if (!b) return;
    for (int t = 0; t < b->nblobs; t++) blob_put(b->blobs[t], b->blob_caps[t]);
    free(b->names); free(b->seqs); free(b->quals); free(b->lens); free(b->blob); free(b->off); free(b);
}
int lf_read_batch_size(const lf_read_batch_t *b) { return b ? b->n : 0; }
const char *const *lf_read_batch_names(const lf_read_batch_t *b) { return b->names; }
const char *const *lf_read_batch_seqs(const lf_read_batch_t *b) { return b->seqs; }
const char *const *lf_read_batch_quals(const lf_read_batch_t *b) { return b->quals; }

static void batch_put(lf_read_batch_t *b, const char *s, size_t n, size_t *off)
{
    if (b->blob_n + n + 1 > b->blob_cap) { while (b->blob_n + n + 1 > b->blob_cap) b->blob_cap = b->blob_cap ? b->blob_cap * 2 : (1 << 20); b->blob = (char *)realloc(b->blob, b->blob_cap); }
    memcpy(b->blob + b->blob_n, s, n); b->blob[b->blob_n + n] = 0;
    *off = b->blob_n; b->blob_n += n + 1;
}

/* ---- mapped plain FASTA: a batch = a byte range of the file that ends at a record boundary, cut into one piece per thread
 * (again at record boundaries: a '>' at the start of a line); every thread copies its records' names and sequence lines
 * into a blob of its own.  Same grammar as read_record: the name ends at the first white space, sequence lines are
 * concatenated, "\r\n" line ends and empty lines are dropped.  A line that starts with '@' or '+' inside a record would
 * start a new record / a quality string in the reference's reader (kseq): such a file is handed to the sequential parser. ---- */
typedef struct {
    const unsigned char *p, *end; size_t file_off0;
    char *blob; size_t n, cap;
    size_t *off; size_t *rec_file_off; int nrec, caprec; uint64_t bases; int weird;
} mpiece_t;
static const unsigned char *next_record_start(const unsigned char *p, const unsigned char *end)
{   /* first '>' at the start of a line at or after p (p itself counts only if it is the start of the mapping or follows '\n') */
    while (p < end) {
        const unsigned char *nl = (const unsigned char *)memchr(p, '\n', (size_t)(end - p));
        if (!nl || nl + 1 >= end) return end;
        if (nl[1] == '>') return nl + 1;
        p = nl + 1;
    }
    return end;
}
static void *mpiece_main(void *arg)
{
    mpiece_t *M = (mpiece_t *)arg;
    const unsigned char *p = M->p, *end = M->end;
    M->blob = blob_get((size_t)(end - p) + 64, &M->cap); M->n = 0;
    while (p < end) {
        if (*p != '>') { M->weird = 1; return NULL; }
        if (M->nrec == M->caprec) { M->caprec = M->caprec ? M->caprec * 2 : 4096; M->off = (size_t *)realloc(M->off, (size_t)M->caprec * 2 * sizeof(size_t)); M->rec_file_off = (size_t *)realloc(M->rec_file_off, (size_t)M->caprec * sizeof(size_t)); }
        M->rec_file_off[M->nrec] = (size_t)(p - M->p) + M->file_off0;
        const unsigned char *q = p + 1;
        const unsigned char *nl = (const unsigned char *)memchr(q, '\n', (size_t)(end - q));
        const unsigned char *le = nl ? nl : end;
        const unsigned char *ne = q;
        while (ne < le && !(*ne == ' ' || *ne == '\t' || *ne == '\r' || *ne == '\v' || *ne == '\f')) ne++;
        M->off[2 * M->nrec] = M->n;
        memcpy(M->blob + M->n, q, (size_t)(ne - q)); M->n += (size_t)(ne - q); M->blob[M->n++] = 0;
        M->off[2 * M->nrec + 1] = M->n;
        p = nl ? nl + 1 : end;
        while (p < end && *p != '>') {
            if (*p == '@' || *p == '+') { M->weird = 1; return NULL; }
            nl = (const unsigned char *)memchr(p, '\n', (size_t)(end - p));
            le = nl ? nl : end;
            size_t len = (size_t)(le - p);
            if (len > 0) {                                    /* (an empty line adds nothing) */
                memcpy(M->blob + M->n, p, len); M->n += len;
                if (M->n - M->off[2 * M->nrec + 1] > 1 && M->blob[M->n - 1] == '\r') M->n--;      /* kseq.h:140 */
            }
            p = nl ? nl + 1 : end;
        }
        M->bases += M->n - M->off[2 * M->nrec + 1];
        M->blob[M->n++] = 0;
        M->nrec++;
    }
    return NULL;
}
static int lf_reads_next_mapped(lf_reads_t *r, int max_reads, uint64_t max_bases, lf_read_batch_t **out)
{
    *out = NULL;
    if (r->map_pos >= r->map_size) return LF_OK;
    const unsigned char *base = r->map, *fend = r->map + r->map_size;
    const unsigned char *a = base + r->map_pos;
    uint64_t want = max_bases + (max_bases >> 6) + 4096;                 /* bytes: bases + headers + newlines */
    if (want > (uint64_t)(fend - a)) want = (uint64_t)(fend - a);
    const unsigned char *e = (a + want >= fend) ? fend : next_record_start(a + want - 1, fend);
    int nt = (int)sysconf(_SC_NPROCESSORS_ONLN); if (nt > 8) nt = 8; if (nt < 1) nt = 1;
    if ((size_t)(e - a) < ((size_t)4 << 20)) nt = 1;
    mpiece_t M[8]; pthread_t th[8]; int started[8];
    memset(M, 0, sizeof M);
    const unsigned char *cut = a;
    for (int t = 0; t < nt; t++) {
        const unsigned char *nx = (t == nt - 1) ? e : next_record_start(a + (size_t)(e - a) / (size_t)nt * (size_t)(t + 1), e);
        M[t].p = cut; M[t].end = nx; M[t].file_off0 = (size_t)(cut - base);
        cut = nx;
    }
    for (int t = 0; t < nt; t++) { started[t] = (t > 0 && M[t].p < M[t].end) ? pthread_create(&th[t], NULL, mpiece_main, &M[t]) == 0 : 0; }
    mpiece_main(&M[0]);
    for (int t = 1; t < nt; t++) { if (started[t]) pthread_join(th[t], NULL); else if (M[t].p < M[t].end) mpiece_main(&M[t]); }
    int weird = 0, total = 0;
    for (int t = 0; t < nt; t++) { weird |= M[t].weird; total += M[t].nrec; }
    if (weird) {
        /* not a plain FASTA after all: the sequential parser takes over from the start of this batch */
        for (int t = 0; t < nt; t++) { blob_put(M[t].blob, M[t].cap); free(M[t].off); free(M[t].rec_file_off); }
        r->mapped = 0;
        if (gzseek(r->fp, (z_off_t)r->map_pos, SEEK_SET) < 0) { lf_set_error("lf_reads_next: cannot reposition %s", r->path); return LF_ERR_IO; }
        r->beg = r->end = 0; r->eof = 0; r->last_char = 0;
        return lf_reads_next(r, max_reads, max_bases, out);
    }
    /* max_reads / max_bases: keep the leading records that fit (at least one), continue at the first one that does not */
    lf_read_batch_t *b = (lf_read_batch_t *)calloc(1, sizeof *b);
    int keep = 0; uint64_t bases = 0; size_t next_pos = (size_t)(e - base);
    if (max_reads <= 0) max_reads = 1 << 30;
    b->names = (const char **)malloc(((size_t)total + 1) * sizeof(char *)); b->seqs = (const char **)malloc(((size_t)total + 1) * sizeof(char *)); b->quals = (const char **)malloc(((size_t)total + 1) * sizeof(char *));
    b->lens = (uint32_t *)malloc(((size_t)total + 1) * sizeof(uint32_t));
    int stop = 0;
    for (int t = 0; t < nt && !stop; t++) for (int k = 0; k < M[t].nrec; k++) {
        const size_t so = M[t].off[2 * k + 1], eo = (k + 1 < M[t].nrec) ? M[t].off[2 * k + 2] : M[t].n;
        const uint32_t len = (uint32_t)(eo - so - 1);
        if (keep > 0 && (keep >= max_reads || bases >= max_bases)) { next_pos = M[t].rec_file_off[k]; stop = 1; break; }
        b->names[keep] = M[t].blob + M[t].off[2 * k]; b->seqs[keep] = M[t].blob + so; b->quals[keep] = "";
        b->lens[keep] = len; bases += len; keep++;
    }
    b->n = keep; b->bases = bases;
    b->nblobs = nt; for (int t = 0; t < nt; t++) { b->blobs[t] = M[t].blob; b->blob_caps[t] = M[t].cap; free(M[t].off); free(M[t].rec_file_off); }
    r->map_pos = next_pos;
    if (keep == 0) { lf_read_batch_free(b); return LF_OK; }
    *out = b;
    return LF_OK;
}

/* next batch: up to max_reads records / max_bases sequence bytes; *out = NULL at end of file */
int lf_reads_next(lf_reads_t *r, int max_reads, uint64_t max_bases, lf_read_batch_t **out)
{
    *out = NULL;
    if (max_reads <= 0) max_reads = 1 << 30;
    if (max_bases == 0) max_bases = ~0ull;
    if (r->mapped) return lf_reads_next_mapped(r, max_reads, max_bases == ~0ull ? ((uint64_t)1 << 40) : max_bases, out);
    lf_read_batch_t *b = (lf_read_batch_t *)calloc(1, sizeof *b);
    gstr_t name = { 0, 0, 0 }, seq = { 0, 0, 0 }, qual = { 0, 0, 0 };
    int rc = LF_OK;
    while (b->n < max_reads && b->bases < max_bases) {
        const int k = read_record(r, &name, &seq, &qual);
        if (k == 0) break;
        if (k < 0) {      /* the reference's loop `while (kseq_read(ks) >= 0)` (src/Reads.cpp:76) stops reading here */
            fprintf(stderr, "[WARNING] (lf_reads_next) %s: quality string of record %s does not match its sequence; input ends here\n", r->path, name.s ? name.s : "?");
            r->eof = 1; r->beg = r->end = 0; r->last_char = 0;
            break;
        }
        if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 4096; b->off = (size_t *)realloc(b->off, (size_t)b->cap * 3 * sizeof(size_t)); }
        batch_put(b, name.s, name.n, &b->off[3 * b->n]);
        batch_put(b, seq.s, seq.n, &b->off[3 * b->n + 1]);
        if (qual.n) batch_put(b, qual.s, qual.n, &b->off[3 * b->n + 2]); else batch_put(b, "", 0, &b->off[3 * b->n + 2]);     /* "" = FASTA -> "*" */
        b->bases += seq.n; b->n++;
    }
    free(name.s); free(seq.s); free(qual.s);
    if (rc != LF_OK || b->n == 0) { lf_read_batch_free(b); return rc; }
    b->names = (const char **)malloc((size_t)b->n * sizeof(char *)); b->seqs = (const char **)malloc((size_t)b->n * sizeof(char *)); b->quals = (const char **)malloc((size_t)b->n * sizeof(char *));
    b->lens = (uint32_t *)malloc((size_t)b->n * sizeof(uint32_t));
    for (int i = 0; i < b->n; i++) {
        b->names[i] = b->blob + b->off[3 * i]; b->seqs[i] = b->blob + b->off[3 * i + 1]; b->quals[i] = b->blob + b->off[3 * i + 2];
        b->lens[i] = (uint32_t)(b->off[3 * i + 2] - b->off[3 * i + 1] - 1);          /* the record's Read.length (src/Reads.cpp:96): the mapper does not measure it again */
    }
    *out = b;
    return LF_OK;
}

/* ---------------------------------------------------------------- file -> SAM, reading ahead of the GPU */
typedef struct {
    lf_reads_t *rd; int max_reads; uint64_t max_bases;
    pthread_mutex_t mu; pthread_cond_t cv;
    lf_read_batch_t *slot; int full, done, stop, rc; char err[512];
} ahead_t;
static void *ahead_main(void *arg)
{
    ahead_t *A = (ahead_t *)arg;
    for (;;) {
        lf_read_batch_t *b = NULL;
        const int rc = lf_reads_next(A->rd, A->max_reads, A->max_bases, &b);
        pthread_mutex_lock(&A->mu);
        while (A->full && !A->stop) pthread_cond_wait(&A->cv, &A->mu);
        if (A->stop) { pthread_mutex_unlock(&A->mu); lf_read_batch_free(b); return NULL; }
        if (rc != LF_OK) { A->rc = rc; snprintf(A->err, sizeof A->err, "%s", lf_last_error()); }
        A->slot = b; A->full = 1; A->done = (b == NULL);
        pthread_cond_broadcast(&A->cv);
        pthread_mutex_unlock(&A->mu);
        if (!b) return NULL;
    }
}

int lf_map_file(const lf_index_t *ix, const lf_params_t *p, const char *reads_path, const char *out_path, int no_header,
                const char *cmdline, int batch_reads, lf_stats_t *total)
{
    return lf_map_file_multi(&ix, 1, p, reads_path, out_path, no_header, cmdline, batch_reads, total);
}

/* ---- the writer: batch k's SAM text goes to the file while batch k + 1 is on the GPU.  A regular file is written by four
 * threads at once (pwrite into disjoint ranges: the copy into the page cache is what costs), anything else in order. ---- */
static double wall_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
typedef struct { int fd; const char *p; size_t n; off_t at; int rc; } wpiece_t;
static void *wpiece_main(void *arg)
{
    wpiece_t *w = (wpiece_t *)arg;
    size_t done = 0;
    while (done < w->n) {
        const ssize_t k = pwrite(w->fd, w->p + done, w->n - done > ((size_t)64 << 20) ? ((size_t)64 << 20) : w->n - done, w->at + (off_t)done);
        if (k <= 0) { w->rc = 1; return NULL; }
        done += (size_t)k;
    }
    return NULL;
}
typedef struct {
    FILE *fo; int regular; off_t at;
    pthread_mutex_t mu; pthread_cond_t cv;
    const char *buf; size_t len; int have, stop, busy, rc;
} writer_t;
static void *writer_main(void *arg)
{
    writer_t *W = (writer_t *)arg;
    for (;;) {
        pthread_mutex_lock(&W->mu);
        while (!W->have && !W->stop) pthread_cond_wait(&W->cv, &W->mu);
        if (!W->have && W->stop) { pthread_mutex_unlock(&W->mu); return NULL; }
        const char *buf = W->buf; const size_t len = W->len;
        W->have = 0; W->busy = 1;
        pthread_mutex_unlock(&W->mu);
        int rc = 0;
        if (W->regular && len > ((size_t)8 << 20)) {
            enum { NT = 4 };
            wpiece_t pc[NT]; pthread_t th[NT]; int started[NT];
            const size_t part = (len + NT - 1) / NT;
            for (int t = 0; t < NT; t++) {
                const size_t o = (size_t)t * part;
                pc[t].fd = fileno(W->fo); pc[t].p = buf + (o < len ? o : len); pc[t].n = o < len ? (len - o < part ? len - o : part) : 0; pc[t].at = W->at + (off_t)o; pc[t].rc = 0;
                started[t] = pc[t].n && pthread_create(&th[t], NULL, wpiece_main, &pc[t]) == 0;
                if (pc[t].n && !started[t]) wpiece_main(&pc[t]);
            }
            for (int t = 0; t < NT; t++) { if (started[t]) pthread_join(th[t], NULL); rc |= pc[t].rc; }
            W->at += (off_t)len;
        } else if (W->regular) {
            wpiece_t one; one.fd = fileno(W->fo); one.p = buf; one.n = len; one.at = W->at; one.rc = 0;
            wpiece_main(&one); rc = one.rc; W->at += (off_t)len;
        } else if (fwrite(buf, 1, len, W->fo) != len) rc = 1;
        pthread_mutex_lock(&W->mu);
        W->busy = 0; if (rc) W->rc = rc;
        pthread_cond_broadcast(&W->cv);
        pthread_mutex_unlock(&W->mu);
    }
}
typedef struct { size_t bytes; void *p; } palloc_t;
static void *palloc_main(void *arg) { palloc_t *P = (palloc_t *)arg; P->p = lfg_host_alloc(P->bytes); return NULL; }
static void writer_wait_idle(writer_t *W)
{
    pthread_mutex_lock(&W->mu);
    while (W->have || W->busy) pthread_cond_wait(&W->cv, &W->mu);
    pthread_mutex_unlock(&W->mu);
}

/* the same loop over several devices of this process (idx[d]: replica of the index on device d): every batch is
 * spread over all of them by lf_map_batch_multi; the output is the one-device output, byte for byte.
 * Three stages run at once: the reader parses batch k + 1, the GPU(s) map batch k into one of two reusable pinned buffers,
 * the writer puts batch k - 1 into the file. */
int lf_map_file_multi(const lf_index_t *const *idx, int n_idx, const lf_params_t *p, const char *reads_path, const char *out_path, int no_header,
                      const char *cmdline, int batch_reads, lf_stats_t *total)
{
    if (!idx || n_idx < 1 || !idx[0] || !p || !reads_path) { lf_set_error("lf_map_file: bad argument"); return LF_ERR_ARG; }
    const lf_index_t *ix = idx[0];
    lf_reads_t *rd = NULL;
    int rc = lf_reads_open(reads_path, &rd);
    if (rc != LF_OK) return rc;
    FILE *fo = (!out_path || !strcmp(out_path, "-")) ? stdout : fopen(out_path, "w");
    if (!fo) { lf_reads_close(rd); lf_set_error("lf_map_file: cannot write %s", out_path); return LF_ERR_IO; }
    if (!no_header) { char *h = lf_sam_header(ix, p, cmdline ? cmdline : ""); if (h) { fputs(h, fo); lf_free(h); } }
    fflush(fo);
    if (total) memset(total, 0, sizeof *total);
    ahead_t A; memset(&A, 0, sizeof A);
    /* batches of ~0.75 Gbp per device: large enough for the mapper's sixteen chunks in flight, small enough that reading batch
     * k + 1, mapping batch k and writing batch k - 1 overlap from the second batch on */
    A.rd = rd; A.max_reads = batch_reads > 0 ? batch_reads : 50000 * n_idx; A.max_bases = (768ull << 20) * (unsigned)n_idx;
    pthread_mutex_init(&A.mu, NULL); pthread_cond_init(&A.cv, NULL);
    writer_t W; memset(&W, 0, sizeof W);
    W.fo = fo;
    { struct stat sb; W.regular = fstat(fileno(fo), &sb) == 0 && S_ISREG(sb.st_mode); W.at = W.regular ? lseek(fileno(fo), 0, SEEK_CUR) : 0; if (W.at < 0) W.regular = 0; }
    pthread_mutex_init(&W.mu, NULL); pthread_cond_init(&W.cv, NULL);
    char *obuf[2] = { NULL, NULL }; size_t ocap[2] = { 0, 0 };
    pthread_t th, wth, ath; int have_w = 0, have_a = 0;
    if (pthread_create(&th, NULL, ahead_main, &A) != 0) { lf_set_error("lf_map_file: pthread_create failed"); rc = LF_ERR_NOMEM; goto out; }
    have_w = pthread_create(&wth, NULL, writer_main, &W) == 0;
    /* the two SAM buffers are pinned ONCE, sized for the largest batch the reader may deliver (pinning ~2.7 GB costs ~0.5 s:
     * the first one is pinned while the reader parses the first batch, the second one by a helper while the first batch maps) */
    const size_t cap_all = (size_t)(3.4 * (double)A.max_bases) + (size_t)A.max_reads * 1536 + ((size_t)64 << 20);
    palloc_t PA; PA.bytes = cap_all; PA.p = NULL;
    if (A.max_bases < (16ull << 30)) {
        obuf[0] = (char *)lfg_host_alloc(cap_all); ocap[0] = obuf[0] ? cap_all : 0;
        have_a = pthread_create(&ath, NULL, palloc_main, &PA) == 0;
    }
    const int ftiming = getenv("LF_TIMING") != NULL;
    for (int k = 0;; k++) {
        const double tq0 = ftiming ? wall_ms() : 0;
        pthread_mutex_lock(&A.mu);
        while (!A.full) pthread_cond_wait(&A.cv, &A.mu);
        lf_read_batch_t *b = A.slot; A.slot = NULL; A.full = 0;
        const int rrc = A.rc;
        pthread_cond_broadcast(&A.cv);
        pthread_mutex_unlock(&A.mu);
        if (rrc != LF_OK) { rc = rrc; lf_set_error("%s", A.err); lf_read_batch_free(b); break; }
        if (!b) break;
        /* this batch's buffer was last used two batches ago: with ONE writer in flight, waiting for the writer to go idle
         * before handing it the next text is enough -- but the buffer must also be free before the map starts */
        const int slot = k & 1;
        if (slot == 1 && have_a) { pthread_join(ath, NULL); have_a = 0; obuf[1] = (char *)PA.p; ocap[1] = obuf[1] ? cap_all : 0; }
        const double tq1 = ftiming ? wall_ms() : 0;
        size_t need = (size_t)(3.3 * (double)b->bases) + (size_t)b->n * 1536 + ((size_t)1 << 20);
        size_t len = 0; lf_stats_t st;
        for (int attempt = 0; attempt < 3; attempt++) {
            if (ocap[slot] < need) {
                if (have_w) writer_wait_idle(&W);                               /* nothing of ours may still be read */
                lfg_host_free(obuf[slot]); obuf[slot] = (char *)lfg_host_alloc(need); ocap[slot] = obuf[slot] ? need : 0;
                if (!obuf[slot]) { lf_set_error("lf_map_file: cannot allocate %zu bytes of pinned output", need); rc = LF_ERR_NOMEM; break; }
            }
            rc = lf_map_batch_multi(idx, n_idx, p, b->n, b->names, b->seqs, b->quals, b->lens, obuf[slot], ocap[slot], NULL, &len, &st);       /* the reader is already on the next batch */
            if (rc != LF_ERR_NOMEM) break;
            need = need * 2;                                                    /* "output buffer too small": unusually long records */
        }
        const double tq2 = ftiming ? wall_ms() : 0;
        if (rc == LF_OK) {
            if (have_w) {
                writer_wait_idle(&W);                                           /* the previous text is in the file: its buffer is free for batch k + 1 */
                pthread_mutex_lock(&W.mu); W.buf = obuf[slot]; W.len = len; W.have = 1; pthread_cond_broadcast(&W.cv); pthread_mutex_unlock(&W.mu);
            } else if (fwrite(obuf[slot], 1, len, fo) != len) { lf_set_error("lf_map_file: short write"); rc = LF_ERR_IO; }
            if (total) {
                total->ms_total += st.ms_total; total->n_reads += st.n_reads; total->n_bases += st.n_bases; total->n_seeds += st.n_seeds;
                total->n_edlib_problems += st.n_edlib_problems; total->n_chain_problems += st.n_chain_problems; total->n_ksw_problems += st.n_ksw_problems;
            }
        }
        if (ftiming) fprintf(stderr, "[lf] file batch %d: %d reads, waited for the reader %.1f ms, buffer + map %.1f ms (map %.1f), waited for the writer %.1f ms\n", k, b->n, tq1 - tq0, tq2 - tq1, st.ms_total, wall_ms() - tq2);
        lf_read_batch_free(b);
        if (rc != LF_OK) break;
        if (W.rc) { lf_set_error("lf_map_file: short write"); rc = LF_ERR_IO; break; }
    }
    /* stop the reader: it may be waiting for the slot to empty, or about to fill it */
    pthread_mutex_lock(&A.mu);
    A.stop = 1;
    if (A.full) { lf_read_batch_free(A.slot); A.slot = NULL; A.full = 0; }
    pthread_cond_broadcast(&A.cv);
    pthread_mutex_unlock(&A.mu);
    pthread_join(th, NULL);
    if (A.slot) lf_read_batch_free(A.slot);
    if (have_w) {
        writer_wait_idle(&W);
        pthread_mutex_lock(&W.mu); W.stop = 1; pthread_cond_broadcast(&W.cv); pthread_mutex_unlock(&W.mu);
        pthread_join(wth, NULL);
        if (W.rc && rc == LF_OK) { lf_set_error("lf_map_file: short write"); rc = LF_ERR_IO; }
        if (W.regular) (void)lseek(fileno(fo), W.at, SEEK_SET);
    }
out:
    if (have_a) { pthread_join(ath, NULL); obuf[1] = (char *)PA.p; }
    lfg_host_free(obuf[0]); lfg_host_free(obuf[1]);
    pthread_mutex_destroy(&A.mu); pthread_cond_destroy(&A.cv);
    pthread_mutex_destroy(&W.mu); pthread_cond_destroy(&W.cv);
    if (fo != stdout) fclose(fo); else fflush(fo);
    lf_reads_close(rd);
    return rc;
}
