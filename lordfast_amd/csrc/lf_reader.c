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
#include "lf_batch.h"

/* How a file is read:
 *   window parser   plain FASTA / FASTQ, BGZF and gzip files whose text starts with '>' or '@': a WINDOW of text (a slice of the
 *                   mapped file, or freshly inflated bytes) is cut into pieces at record starts and parsed by several threads;
 *                   BGZF blocks (gzip members with a 'BC' size field, what bgzip writes) are inflated by several threads too.
 *                   FASTQ records must be the usual four lines; anything the fast grammar does not cover (wrapped FASTQ, '@' /
 *                   '+' / '>' line starts inside a FASTA record, junk between records) hands the rest of the file to the
 *   sequential parser, which states kseq's grammar byte by byte over gzread (any input, pipes included). */
struct lf_reads {
    char path[1024];
    /* sequential parser */
    gzFile fp;
    unsigned char *buf; int beg, end, eof;
    int last_char;                     /* header character already consumed ('>' / '@'), 0 if none */
    /* window parser */
    int win;                           /* 1: active */
    int src;                           /* 0 plain text (mapped), 1 BGZF, 2 gzip stream (one or several members, inflated by one thread) */
    int fmt;                           /* '>' or '@' */
    const unsigned char *map; size_t map_size;      /* the file as it is on disk */
    size_t cpos;                       /* compressed sources: first byte of the file not yet inflated */
    unsigned char *tbuf; size_t tcap, tbeg, tend;   /* compressed sources: inflated text not yet consumed = tbuf[tbeg .. tend) */
    int src_eof;
    uint64_t text_pos;                 /* offset (in the uncompressed text) of the first unconsumed byte */
    z_stream zs; int zs_live;
    double avg_rec;                    /* bytes of text per record, from the batches so far (window sizing) */
    double inflate_cpu_s;              /* CPU seconds spent in inflate (all threads): LF_TIMING prints it at close */
};
#define LF_RBUF (4 << 20)
#define LF_PARSE_THREADS 8
#define LF_INFLATE_THREADS 16

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

static int cpu_budget(int cap)
{
    int nt = (int)sysconf(_SC_NPROCESSORS_ONLN);
    FILE *fq = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (fq) {
        long long period = 0; char qs[64];
        if (fscanf(fq, "%63s %lld", qs, &period) == 2 && strcmp(qs, "max") != 0 && period > 0) { const int lim = (int)((atoll(qs) + period - 1) / period); if (lim >= 1 && lim < nt) nt = lim; }
        fclose(fq);
    }
    if (nt > cap) nt = cap;
    return nt < 1 ? 1 : nt;
}
static double thread_cpu_s(void) { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

/* BGZF block header at p (RFC 1952 member with an extra subfield 'B' 'C' holding the block size - 1): returns the block's size
 * and where its deflate data starts, 0 if p is not such a header */
static size_t bgzf_block(const unsigned char *p, size_t avail, size_t *data_off)
{
    if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (avail < 12 + xlen) return 0;
    size_t bsize = 0;
    for (size_t o = 12; o + 4 <= 12 + xlen; ) {
        const size_t slen = (size_t)p[o + 2] | ((size_t)p[o + 3] << 8);
        if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) bsize = ((size_t)p[o + 4] | ((size_t)p[o + 5] << 8)) + 1;
        o += 4 + slen;
    }
    if (!bsize || bsize > avail || bsize < 12 + xlen + 8 || (p[3] & ~4)) return 0;      /* other header fields (name, comment, crc): not what bgzip writes */
    *data_off = 12 + xlen;
    return bsize;
}

static void blob_pool_reader(int delta);
int lf_reads_open(const char *path, lf_reads_t **out)
{
    if (!path || !out) { lf_set_error("lf_reads_open: bad argument"); return LF_ERR_ARG; }
    gzFile fp = gzopen(path, "r");
    if (!fp) { lf_set_error("lf_reads_open: cannot open %s", path); return LF_ERR_IO; }
    (void)gzbuffer(fp, 1 << 20);
    struct lf_reads *r = (struct lf_reads *)calloc(1, sizeof *r);
    r->fp = fp; r->buf = (unsigned char *)malloc(LF_RBUF);
    snprintf(r->path, sizeof r->path, "%s", path);
    if (!lf_env_set("LF_READER_SEQUENTIAL")) {
        /* a regular file: map it and look at the first bytes (of the text, for gzip) */
        const int fd = open(path, O_RDONLY);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            void *m = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                const unsigned char *mp = (const unsigned char *)m; const size_t ms = (size_t)sb.st_size;
                int first = -1, src = 0;
                if (ms >= 2 && mp[0] == 0x1f && mp[1] == 0x8b) {
                    size_t doff; src = bgzf_block(mp, ms, &doff) ? 1 : 2;
                    unsigned char c1; z_stream z; memset(&z, 0, sizeof z);
                    if (inflateInit2(&z, 15 + 32) == Z_OK) {
                        z.next_in = (Bytef *)mp; z.avail_in = (uInt)(ms > (1u << 20) ? (1u << 20) : ms); z.next_out = &c1; z.avail_out = 1;
                        /* an empty first member (bgzip's end marker, or an empty file) yields nothing: leave it to the sequential parser */
                        const int zr = inflate(&z, Z_NO_FLUSH);
                        if ((zr == Z_OK || zr == Z_STREAM_END) && z.avail_out == 0) first = c1;
                        inflateEnd(&z);
                    }
                } else first = mp[0];
                if (first == '>' || first == '@') {
                    r->map = mp; r->map_size = ms; r->win = 1; r->src = src; r->fmt = first;
                    (void)madvise(m, ms, MADV_SEQUENTIAL);
                } else munmap(m, ms);
            }
        }
        if (fd >= 0) close(fd);
    }
    blob_pool_reader(+1);
    *out = r;
    return LF_OK;
}
static void blob_pool_trim(void);
void lf_reads_close(lf_reads_t *r)
{
    if (!r) return;
    if (lf_env_set("LF_TIMING") && r->inflate_cpu_s > 0) fprintf(stderr, "[lf] reader %s: %.2f CPU-s in inflate (%s)\n", r->path, r->inflate_cpu_s, r->src == 1 ? "BGZF blocks, several threads" : "one gzip stream, one thread");
    blob_pool_reader(-1);                       /* the last open reader takes the cached blobs with it (batches freed after that are freed, not cached) */
    if (r->zs_live) inflateEnd(&r->zs);
    if (r->map) munmap((void *)r->map, r->map_size);
    free(r->tbuf);
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

/* Blobs of the window parser's batches are recycled: a batch is ~0.75 GB in eight blobs, and memory fresh from malloc (= mmap
 * at that size) costs a page fault and a zeroed page per 4 KiB on first touch and an munmap on free -- more than the parse
 * itself (1 M reads: the reader took 3.0 s, the mapper 2.6 s).  Up to 32 freed blobs wait here for the next batch. */
#define LF_BLOB_POOL 32
static struct { char *p; size_t cap; } g_blob_pool[LF_BLOB_POOL];
static pthread_mutex_t g_blob_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_blob_readers;                       /* open readers: the pool serves all of them and is emptied when the last one closes */
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
static void blob_pool_reader(int delta)
{
    pthread_mutex_lock(&g_blob_mu);
    g_blob_readers += delta;
    const int last = g_blob_readers <= 0;
    if (last) g_blob_readers = 0;
    pthread_mutex_unlock(&g_blob_mu);
    if (last && delta < 0) blob_pool_trim();
}
static void blob_put(char *p, size_t cap)
{
    if (!p) return;
    pthread_mutex_lock(&g_blob_mu);
    int slot = -1, smallest = -1;
    if (g_blob_readers <= 0) { pthread_mutex_unlock(&g_blob_mu); free(p); return; }      /* nobody left to reuse it (a batch freed after its reader was closed) */
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

void lf_read_batch_free(lf_read_batch_t *b)
{
    if (!b) return;
    for (int t = 0; t < b->nblobs; t++) blob_put(b->blobs[t], b->blob_caps[t]);
    free(b->blobs); free(b->blob_caps);
    lf_prepack_free(b->pre);
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

/* ================================================================ the window parser
 * A window = [a, e) of text that starts at a record.  It is cut into one piece per thread at record starts; every thread
 * copies its records' names, sequences (lines concatenated) and qualities into a blob of its own.  Same grammar as read_record
 * for what it accepts: the name ends at the first white space, "\r\n" line ends and empty lines are dropped.
 *   FASTA   a record runs to the next line that starts with '>'; a line that starts with '@' or '+' inside a record would start
 *           a new record / a quality string in kseq: not ours (weird)
 *   FASTQ   '@' header line, ONE sequence line, '+' line, ONE quality line of the same length, then '@' again (or blank lines,
 *           or the end); anything else -- wrapped records, '>' records in between, junk -- is weird
 * weird = the sequential parser takes over at the start of the batch.  The last piece of a window that is not the end of the
 * text stops in front of the first record it cannot see the end of. */
typedef struct {
    const unsigned char *p, *end; const unsigned char *win0; int fmt, open_end;
    char *blob; size_t n, cap;
    size_t *off /* 3 per record: name, seq, qual (== seq's NUL for FASTA) */, *rec_off /* offset of the record in the window */; int nrec, caprec;
    uint64_t bases; int weird; size_t consumed;      /* offset in the window behind the last complete record (+ blank lines) */
} piece_t;
static inline const unsigned char *line_end(const unsigned char *p, const unsigned char *end) { const unsigned char *nl = (const unsigned char *)memchr(p, '\n', (size_t)(end - p)); return nl ? nl : end; }
static const unsigned char *next_record_start(int fmt, const unsigned char *p, const unsigned char *end)
{   /* first record start at a line start behind p; `end` if none can be confirmed inside the window */
    while (p < end) {
        const unsigned char *nl = (const unsigned char *)memchr(p, '\n', (size_t)(end - p));
        if (!nl || nl + 1 >= end) return end;
        const unsigned char *l0 = nl + 1;
        if (fmt == '>') { if (*l0 == '>') return l0; }
        else if (*l0 == '@') {
            /* a header, not a quality line that happens to start with '@': the line after next starts with '+' (after a quality
             * line come a header and a sequence line), and the lines around it have one length */
            const unsigned char *e0 = line_end(l0, end); if (e0 >= end) return end;
            const unsigned char *l1 = e0 + 1, *e1 = line_end(l1, end); if (e1 >= end) return end;
            const unsigned char *l2 = e1 + 1; if (l2 >= end) return end;
            if (*l2 == '+') {
                const unsigned char *e2 = line_end(l2, end); if (e2 >= end) return end;
                const unsigned char *l3 = e2 + 1, *e3 = line_end(l3, end);
                if (e3 >= end) return end;
                if (e3 - l3 == e1 - l1) return l0;
            }
        }
        p = l0;
    }
    return end;
}
static inline int is_space_c(unsigned char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }
static void *piece_main(void *arg)
{
    piece_t *M = (piece_t *)arg;
    const unsigned char *p = M->p, *end = M->end;
    M->blob = blob_get((size_t)(end - p) + 64, &M->cap); M->n = 0;
    M->consumed = (size_t)(p - M->win0);
    while (p < end) {
        if (*p == '\n' || (*p == '\r' && p + 1 < end && p[1] == '\n')) { p += (*p == '\r') ? 2 : 1; if (M->nrec || M->fmt == '@') { M->consumed = (size_t)(p - M->win0); continue; } M->weird = 1; return NULL; }   /* blank line between records */
        if (*p != (unsigned char)M->fmt) { M->weird = 1; return NULL; }
        if (M->nrec == M->caprec) { M->caprec = M->caprec ? M->caprec * 2 : 4096; M->off = (size_t *)realloc(M->off, (size_t)M->caprec * 3 * sizeof(size_t)); M->rec_off = (size_t *)realloc(M->rec_off, (size_t)M->caprec * sizeof(size_t)); }
        const size_t n0 = M->n;
        const unsigned char *rec = p;
        const unsigned char *q = p + 1, *le = line_end(q, end);
        if (le >= end && M->open_end) break;                          /* header line not complete */
        const unsigned char *ne = q;
        while (ne < le && !is_space_c(*ne)) ne++;
        size_t *o = &M->off[3 * M->nrec];
        o[0] = M->n; memcpy(M->blob + M->n, q, (size_t)(ne - q)); M->n += (size_t)(ne - q); M->blob[M->n++] = 0;
        o[1] = M->n;
        p = le < end ? le + 1 : end;
        if (M->fmt == '>') {
            int complete = 0;
            while (p < end) {
                if (*p == '>') { complete = 1; break; }
                if (*p == '@' || *p == '+') { M->weird = 1; return NULL; }
                le = line_end(p, end);
                if (le >= end && M->open_end) break;                  /* the line may go on behind the window */
                const size_t len = (size_t)(le - p);
                if (len > 0) {
                    memcpy(M->blob + M->n, p, len); M->n += len;
                    if (M->n - o[1] > 1 && M->blob[M->n - 1] == '\r') M->n--;      /* kseq.h:140 */
                }
                p = le < end ? le + 1 : end;
            }
            if (!complete && M->open_end) { M->n = n0; break; }      /* its end is not in sight */
            M->bases += M->n - o[1];
            M->blob[M->n] = 0; o[2] = M->n; M->n++;                  /* QUAL "" = the sequence's terminator */
        } else {
            /* sequence line */
            if (p >= end) { if (M->open_end) { M->n = n0; break; } M->weird = 1; return NULL; }
            if (*p == '>' || *p == '@' || *p == '+' || *p == '\n') { M->weird = 1; return NULL; }
            le = line_end(p, end);
            if (le >= end) { if (M->open_end) { M->n = n0; break; } M->weird = 1; return NULL; }      /* a FASTQ record that ends with its sequence line: kseq's business */
            size_t len = (size_t)(le - p);
            memcpy(M->blob + M->n, p, len); M->n += len;
            if (len > 1 && M->blob[M->n - 1] == '\r') { M->n--; len--; }
            M->blob[M->n++] = 0;
            p = le + 1;
            /* '+' line */
            if (p >= end) { if (M->open_end) { M->n = n0; break; } M->weird = 1; return NULL; }
            if (*p != '+') { M->weird = 1; return NULL; }
            le = line_end(p, end);
            if (le >= end) { if (M->open_end) { M->n = n0; break; } M->weird = 1; return NULL; }
            p = le + 1;
            /* quality line: complete when its newline is in sight, or at the end of the text */
            le = line_end(p, end);
            if (le >= end && M->open_end) { M->n = n0; break; }
            size_t ql = (size_t)(le - p);
            if (ql > 1 && p[ql - 1] == '\r') ql--;
            if (ql != len) { M->weird = 1; return NULL; }             /* wrapped or truncated: kseq's business */
            o[2] = M->n; memcpy(M->blob + M->n, p, ql); M->n += ql; M->blob[M->n++] = 0;
            M->bases += len;
            p = le < end ? le + 1 : end;
        }
        M->rec_off[M->nrec] = (size_t)(rec - M->win0);
        M->nrec++;
        M->consumed = (size_t)(p - M->win0);
    }
    return NULL;
}

/* ---- text for the window parser ---- */
typedef struct { const unsigned char *src; size_t n_in; unsigned char *dst; size_t n_out; } zblock_t;
typedef struct { zblock_t *blk; int n; int rc; double cpu_s; } zjob_t;
static void *zjob_main(void *arg)
{
    zjob_t *J = (zjob_t *)arg;
    const double c0 = thread_cpu_s();
    z_stream z; memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) { J->rc = 1; return NULL; }
    for (int k = 0; k < J->n; k++) {
        if (k && inflateReset(&z) != Z_OK) { J->rc = 1; break; }
        z.next_in = (Bytef *)J->blk[k].src; z.avail_in = (uInt)J->blk[k].n_in; z.next_out = J->blk[k].dst; z.avail_out = (uInt)J->blk[k].n_out;
        const int zr = inflate(&z, Z_FINISH);
        if (zr != Z_STREAM_END || z.avail_out != 0) { J->rc = 1; break; }
    }
    inflateEnd(&z);
    J->cpu_s = thread_cpu_s() - c0;
    return NULL;
}
static int tbuf_room(struct lf_reads *r, size_t extra)
{
    if (r->tbeg > 0) { memmove(r->tbuf, r->tbuf + r->tbeg, r->tend - r->tbeg); r->tend -= r->tbeg; r->tbeg = 0; }
    if (r->tend + extra + 64 > r->tcap) {
        const size_t nc = (r->tend + extra) + (r->tend + extra) / 4 + (1 << 20);
        unsigned char *nb = (unsigned char *)realloc(r->tbuf, nc);
        if (!nb) { lf_set_error("lf_reads_next: out of memory"); return LF_ERR_NOMEM; }
        r->tbuf = nb; r->tcap = nc;
    }
    return LF_OK;
}
/* compressed sources: make tbuf[tbeg .. tend) at least `want` bytes long (or reach the end of the file) */
static int inflate_more(struct lf_reads *r, size_t want)
{
    while (!r->src_eof && r->tend - r->tbeg < want) {
        const size_t need = want - (r->tend - r->tbeg);
        if (r->src == 1) {
            /* the BGZF blocks that hold the next `need` bytes: sizes from the headers (one pass over 18 bytes per 64 KiB), then
             * every thread inflates a run of blocks straight into its place */
            size_t cp = r->cpos, out = 0; int nb = 0, capb = 0; zblock_t *blk = NULL;
            while (cp < r->map_size && out < need) {
                size_t doff; const size_t bs = bgzf_block(r->map + cp, r->map_size - cp, &doff);
                if (!bs) break;
                const unsigned char *tl = r->map + cp + bs - 4;
                const size_t isize = (size_t)tl[0] | ((size_t)tl[1] << 8) | ((size_t)tl[2] << 16) | ((size_t)tl[3] << 24);
                if (isize > 65536) break;                               /* not a BGZF block (they hold at most 64 KiB): the streaming inflater takes it from here */
                if (nb == capb) { capb = capb ? capb * 2 : 4096; blk = (zblock_t *)realloc(blk, (size_t)capb * sizeof(zblock_t)); }
                blk[nb].src = r->map + cp + doff; blk[nb].n_in = bs - doff - 8; blk[nb].dst = (unsigned char *)(uintptr_t)out; blk[nb].n_out = isize; nb++;
                out += isize; cp += bs;
            }
            if (nb == 0) {
                if (cp >= r->map_size) { r->src_eof = 1; free(blk); break; }
                free(blk);
                /* a plain gzip member (or a BGZF block with extra flags) in the middle of a BGZF file -- `cat a.bgz b.gz`: gzread-based readers
                 * (the reference's kseq) read such files, so the rest of the file goes through the one-stream inflater below */
                if (cp + 2 <= r->map_size && r->map[cp] == 0x1f && r->map[cp + 1] == 0x8b) { r->src = 2; r->cpos = cp; continue; }
                lf_set_error("lf_reads_next: %s: neither a BGZF block nor a gzip member at offset %zu", r->path, cp); return LF_ERR_IO;
            }
            const int rc0 = tbuf_room(r, out); if (rc0 != LF_OK) { free(blk); return rc0; }
            for (int k = 0; k < nb; k++) blk[k].dst = r->tbuf + r->tend + (size_t)(uintptr_t)blk[k].dst;
            int nt = cpu_budget(LF_INFLATE_THREADS); if (nt > nb) nt = nb;
            zjob_t J[LF_INFLATE_THREADS]; pthread_t th[LF_INFLATE_THREADS]; int started[LF_INFLATE_THREADS];
            for (int t = 0; t < nt; t++) { const int b0 = (int)((long long)nb * t / nt), b1 = (int)((long long)nb * (t + 1) / nt); J[t].blk = blk + b0; J[t].n = b1 - b0; J[t].rc = 0; J[t].cpu_s = 0; }
            for (int t = 1; t < nt; t++) started[t] = pthread_create(&th[t], NULL, zjob_main, &J[t]) == 0;
            zjob_main(&J[0]);
            int bad = J[0].rc;
            for (int t = 1; t < nt; t++) { if (started[t]) pthread_join(th[t], NULL); else zjob_main(&J[t]); bad |= J[t].rc; }
            for (int t = 0; t < nt; t++) r->inflate_cpu_s += J[t].cpu_s;
            free(blk);
            if (bad) { lf_set_error("lf_reads_next: %s: corrupt BGZF block", r->path); return LF_ERR_IO; }
            r->tend += out; r->cpos = cp;
            if (r->cpos >= r->map_size) r->src_eof = 1;
        } else {
            /* one gzip stream (possibly several members back to back): one thread, zlib's speed */
            const double c0 = thread_cpu_s();
            if (!r->zs_live) { memset(&r->zs, 0, sizeof r->zs); if (inflateInit2(&r->zs, 15 + 32) != Z_OK) { lf_set_error("lf_reads_next: inflateInit failed"); return LF_ERR_IO; } r->zs_live = 1; }
            size_t chunk = need < ((size_t)8 << 20) ? ((size_t)8 << 20) : need;
            const int rc0 = tbuf_room(r, chunk); if (rc0 != LF_OK) return rc0;
            while (chunk > 0 && !r->src_eof) {
                const size_t in_left = r->map_size - r->cpos;
                if (in_left == 0) { r->src_eof = 1; break; }
                r->zs.next_in = (Bytef *)(r->map + r->cpos); r->zs.avail_in = (uInt)(in_left > (1u << 30) ? (1u << 30) : in_left);
                r->zs.next_out = r->tbuf + r->tend; r->zs.avail_out = (uInt)(chunk > (1u << 30) ? (1u << 30) : chunk);
                const uInt in0 = r->zs.avail_in, out0 = r->zs.avail_out;
                const int zr = inflate(&r->zs, Z_NO_FLUSH);
                r->cpos += in0 - r->zs.avail_in; r->tend += out0 - r->zs.avail_out; chunk -= out0 - r->zs.avail_out;
                if (zr == Z_STREAM_END) {                               /* next member, if any (gzread does the same) */
                    if (r->cpos >= r->map_size) { r->src_eof = 1; break; }
                    if (r->map[r->cpos] != 0x1f) { r->src_eof = 1; break; }      /* trailing garbage: ignored like gzread does */
                    if (inflateReset(&r->zs) != Z_OK) { lf_set_error("lf_reads_next: inflateReset failed"); return LF_ERR_IO; }
                } else if (zr != Z_OK && zr != Z_BUF_ERROR) { lf_set_error("lf_reads_next: %s: corrupt gzip data", r->path); return LF_ERR_IO; }
                else if (zr == Z_BUF_ERROR && in0 == r->zs.avail_in && out0 == r->zs.avail_out) { r->src_eof = 1; break; }      /* truncated file */
            }
            r->inflate_cpu_s += thread_cpu_s() - c0;
        }
    }
    return LF_OK;
}

static void batch_room(lf_read_batch_t *b, int more)
{
    if (b->n + more + 1 <= b->rcap) return;
    b->rcap = (b->n + more + 1) + (b->n + more + 1) / 2;
    b->names = (const char **)realloc(b->names, (size_t)b->rcap * sizeof(char *)); b->seqs = (const char **)realloc(b->seqs, (size_t)b->rcap * sizeof(char *));
    b->quals = (const char **)realloc(b->quals, (size_t)b->rcap * sizeof(char *)); b->lens = (uint32_t *)realloc(b->lens, (size_t)b->rcap * sizeof(uint32_t));
}
static int lf_reads_next_window(lf_reads_t *r, int max_reads, uint64_t max_bases, lf_read_batch_t **out)
{
    *out = NULL;
    lf_read_batch_t *b = (lf_read_batch_t *)calloc(1, sizeof *b);
    const uint64_t batch_text0 = r->text_pos;
    const double per_base = r->fmt == '@' ? 2.04 : 1.02;          /* bytes of text per base: sequence (+ quality) + newlines */
    size_t grow = 0;                                                /* a window that holds no complete record is doubled */
    for (;;) {
        /* how much text the rest of the batch is likely to be: bounded by BOTH limits (a batch limited by reads only must not
         * parse the rest of the file), from the record size seen so far; the first window of a file is a small probe */
        const double rec = r->avg_rec > 0 ? r->avg_rec : 0;
        uint64_t want = (uint64_t)1 << 40;
        if (max_bases - b->bases < ((uint64_t)1 << 38)) want = (uint64_t)((double)(max_bases - b->bases) * per_base) + 4096;
        const uint64_t by_reads = rec > 0 ? (uint64_t)((double)(max_reads - b->n) * rec * 1.05) + 65536 : ((uint64_t)4 << 20);
        if (by_reads < want) want = by_reads;
        if (rec <= 0 && want > ((uint64_t)64 << 20)) want = (uint64_t)64 << 20;
        if (r->src != 0 && want > ((uint64_t)256 << 20)) want = (uint64_t)256 << 20;      /* inflated text is buffered: a window at a time */
        if (want < grow) want = grow;
        const unsigned char *a, *e; int at_eof;
        if (r->src == 0) {
            a = r->map + r->text_pos; const size_t left = r->map_size - (size_t)r->text_pos;
            e = a + (want < left ? (size_t)want : left); at_eof = e == r->map + r->map_size;
        } else {
            const int irc = inflate_more(r, (size_t)want);
            if (irc != LF_OK) { lf_read_batch_free(b); return irc; }
            a = r->tbuf + r->tbeg; e = r->tbuf + r->tend; at_eof = r->src_eof;
            if ((size_t)(e - a) > want && !at_eof) e = a + want;
            if (e < r->tbuf + r->tend) at_eof = 0;
        }
        if (a >= e) break;                                          /* end of the text */
        int nt = cpu_budget(LF_PARSE_THREADS);
        if ((size_t)(e - a) < ((size_t)4 << 20)) nt = 1;
        piece_t M[LF_PARSE_THREADS]; pthread_t th[LF_PARSE_THREADS]; int started[LF_PARSE_THREADS];
        memset(M, 0, sizeof M);
        const unsigned char *cut = a;
        for (int t = 0; t < nt; t++) {
            const unsigned char *nx = (t == nt - 1) ? e : next_record_start(r->fmt, a + (size_t)(e - a) / (size_t)nt * (size_t)(t + 1), e);
            if (nx < cut) nx = cut;
            M[t].p = cut; M[t].end = nx; M[t].win0 = a; M[t].fmt = r->fmt; M[t].open_end = (nx == e) && !at_eof;
            cut = nx;
        }
        for (int t = 0; t < nt; t++) started[t] = (t > 0 && M[t].p < M[t].end) ? pthread_create(&th[t], NULL, piece_main, &M[t]) == 0 : 0;
        piece_main(&M[0]);
        for (int t = 1; t < nt; t++) { if (started[t]) pthread_join(th[t], NULL); else if (M[t].p < M[t].end) piece_main(&M[t]); }
        int weird = 0, total = 0;
        for (int t = 0; t < nt; t++) { weird |= M[t].weird; total += M[t].nrec; }
        /* a piece that stopped early (open end) must be the last one with any text */
        for (int t = 0; t + 1 < nt; t++) if (M[t].p < M[t].end && M[t].consumed != (size_t)(M[t].end - a)) { int later = 0; for (int u = t + 1; u < nt; u++) later |= M[u].p < M[u].end; if (later) weird = 1; }
        if (weird) {
            /* not what the window grammar covers: the sequential parser takes over from the start of this BATCH */
            for (int t = 0; t < nt; t++) { blob_put(M[t].blob, M[t].cap); free(M[t].off); free(M[t].rec_off); }
            lf_read_batch_free(b);
            r->win = 0;
            if (gzseek(r->fp, (z_off_t)batch_text0, SEEK_SET) < 0) { lf_set_error("lf_reads_next: cannot reposition %s", r->path); return LF_ERR_IO; }
            r->beg = r->end = 0; r->eof = 0; r->last_char = 0;
            return lf_reads_next(r, max_reads, max_bases, out);
        }
        /* keep the leading records that fit (at least one per batch); the text continues at the first one that does not */
        size_t consumed = 0; int stop = 0, kept = 0; uint64_t kept_bytes = 0;
        batch_room(b, total);
        for (int t = 0; t < nt; t++) { if (M[t].p < M[t].end) consumed = M[t].consumed; }
        for (int t = 0; t < nt && !stop; t++) for (int k = 0; k < M[t].nrec; k++) {
            const size_t *o = &M[t].off[3 * k];
            const uint32_t len = (uint32_t)(o[2] - o[1] - (r->fmt == '@' ? 1 : 0));
            if (b->n > 0 && (b->n >= max_reads || b->bases >= max_bases)) { consumed = M[t].rec_off[k]; stop = 1; break; }
            b->names[b->n] = M[t].blob + o[0]; b->seqs[b->n] = M[t].blob + o[1]; b->quals[b->n] = M[t].blob + o[2];
            b->lens[b->n] = len; b->bases += len; b->n++; kept++;
        }
        kept_bytes = consumed;
        if (b->nblobs + nt > b->capblobs) { b->capblobs = (b->nblobs + nt) * 2; b->blobs = (char **)realloc(b->blobs, (size_t)b->capblobs * sizeof(char *)); b->blob_caps = (size_t *)realloc(b->blob_caps, (size_t)b->capblobs * sizeof(size_t)); }
        for (int t = 0; t < nt; t++) { if (M[t].blob) { b->blobs[b->nblobs] = M[t].blob; b->blob_caps[b->nblobs] = M[t].cap; b->nblobs++; } free(M[t].off); free(M[t].rec_off); }
        if (kept > 0) r->avg_rec = r->avg_rec > 0 ? 0.5 * r->avg_rec + 0.5 * ((double)kept_bytes / kept) : (double)kept_bytes / kept;
        r->text_pos += consumed;
        if (r->src != 0) r->tbeg += consumed;
        if (stop || b->n >= max_reads || b->bases >= max_bases) break;
        if (at_eof) {
            if (consumed < (size_t)(e - a) && total == 0) { r->text_pos += (size_t)(e - a) - consumed; if (r->src != 0) r->tbeg = r->tend; }
            break;
        }
        if (total == 0) grow = (size_t)(e - a) * 2 + ((size_t)1 << 20);      /* a record longer than the window */
        else grow = 0;
    }
    if (b->n == 0) { lf_read_batch_free(b); return LF_OK; }
    *out = b;
    return LF_OK;
}

/* next batch: up to max_reads records / max_bases sequence bytes; *out = NULL at end of file */
int lf_reads_next(lf_reads_t *r, int max_reads, uint64_t max_bases, lf_read_batch_t **out)
{
    *out = NULL;
    if (max_reads <= 0) max_reads = 1 << 30;
    if (max_bases == 0) max_bases = ~0ull;
    if (r->win) return lf_reads_next_window(r, max_reads, max_bases, out);
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
    {   /* pwrite at explicit offsets needs a descriptor whose position is ours alone: on an O_APPEND descriptor (`-o - >> out.sam`)
         * Linux ignores the offset and appends, so the four pieces of a batch would land in completion order */
        struct stat sb; const int fl = fcntl(fileno(fo), F_GETFL);
        W.regular = fstat(fileno(fo), &sb) == 0 && S_ISREG(sb.st_mode) && fl != -1 && !(fl & O_APPEND);
        W.at = W.regular ? lseek(fileno(fo), 0, SEEK_CUR) : 0; if (W.at < 0) W.regular = 0;
    }
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
    const int ftiming = lf_env_set("LF_TIMING");
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
