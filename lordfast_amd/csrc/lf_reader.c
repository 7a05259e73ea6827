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
#include "lf_internal.h"

struct lf_reads {
    gzFile fp;
    unsigned char *buf; int beg, end, eof;
    int last_char;                     /* header character already consumed ('>' / '@'), 0 if none */
    char path[1024];
};
#define LF_RBUF (1 << 18)

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
/* append the rest of the current line (without the newline); returns the terminating character (-1 at EOF) */
static int gs_getline(struct lf_reads *r, gstr_t *g, int stop_at_space)
{
    for (;;) {
        /* bulk scan of the buffered bytes */
        while (r->beg < r->end) {
            const unsigned char c = r->buf[r->beg];
            if (c == '\n' || (stop_at_space && (c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'))) { r->beg++; return c; }
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
    *out = r;
    return LF_OK;
}
void lf_reads_close(lf_reads_t *r)
{
    if (!r) return;
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
    if (c != '\n' && c != -1) { gstr_t skip = { 0, 0, 0 }; (void)gs_getline(r, &skip, 0); free(skip.s); }      /* comment */
    while ((c = rd_getc(r)) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;                   /* empty line */
        gs_putc(seq, (char)c);
        (void)gs_getline(r, seq, 0);
        if (seq->n > 1 && seq->s[seq->n - 1] == '\r') seq->n--;          /* "\r\n" line ends (kseq.h:140) */
    }
    if (c == '>' || c == '@') r->last_char = c; else r->last_char = 0;
    gs_putc(name, 0); name->n--; gs_putc(seq, 0); seq->n--;
    if (c != '+') { gs_putc(qual, 0); qual->n--; return 1; }                 /* FASTA */
    { gstr_t skip = { 0, 0, 0 }; c = gs_getline(r, &skip, 0); free(skip.s); }  /* rest of the '+' line */
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

struct lf_read_batch {
    int n; uint64_t bases;
    const char **names, **seqs, **quals;
    char *blob; size_t blob_n, blob_cap;
    size_t *off; int cap;                          /* 3 offsets per record into blob */
};
void lf_read_batch_free(lf_read_batch_t *b)
{
    if (!b) return;
    free(b->names); free(b->seqs); free(b->quals); free(b->blob); free(b->off); free(b);
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

/* next batch: up to max_reads records / max_bases sequence bytes; *out = NULL at end of file */
int lf_reads_next(lf_reads_t *r, int max_reads, uint64_t max_bases, lf_read_batch_t **out)
{
    *out = NULL;
    if (max_reads <= 0) max_reads = 1 << 30;
    if (max_bases == 0) max_bases = ~0ull;
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
    for (int i = 0; i < b->n; i++) { b->names[i] = b->blob + b->off[3 * i]; b->seqs[i] = b->blob + b->off[3 * i + 1]; b->quals[i] = b->blob + b->off[3 * i + 2]; }
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

/* the same loop over several devices of this process (idx[d]: replica of the index on device d): every batch is
 * spread over all of them by lf_map_batch_multi; the output is the one-device output, byte for byte */
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
    if (total) memset(total, 0, sizeof *total);
    ahead_t A; memset(&A, 0, sizeof A);
    A.rd = rd; A.max_reads = batch_reads > 0 ? batch_reads : 100000 * n_idx; A.max_bases = (3ull << 30) * (unsigned)n_idx;
    pthread_mutex_init(&A.mu, NULL); pthread_cond_init(&A.cv, NULL);
    pthread_t th;
    if (pthread_create(&th, NULL, ahead_main, &A) != 0) { lf_set_error("lf_map_file: pthread_create failed"); rc = LF_ERR_NOMEM; goto out; }
    for (;;) {
        pthread_mutex_lock(&A.mu);
        while (!A.full) pthread_cond_wait(&A.cv, &A.mu);
        lf_read_batch_t *b = A.slot; A.slot = NULL; A.full = 0;
        const int rrc = A.rc;
        pthread_cond_broadcast(&A.cv);
        pthread_mutex_unlock(&A.mu);
        if (rrc != LF_OK) { rc = rrc; lf_set_error("%s", A.err); lf_read_batch_free(b); break; }
        if (!b) break;
        char *sam = NULL; size_t len = 0; lf_stats_t st;
        rc = lf_map_batch_multi(idx, n_idx, p, b->n, b->names, b->seqs, b->quals, NULL, NULL, 0, &sam, &len, &st);       /* the reader is already on the next batch */
        if (rc == LF_OK) {
            if (fwrite(sam, 1, len, fo) != len) { lf_set_error("lf_map_file: short write"); rc = LF_ERR_IO; }
            lf_free(sam);
            if (total) {
                total->ms_total += st.ms_total; total->n_reads += st.n_reads; total->n_bases += st.n_bases; total->n_seeds += st.n_seeds;
                total->n_edlib_problems += st.n_edlib_problems; total->n_chain_problems += st.n_chain_problems; total->n_ksw_problems += st.n_ksw_problems;
            }
        }
        lf_read_batch_free(b);
        if (rc != LF_OK) break;
    }
    /* stop the reader: it may be waiting for the slot to empty, or about to fill it */
    pthread_mutex_lock(&A.mu);
    A.stop = 1;
    if (A.full) { lf_read_batch_free(A.slot); A.slot = NULL; A.full = 0; }
    pthread_cond_broadcast(&A.cv);
    pthread_mutex_unlock(&A.mu);
    pthread_join(th, NULL);
    if (A.slot) lf_read_batch_free(A.slot);
out:
    pthread_mutex_destroy(&A.mu); pthread_cond_destroy(&A.cv);
    if (fo != stdout) fclose(fo); else fflush(fo);
    lf_reads_close(rd);
    return rc;
}
