/* lf_pipe.h -- private to the host glue of lf_map_batch (lf_pipeline.c, lf_sched.c, lf_replay.c, lf_samdesc.c, lf_crosscheck.c):
 * the chunk context, the data model of src/LordFAST.h:43-118 as the glue keeps it, small containers, and the few functions
 * the five files call across each other.
 *
 *   lf_sched.c       lanes and batches: worker pool, lane allocator, chunk order / output placement, the lf_map_batch* entry points
 *   lf_pipeline.c    one chunk through the stages (map_chunk): which kernel stage runs when, what the host decides in between
 *   lf_replay.c      alignChain_edlib (src/LordFAST.cpp:1765-2258) as a host REPLAY for the few chains that leave the device's common path
 *   lf_samdesc.c     printSamEntry (src/LordFAST.cpp:318-459): MAPQ, flags, SA:Z, one 48-byte descriptor per line; host fill of SEQ / QUAL
 *   lf_crosscheck.c  host re-implementations of device stages: a TEST library (liblfxcheck.so), loaded by lf_debug_crosscheck() only
 */
#ifndef LF_PIPE_H
#define LF_PIPE_H
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "lf_internal.h"
#include "lf_stdsort.h"

#pragma GCC diagnostic ignored "-Wunused-function"      /* the static helpers below are not all used by every file */

void lf_sort_seeds_by_qpos(Seed_t *s, long n);

/* src/LordFAST.cpp:88-92 */
#define CLIP_LEN    500
#define CLIP_SIM    0.75
#define SPLIT_LEN   80
#define SPLIT_SIM   0.40
#define REVERSE_SIM 0.60

#define LF_XC_HOST_VOTE  1u
#define LF_XC_HOST_CIGAR 2u
#define LF_XC_HOST_WALK  4u
#define LF_XC_HOST_SAM   8u
extern volatile unsigned g_crosscheck;      /* lf_debug_crosscheck(): which host cross-check implementations are switched in (0 = product path) */

#define LF_RC_SPLIT 100        /* internal: map_chunk wants its chunk cut in two (too many seed hits for one vote sort) */
double now_ms(void);

/* ---------------------------------------------------------------- small containers */
/* ---- per-worker bump arenas: every chunk-lifetime object of the host glue (seed lists, jobs, memos, CIGAR/MD strings,
 * ...) is carved from the arena of the thread that creates it and released by ONE reset at the end of the chunk.
 * With several chunks in flight, malloc/free pairs that cross threads contend on glibc's arena locks; a bump allocator
 * has no locks and no per-object free at all. ---- */
typedef struct { char **blk; size_t *bsz; int nblk, cur; size_t off; } arena_t;
#define AR_BLOCK ((size_t)8 << 20)
static void *ar_alloc(arena_t *a, size_t n)
{
    n = (n + 15) & ~(size_t)15;
    for (;;) {
        if (a->cur < a->nblk && a->off + n <= a->bsz[a->cur]) { void *p = a->blk[a->cur] + a->off; a->off += n; return p; }
        if (a->cur + 1 < a->nblk && n <= a->bsz[a->cur + 1]) { a->cur++; a->off = 0; continue; }
        /* new block (inserted after the current one so that larger requests do not strand the rest) */
        size_t sz = n > AR_BLOCK ? n : AR_BLOCK;
        a->blk = (char **)realloc(a->blk, ((size_t)a->nblk + 1) * sizeof(char *)); a->bsz = (size_t *)realloc(a->bsz, ((size_t)a->nblk + 1) * sizeof(size_t));
        int at = a->nblk ? a->cur + 1 : 0;
        for (int i = a->nblk; i > at; i--) { a->blk[i] = a->blk[i - 1]; a->bsz[i] = a->bsz[i - 1]; }
        a->blk[at] = (char *)malloc(sz); a->bsz[at] = sz; a->nblk++;
        a->cur = at; a->off = 0;
    }
}
static void *ar_zalloc(arena_t *a, size_t n) { void *p = ar_alloc(a, n); memset(p, 0, n); return p; }
static void *ar_grow(arena_t *a, void *old, size_t old_bytes, size_t new_bytes)
{
    void *p = ar_alloc(a, new_bytes);
    if (old && old_bytes) memcpy(p, old, old_bytes);
    return p;
}
static void ar_reset(arena_t *a) { a->cur = 0; a->off = 0; }
#define LF_MAX_LANES 32
extern arena_t g_arena[LF_MAX_LANES][260];          /* [lane][worker]; blocks are kept across chunks and batches (lf_sched.c) */

/* mode 0: owned, growable, NUL-terminated; mode 1: count only (nothing is written); mode 2: fixed external window;
 * mode 3: growable inside an arena (never freed individually) */
typedef struct { char *s; size_t n, cap; int mode; arena_t *ar; } str_t;
static void str_init(str_t *b) { b->cap = 256; b->s = (char *)malloc(b->cap); b->n = 0; b->s[0] = 0; b->mode = 0; b->ar = NULL; }
static void str_init_ar(str_t *b, arena_t *ar, size_t cap0) { b->cap = cap0 < 64 ? 64 : cap0; b->s = (char *)ar_alloc(ar, b->cap); b->n = 0; b->s[0] = 0; b->mode = 3; b->ar = ar; }
static void str_room(str_t *b, size_t extra)
{
    if (b->mode == 1 || b->mode == 2 || b->n + extra + 1 <= b->cap) return;
    size_t nc = b->cap;
    while (b->n + extra + 1 > nc) nc *= 2;
    if (b->mode == 3) b->s = (char *)ar_grow(b->ar, b->s, b->n + 1, nc); else b->s = (char *)realloc(b->s, nc);
    b->cap = nc;
}
static inline void str_putn(str_t *b, const char *s, size_t l)
{
    str_room(b, l);
    if (b->mode != 1) memcpy(b->s + b->n, s, l);
    b->n += l;
    if (b->mode == 0 || b->mode == 3) b->s[b->n] = 0;
}
static void str_puts(str_t *b, const char *s) { str_putn(b, s, strlen(s)); }
static inline void str_putc(str_t *b, char c) { str_putn(b, &c, 1); }
static void str_putu(str_t *b, unsigned long long v)
{
    char tmp[24]; int k = 0;
    do { tmp[23 - k++] = (char)('0' + v % 10); v /= 10; } while (v);
    str_putn(b, tmp + 24 - k, (size_t)k);
}
static void str_puti(str_t *b, long long v) { if (v < 0) { str_putc(b, '-'); str_putu(b, (unsigned long long)(-v)); } else str_putu(b, (unsigned long long)v); }

/* per-base op track that grows at both ends (the reference uses std::deque<char>) */
typedef struct { char *buf; size_t cap, beg, end; } track_t;
static void tr_init(track_t *d, size_t hint) { d->cap = hint * 2 + 256; d->buf = (char *)malloc(d->cap); d->beg = d->end = d->cap / 2; }
static void tr_clear(track_t *d) { d->beg = d->end = d->cap / 2; }
static size_t tr_size(const track_t *d) { return d->end - d->beg; }
static void tr_room(track_t *d, size_t front, size_t back)
{
    if (d->beg >= front && d->cap - d->end >= back) return;
    size_t n = tr_size(d), ncap = (n + front + back) * 2 + 1024;
    char *nb = (char *)malloc(ncap);
    size_t nbeg = front + (ncap - n - front - back) / 2;
    memcpy(nb + nbeg, d->buf + d->beg, n);
    free(d->buf);
    d->buf = nb; d->cap = ncap; d->beg = nbeg; d->end = nbeg + n;
}
static void tr_back_n(track_t *d, size_t n, char c) { tr_room(d, 0, n); memset(d->buf + d->end, c, n); d->end += n; }
static void tr_front_n(track_t *d, size_t n, char c) { tr_room(d, n, 0); d->beg -= n; memset(d->buf + d->beg, c, n); }

/* ---------------------------------------------------------------- data model (src/LordFAST.h:43-118) */
typedef struct { uint32_t tStart, tEnd; uint8_t isReverse; float score; int req; } win_t;
#define WIN_LESS(a, b) ((a)->score > (b)->score)              /* compareWin, src/LordFAST.cpp:981-984 */
LF_DEFINE_STDSORT(winh, win_t, WIN_LESS)

typedef struct {
    uint32_t qStart, qEnd, pos, posEnd;
    uint16_t flag;
    int32_t alnScore, nmCount;
    char *cigar, *md;
    int rec, rtid;             /* >= 0: CIGAR / MD are rendered on the GPU (record `rec` of worker `rtid`) */
} sam_t;
typedef struct { sam_t *v; int n, cap; int32_t totalScore; } samlist_t;
#define SAM_LESS(a, b) ((a)->totalScore > (b)->totalScore)    /* compareSam, src/LordFAST.cpp:986-992 */
LF_DEFINE_STDSORT(samsort, samlist_t, SAM_LESS)

static void samlist_clear(samlist_t *l) { l->n = 0; }                 /* strings and the array live in an arena */
static void samlist_push(samlist_t *l, const sam_t *s, char *cigar, char *md, arena_t *ar)
{
    if (l->n == l->cap) { int nc = l->cap ? l->cap * 2 : 2; l->v = (sam_t *)ar_grow(ar, l->v, (size_t)l->cap * sizeof(sam_t), (size_t)nc * sizeof(sam_t)); l->cap = nc; }
    l->v[l->n] = *s; l->v[l->n].cigar = cigar; l->v[l->n].md = md; l->v[l->n].rec = -1; l->n++;
}

/* ---------------------------------------------------------------- requests */
typedef struct {           /* identity of one alignment request inside a chain walk */
    uint8_t type;          /* 0 edlib, 1 ksw */
    uint8_t qrc, trc;      /* sequence = reverse complement of the segment */
    uint8_t mode;          /* edlib: 0 NW 1 SHW ; ksw: 0 clip set, 1 split set */
    uint32_t qs, qseg, qn; /* query segment [qs, qs+qseg) of the walk's query string, first qn bases of it used */
    uint32_t ts, tseg, tn; /* reference segment [ts, ts+tseg), first tn bases used */
} rkey_t;

typedef struct {
    rkey_t key;
    int round;             /* -1 = requested, not yet computed */
    int64_t slot;          /* index in the round's result arrays */
    uint8_t *hops;         /* host copy of the ops region, fetched on demand (per-base fallback only) */
} memo_t;

typedef struct {
    int32_t *ed, *end; uint32_t *ops_len; uint64_t *ops_off; int n, pinned;
    uint8_t *ops;          /* host copy of the edit paths, or NULL when they stay in HBM ... */
    uint8_t *d_ops;        /* ... at this device address */
    void *d_desc;          /* the round's descriptors in HBM */
    int lazy;              /* its paths carry op 0 for every diagonal move (LF_F_LAZYX) */
    uint64_t ops_bytes;
} ed_round_t;
typedef struct { int32_t *score, *qle, *tle; int n; } ksw_round_t;

typedef struct {
    /* staged edlib requests of one worker */
    char *qb, *tb; uint64_t qn, qcap, tn, tcap;
    uint64_t *qoff, *toff; uint8_t *mode; int n, cap;
    memo_t **owner;        /* memo entry to patch */
    /* staged ksw requests */
    uint8_t *kq, *kt; uint64_t kqn, kqcap, ktn, ktcap;
    uint64_t *kqoff, *ktoff; int32_t *kprm; int kn, kcap;
    memo_t **kowner;
    uint64_t ext_bytes, blk_steps;
    /* staged edlib requests as descriptors into HBM-resident reads / pac (leaf-size problems: the common case) */
    lf_aln_desc_t *dd; uint64_t *dops; uintptr_t *downer; int dn, dcap; uint64_t dops_total;
    /* CIGAR / MD recipes of the finished records (rendered by lf_render.hip after the last round) */
    lf_ritem_t *ri; uint64_t rin, ricap; lf_rrecord_t *rr; int rrn, rrcap;
} stage_t;

typedef struct job {
    int read, widx;        /* owning read, slot in that read's mapping list */
    int req;               /* chain request whose chain this window is aligned with (its own, or -- clasp, window without seeds -- a stale one) */
    int isRev;
    Seed_t *chain; uint32_t chainLen;
    memo_t *memo; int nmemo, capmemo;
    int complete, hint;
} job_t;

typedef struct {
    const char *name, *seq, *qual;
    uint32_t len; int isFq;
    uint64_t src_off;      /* lf_map_batch_dev: where the read's bases (and qualities) are in the caller's device blobs; seq / qual == NULL until rd_host_bases */
    char *seq_rev, *qual_rev;
    Seed_t *F, *R; uint32_t nF, nR;
    int mode;              /* 0 short, 1 no window, 2 coarse, 3 fine */
    int vote_tid;          /* worker that voted this read (owns its chain requests) */
    int seed_idx;          /* position in the seed batch */
    /* fine-mode candidates in scan order */
    struct cand { uint32_t win; uint8_t isRev; int req; } *cands; int ncand, capcand;
    win_t *wins; int nWins;
    job_t *jobs;           /* one per kept window */
    samlist_t *maps;
    str_t out;
} rd_t;

typedef struct {           /* one chain request = (read, window) */
    int read; uint8_t isRev; uint32_t tStart, tEnd;
    uint64_t off; uint32_t n;
} creq_t;

typedef struct ctx {
    const struct lf_index *ix;
    const lf_params_t *p;
    int n_threads;
    rd_t *reads; int n_reads;
    /* chain requests (built per worker, then merged) */
    creq_t *creq; int n_creq;
    Seed_t *cseeds; uint64_t n_cseeds;
    uint32_t *chain_idx, *chain_len; float *chain_score;
    /* extension rounds */
    ed_round_t *ed_rounds; int n_ed_rounds;
    ksw_round_t *ksw_rounds; int n_ksw_rounds;
    stage_t *stages;       /* per worker */
    lf_stats_t *st;
    int lane;                       /* 0 / 1: which of the two in-flight chunks this is */
    arena_t *arena;                 /* g_arena[lane]: one per worker */
    struct cstage *cstage; struct jobvec *ed_jobs, *ksw_jobs, *edd_jobs;      /* per worker thread */
    const lfg_hits_t *hits;
    /* scratch for the parallel merge of staged alignment requests */
    lf_aln_desc_t *mg_desc; char *mg_qb, *mg_tb; uint64_t *mg_qoff, *mg_toff, *mg_qbase, *mg_tbase; uint8_t *mg_mode; int *mg_gbase; ed_round_t *mg_R; int mg_round;
    struct { const char *label; double t; } marks[96]; int n_marks; int timing;     /* LF_TIMING=1: per-chunk timeline */
    int lazy;                       /* paths leave the edlib kernels with unclassified diagonal moves (resolved by the renderer) */
    int host_vote;                  /* cross-check: vote / select / sort on the host from copied-back hits (cross-check) */
    lfg_vc_t vc;                    /* device path: modes, requests and chains of this chunk */
    uint64_t max_chunk_hits;        /* more seed hits than this in one chunk: map_chunk asks for a split (LF_RC_SPLIT) */
    int host_cigar;                 /* cross-check: build CIGAR / MD on the host from copied-back paths (cross-check) */
    char *rtext; uint64_t *roffs; int *rrbase;      /* rendered text, per-record offsets, first record of each worker */
    int n_dev_recs; uint64_t n_dev_items; void *d_dev_recs, *d_dev_items;       /* the device-planned recipe (lf_walk.hip): records 0 .. n_dev_recs-1 */
    int dev_sam; lfg_rtext_t rtext_dev; uint32_t *rlens; uint64_t sam_total; int sam_parity;       /* SAM lines assembled on the device (lf_sam.hip): text size of the chunk */
    /* output assembly */
    char *out_base; uint64_t *out_off;
    const char *const *len_seqs; uint32_t *len_out; volatile int len_bad;
    int *seed_map; char *cat; uint64_t *cat_off;
    uint64_t *pk_planes, pk_qw, *pk_xpos, pk_xcap, pk_xn; uint8_t *pk_xbyte; volatile int pk_overflow;      /* packed upload (phase_pack) */
    const struct lf_prepack *pre; int pre_i0;        /* the batch is prepacked (lf_batch.h): its planes, and the chunk's first read in the batch */
    const unsigned char *d_seqs, *d_quals;      /* lf_map_batch_dev: the caller's device blobs (NULL: host strings) */
    int32_t **stage_sink; int stage_i0;          /* lf_map_stages_batch: per read (batch index stage_i0 + ri) its decision, windows and alignWin results */
    /* HOLES mode (lf_sam.hip): the SEQ / QUAL column of every line is filled on the host from the caller's strings */
    int holes; struct fill *fill; int n_fill;
    /* the chains the device walk left to the host replay: the reads that still have an open job (NULL: every read is looked at, the cross-check paths) */
    int *open; int n_open; struct job **wj_owner; const void *wj_rec; volatile int n_rare;
} ctx_t;
typedef struct fill { uint64_t pos; const char *seq, *qual; uint32_t len; uint8_t rev, fq; } fill_t;

typedef void (*pf_fn)(ctx_t *cx, int tid, int i);
/* called by a lane driver (cx->lane); returns when every item ran (lf_sched.c) */
#define parallel_for(cx, n, fn) parallel_for_named(cx, n, fn, #fn)
void parallel_for_named(ctx_t *cx, int n, pf_fn fn, const char *name);
extern int g_phase_on;

/* ---------------------------------------------------------------- reference fetch (src/BWT.cpp:593-666) */
static inline int pac_base(const uint8_t *pac, uint32_t l) { return (pac[l >> 2] >> ((~l & 3) << 1)) & 3; }

static int pos2rid(const struct lf_index *ix, int64_t pos)
{   /* bns_pos2rid (lib/bwa/bntseq.c:349-363). pos >= l_pac is undefined in the reference (anns[-1],
       SURVEY App. B #9); we clamp to the last contig. */
    if (pos >= ix->l_pac) return ix->n_seqs - 1;
    int lo = 0, hi = ix->n_seqs - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (ix->contigs[mid].offset <= pos) lo = mid; else hi = mid - 1; }
    return lo;
}
static void chr_boundaries(const struct lf_index *ix, uint64_t beg, uint64_t end, uint32_t *cb, uint32_t *ce)
{   /* bwt_get_chr_boundaries: contig of the MIDPOINT */
    int rid = pos2rid(ix, (int64_t)((beg + end) >> 1));
    *cb = (uint32_t)ix->contigs[rid].offset;
    *ce = (uint32_t)(ix->contigs[rid].offset + ix->contigs[rid].len - 1);
}

/* reverse complement (tableRev, src/Common.cpp:31-40) -- lf_samdesc.c */
extern char g_rc_tab[256];
void rc_tab_init(void);
static inline char rc_char(char c) { return g_rc_tab[(unsigned char)c]; }
void rc_copy(char *d, const char *s, size_t l);
void rc_copy_stream(char *d, const char *s, size_t l);      /* the same bytes through non-temporal stores */
void lf_copy_stream(char *d, const char *s, size_t n);      /* memcpy through non-temporal stores */
int lf_pack_read(uint64_t *planes, uint64_t qw, uint64_t a, const char *seq, uint32_t len, uint64_t *exc_pos, uint8_t *exc_byte, uint64_t exc_cap, uint64_t *exc_n);
void revcomp_into(const char *s, char *out, uint32_t len);
void str_put_rc(str_t *b, const char *s, size_t l);
void str_put_rev(str_t *b, const char *s, size_t l);

/* job owner bookkeeping: parallel arrays */
typedef struct jobvec { job_t **job; int n, cap; } jobvec_t;
static void jv_push(jobvec_t *v, job_t *j) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->job = (job_t **)realloc(v->job, (size_t)v->cap * sizeof(job_t *)); } v->job[v->n++] = j; }


/* ---- across the files ---- */
void top_push(win_t *l, int *n, int maxWin, uint32_t i, uint32_t L, float score, int isRev, int req);      /* lf_pipeline.c: the heap of src/LordFAST.cpp:634-654 */
int  map_chunk(ctx_t *cx);                                                                                 /* lf_pipeline.c */
void chunk_free(ctx_t *cx);
extern const char *volatile g_lane_mark[LF_MAX_LANES];                                                     /* lf_sched.c: LF_WATCHDOG, the last stage mark of every lane */
void rd_host_bases(ctx_t *cx, rd_t *rd, arena_t *ar);                                                      /* lf_replay.c */
int  walk_chain(ctx_t *cx, int tid, job_t *job, samlist_t *map);
void score_mapping(const lf_params_t *p, samlist_t *map, int isReverse, uint32_t rLen, uint32_t chainLen);
void print_sam_entry(ctx_t *cx, rd_t *r, int num);                                                         /* lf_samdesc.c */
int  sam_stage_dev(ctx_t *cx);
void phase_fill(ctx_t *cx, int tid, int k);
/* lf_crosscheck.c is NOT part of liblfgpu.so: the host re-implementations of four device stages are a test library of their own (liblfxcheck.so, built beside
 * it) that lf_debug_crosscheck() loads on first use and that registers its entry points here */
typedef struct { int (*vote_chain)(ctx_t *cx); void (*bind_text)(ctx_t *cx, int tid, int ri); void (*sam_print)(ctx_t *cx, int tid, int ri); } lf_xc_hooks_t;
extern lf_xc_hooks_t g_xc;
void lf_xc_register(const lf_xc_hooks_t *h);
void phase_fine_select(ctx_t *cx, int tid, int ri);                                                        /* lf_pipeline.c */
void phase_make_jobs(ctx_t *cx, int tid, int ri);
/* lf_sched.c: the by-bases chunk cutter (exported so that tests/test_sched.py can call it without a device) */
int lf_cut_chunks_by_bases(const uint32_t *lens, int n, int n_lanes, double ramp, int sampling_count, int *ends, int cap);

#endif
