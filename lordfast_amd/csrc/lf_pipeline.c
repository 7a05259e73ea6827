/*
 * lf_pipeline.c -- lf_map_batch: mapSeq (src/LordFAST.cpp:461-580) for a whole read batch.
 *
 * The reference runs one read at a time through seed -> vote -> chain -> extend.  Here every stage is a
 * batch over all reads of a chunk so that each HIP kernel sees 10^5..10^7 independent work items:
 *
 *   A  GPU   seeds of every read                                   lf_seed.hip
 *   B  GPU   window vote (sparse, same counts/ties as the dense tagged array), local maxima, top-N heap,
 *            coarse/fine decision, seed selection + std::sort order per candidate window   lf_vote.hip
 *   C  GPU   dp-n2 chains of all candidate windows                 lf_chain_kernel.h
 *      host  fine mode: top-N heap replay on the chain scores
 *   D  GPU   alignments.  alignChain_edlib (src/LordFAST.cpp:1765-2258) is data dependent (clip test,
 *            split test), so it is written as a REPLAY: the host walks the reference's control flow,
 *            every alignment it needs is looked up in a memo; a miss registers a request and the walk
 *            continues speculatively on the common path.  Requests of all chains go to the GPU together
 *            (lf_align.hip), then incomplete chains are replayed.  ~99 % finish after one GPU round.
 *   D' GPU   CIGAR / MD text of every record from the paths left in HBM (src/LordFAST.cpp:1570-1763)   lf_render.hip
 *   E  host  MAPQ, SAM line assembly (src/LordFAST.cpp:318-459)
 *
 * The host never computes a DP cell: no CPU alignment, chaining or FM-index code exists in this library.
 * lf_debug_crosscheck() can switch stages B, D (walk), D' and E to host implementations that work on data copied back
 * from the device: diagnostic cross-checks for the tests, never selected automatically and not selectable by environment.
 */
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

void lf_sort_seeds_by_qpos(Seed_t *s, long n);

/* src/LordFAST.cpp:88-92 */
#define CLIP_LEN    500
#define CLIP_SIM    0.75
#define SPLIT_LEN   80
#define SPLIT_SIM   0.40
#define REVERSE_SIM 0.60

/* The host-side re-implementations of four device stages (vote / selection, chain walk, CIGAR / MD strings, SAM line
 * assembly) are kept as CROSS-CHECKS for the tests.  They are selected through lf_debug_crosscheck() only -- never by an
 * environment variable: nothing in a production environment can send a batch down a path that is ten times slower. */
#define LF_XC_HOST_VOTE  1u
#define LF_XC_HOST_CIGAR 2u
#define LF_XC_HOST_WALK  4u
#define LF_XC_HOST_SAM   8u
static volatile unsigned g_crosscheck = 0;
unsigned lf_debug_crosscheck(unsigned mask) { const unsigned old = g_crosscheck; g_crosscheck = mask & 15u; return old; }

#define LF_RC_SPLIT 100        /* internal: map_chunk wants its chunk cut in two (too many seed hits for one vote sort) */
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

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
static arena_t g_arena[LF_MAX_LANES][260];          /* [lane][worker]; blocks are kept across chunks and batches */

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
    const unsigned char *d_seqs, *d_quals;      /* lf_map_batch_dev: the caller's device blobs (NULL: host strings) */
    int32_t **stage_sink; int stage_i0;          /* lf_map_stages_batch: per read (batch index stage_i0 + ri) its decision, windows and alignWin results */
    /* HOLES mode (lf_sam.hip): the SEQ / QUAL column of every line is filled on the host from the caller's strings */
    int holes; struct fill *fill; int n_fill;
} ctx_t;
typedef struct fill { uint64_t pos; const char *seq, *qual; uint32_t len; uint8_t rev, fq; } fill_t;

/* ---------------------------------------------------------------- parallel for on a persistent thread pool
 * Up to eight chunks ("lanes") are in flight at once so that the host phases of one overlap the GPU phases of the others.
 * The pool therefore serves one job per lane concurrently; each lane's driver thread also works on its own job.
 * Worker ids: pool threads 0..nw-1, lane drivers nw..nw+lanes-1 (per-worker scratch arrays have nw+lanes entries). */
typedef void (*pf_fn)(ctx_t *cx, int tid, int i);
typedef struct { pf_fn fn; ctx_t *cx; int n, grain; volatile int next; int active, inflight; int timed; volatile long long cpu_ns; } pjob_t;
typedef struct {
    pthread_t th[256]; int nw, started, stop;
    pthread_mutex_t mu; pthread_cond_t cv_work, cv_done[LF_MAX_LANES];
    pjob_t job[LF_MAX_LANES];
} pool_t;
static pool_t g_pool = { .mu = PTHREAD_MUTEX_INITIALIZER, .cv_work = PTHREAD_COND_INITIALIZER,
                         .cv_done = { PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
                                      PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER } };

/* LF_TIMING=1: per-phase CPU time (summed over workers) and wall time, printed at the end of each batch */
static struct { const char *name; double cpu_ms, wall_ms; long calls; } g_phase[32];
static int g_phase_n, g_phase_on;
static pthread_mutex_t g_phase_mu = PTHREAD_MUTEX_INITIALIZER;
static inline long long thread_cpu_ns(void) { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1000000000LL + ts.tv_nsec; }
static void phase_account(const char *name, double cpu_ms, double wall_ms)
{
    pthread_mutex_lock(&g_phase_mu);
    int k = 0;
    for (; k < g_phase_n; k++) if (g_phase[k].name == name) break;
    if (k == g_phase_n && g_phase_n < 32) { g_phase[k].name = name; g_phase[k].cpu_ms = g_phase[k].wall_ms = 0; g_phase[k].calls = 0; g_phase_n++; }
    if (k < 32) { g_phase[k].cpu_ms += cpu_ms; g_phase[k].wall_ms += wall_ms; g_phase[k].calls++; }
    pthread_mutex_unlock(&g_phase_mu);
}
static void phase_report(void)
{
    for (int k = 0; k < g_phase_n; k++)
        fprintf(stderr, "[lf] phase %-20s cpu %9.1f ms  wall %8.1f ms  calls %ld\n", g_phase[k].name, g_phase[k].cpu_ms, g_phase[k].wall_ms, g_phase[k].calls);
    g_phase_n = 0;
}

static void pool_run(pjob_t *J, int tid)
{
    if (J->timed) {
        const long long c0 = thread_cpu_ns();
        for (;;) {
            int i = __sync_fetch_and_add(&J->next, J->grain);
            if (i >= J->n) break;
            int e = i + J->grain < J->n ? i + J->grain : J->n;
            for (; i < e; i++) J->fn(J->cx, tid, i);
        }
        __sync_fetch_and_add(&J->cpu_ns, thread_cpu_ns() - c0);
        return;
    }
    for (;;) {
        int i = __sync_fetch_and_add(&J->next, J->grain);
        if (i >= J->n) break;
        int e = i + J->grain < J->n ? i + J->grain : J->n;
        for (; i < e; i++) J->fn(J->cx, tid, i);
    }
}
static void *pool_worker(void *arg)
{
    pool_t *P = &g_pool;
    const int tid = (int)(intptr_t)arg;
    pthread_mutex_lock(&P->mu);
    for (;;) {
        int pick = -1;
        for (int k = 0; k < LF_MAX_LANES; k++) { const int j = (tid + k) % LF_MAX_LANES; if (P->job[j].active && P->job[j].next < P->job[j].n) { pick = j; break; } }
        if (pick >= 0) {
            pjob_t *J = &P->job[pick];
            J->inflight++;
            pthread_mutex_unlock(&P->mu);
            pool_run(J, tid);
            pthread_mutex_lock(&P->mu);
            if (--J->inflight == 0) pthread_cond_signal(&P->cv_done[pick]);
            continue;
        }
        if (P->stop) break;
        pthread_cond_wait(&P->cv_work, &P->mu);
    }
    pthread_mutex_unlock(&P->mu);
    return NULL;
}
static void pool_ensure(int nw)
{
    pool_t *P = &g_pool;
    if (P->started && P->nw == nw) return;
    if (P->started) {                                   /* worker count changed: restart the pool */
        pthread_mutex_lock(&P->mu); P->stop = 1; pthread_cond_broadcast(&P->cv_work); pthread_mutex_unlock(&P->mu);
        for (int t = 0; t < P->nw; t++) pthread_join(P->th[t], NULL);
        P->stop = 0; P->started = 0;
    }
    P->nw = nw;
    pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstacksize(&at, 4u << 20);
    for (int t = 0; t < nw; t++) pthread_create(&P->th[t], &at, pool_worker, (void *)(intptr_t)t);
    P->started = 1;
}
/* called by a lane driver (cx->lane); returns when every item ran */
#define parallel_for(cx, n, fn) parallel_for_named(cx, n, fn, #fn)
static void parallel_for_named(ctx_t *cx, int n, pf_fn fn, const char *name)
{
    pool_t *P = &g_pool;
    if (n <= 0) return;
    pjob_t *J = &P->job[cx->lane];
    const int self = P->nw + cx->lane;
    const double w0 = g_phase_on ? now_ms() : 0;
    pthread_mutex_lock(&P->mu);
    J->fn = fn; J->cx = cx; J->n = n; J->next = 0; J->timed = g_phase_on; J->cpu_ns = 0;
    J->grain = n / ((P->nw + 1) * 16) + 1; if (J->grain > 64) J->grain = 64;
    J->inflight = 1; J->active = 1;
    pthread_cond_broadcast(&P->cv_work);
    pthread_mutex_unlock(&P->mu);
    pool_run(J, self);
    pthread_mutex_lock(&P->mu);
    J->inflight--;
    while (J->inflight > 0) pthread_cond_wait(&P->cv_done[cx->lane], &P->mu);
    J->active = 0;
    pthread_mutex_unlock(&P->mu);
    if (g_phase_on) phase_account(name, J->cpu_ns / 1e6, now_ms() - w0);
}

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

/* tableRev, src/Common.cpp:31-40: case kept, anything else 'N' */
static char g_rc_tab[256]; static pthread_once_t g_rc_once = PTHREAD_ONCE_INIT;
static void rc_tab_init(void)
{
    memset(g_rc_tab, 'N', sizeof g_rc_tab);
    g_rc_tab['A'] = 'T'; g_rc_tab['C'] = 'G'; g_rc_tab['G'] = 'C'; g_rc_tab['T'] = 'A';
    g_rc_tab['a'] = 't'; g_rc_tab['c'] = 'g'; g_rc_tab['g'] = 'c'; g_rc_tab['t'] = 'a';
}
static inline char rc_char(char c) { return g_rc_tab[(unsigned char)c]; }
static void rc_copy(char *d, const char *s, size_t l);
static void revcomp_into(const char *s, char *out, uint32_t len) { rc_copy(out, s, len); out[len] = 0; }
/* reverse complement / reversed copy written straight into the SAM text (src/LordFAST.cpp:501-502 build both strings
 * for every read; only records on the reverse strand ever print them) */
/* reverse complement of l bytes: 16 at a time with two nibble-indexed byte shuffles where the CPU has SSSE3 (every x86-64
 * server of the last 15 years).  A, C, G, T and their lower-case forms differ from every other letter in (low nibble, bit 6,
 * bit 5): the complement comes out of one table indexed by the low nibble, 'N' everywhere else. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("ssse3"))) static void rc_copy_ssse3(char *d, const char *s, size_t l)
{
    /* low nibble -> complement (upper case) for A=0x41 C=0x43 G=0x47 T=0x54: nibbles 1, 3, 7, 4 */
    const __m128i tab = _mm_setr_epi8('N', 'T', 'N', 'G', 'A', 'N', 'N', 'C', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N');
    /* the byte a nibble must come from to be a base: 1 -> 'A', 3 -> 'C', 7 -> 'G', 4 -> 'T' (upper case) */
    const __m128i src = _mm_setr_epi8(0x20, 'A', 0x20, 'C', 'T', 0x20, 0x20, 'G', 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20);   /* 0x20: no upper-cased byte equals it */
    const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m128i lo4 = _mm_set1_epi8(0x0f), caseb = _mm_set1_epi8(0x20), up = _mm_set1_epi8((char)0xDF), enn = _mm_set1_epi8('N');
    size_t i = 0;
    for (; i + 16 <= l; i += 16) {
        __m128i x = _mm_loadu_si128((const __m128i *)(s + l - 16 - i));
        x = _mm_shuffle_epi8(x, rev);
        const __m128i cs = _mm_and_si128(x, caseb), xu = _mm_and_si128(x, up), nib = _mm_and_si128(x, lo4);
        const __m128i ok = _mm_cmpeq_epi8(_mm_shuffle_epi8(src, nib), xu);           /* really one of ACGT / acgt */
        __m128i c = _mm_or_si128(_mm_shuffle_epi8(tab, nib), cs);                       /* complement, case kept */
        c = _mm_or_si128(_mm_and_si128(ok, c), _mm_andnot_si128(ok, enn));
        _mm_storeu_si128((__m128i *)(d + i), c);
    }
    for (; i < l; i++) d[i] = rc_char(s[l - 1 - i]);
}
#endif
static void rc_copy(char *d, const char *s, size_t l)
{
#if defined(__x86_64__)
    static int have = -1;
    if (have < 0) have = __builtin_cpu_supports("ssse3") ? 1 : 0;
    if (have) { rc_copy_ssse3(d, s, l); return; }
#endif
    for (size_t i = 0; i < l; i++) d[i] = rc_char(s[l - 1 - i]);
}
static void str_put_rc(str_t *b, const char *s, size_t l)
{
    str_room(b, l);
    if (b->mode != 1) rc_copy(b->s + b->n, s, l);
    b->n += l;
    if (b->mode == 0 || b->mode == 3) b->s[b->n] = 0;
}
static void str_put_rev(str_t *b, const char *s, size_t l)
{
    str_room(b, l);
    if (b->mode != 1) { char *d = b->s + b->n; for (size_t i = 0; i < l; i++) d[i] = s[l - 1 - i]; }
    b->n += l;
    if (b->mode == 0 || b->mode == 3) b->s[b->n] = 0;
}

/* ================================================================ B: vote, candidates, selection */
typedef struct { uint32_t win, cnt; } wc_t;

/* LSD radix sort of (window, weight) pairs by window id, 11 bits per pass */
static wc_t *radix_sort_wc(wc_t *a, wc_t *tmp, int n, uint32_t maxkey)
{
    for (int shift = 0; shift < 32 && (maxkey >> shift); shift += 11) {
        uint32_t cnt[2049]; memset(cnt, 0, sizeof cnt);
        for (int i = 0; i < n; i++) cnt[((a[i].win >> shift) & 2047) + 1]++;
        for (int i = 1; i <= 2048; i++) cnt[i] += cnt[i - 1];
        for (int i = 0; i < n; i++) tmp[cnt[(a[i].win >> shift) & 2047]++] = a[i];
        wc_t *t = a; a = tmp; tmp = t;
    }
    return a;
}

/* sparse equivalent of the tagged dense array of src/LordFAST.cpp:588-620: every seed adds its weight to
 * windows floor(tPos/L) and floor(tPos/L)-1; returns the touched windows in ascending order in *out */
static int vote(const lf_params_t *p, uint32_t L, const Seed_t *s, uint32_t n, wc_t **buf, size_t *cap, wc_t **out)
{
    if (*cap < 4 * (size_t)n + 4) { *cap = 4 * (size_t)n + 4; *buf = (wc_t *)realloc(*buf, *cap * sizeof(wc_t)); }
    wc_t *w = *buf; int m = 0; uint32_t mx = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t id = s[i].tPos / L;
        uint32_t weight = (uint32_t)(1 + ((int32_t)s[i].len - p->min_anchor_len));
        if (id > mx) mx = id;
        w[m].win = id; w[m].cnt = weight; m++;
        if (id >= 1) { w[m].win = id - 1; w[m].cnt = weight; m++; }
    }
    w = radix_sort_wc(w, *buf + 2 * (size_t)n + 2, m, mx ? mx : 1);
    int d = 0;
    for (int i = 0; i < m; ) {
        uint32_t id = w[i].win, c = 0;
        while (i < m && w[i].win == id) c += w[i++].cnt;
        w[d].win = id; w[d].cnt = c; d++;
    }
    *out = w;
    return d;
}

/* local maximum test of src/LordFAST.cpp:630-632 on the sparse list (k = position of the window) */
static inline int local_max(const wc_t *w, int d, int k, uint32_t refWinNum)
{
    const uint32_t id = w[k].win;
    const int left_ok = (id == 0) || !(k > 0 && w[k - 1].win == id - 1) || w[k].cnt >= w[k - 1].cnt;
    const int right_ok = (id == refWinNum - 1) || !(k + 1 < d && w[k + 1].win == id + 1) || w[k].cnt > w[k + 1].cnt;
    return left_ok && right_ok;
}

static void top_push(win_t *l, int *n, int maxWin, uint32_t i, uint32_t L, float score, int isRev, int req)
{   /* src/LordFAST.cpp:634-654 */
    if (*n < maxWin) {
        win_t *b = &l[*n];
        b->tStart = i * L; b->tEnd = (i + 2) * L - 1; b->score = score; b->isReverse = (uint8_t)isRev; b->req = req;
        (*n)++;
        winh_push_heap(l, *n);
    } else if (score > l[0].score) {
        winh_pop_heap(l, *n);
        win_t *b = &l[*n - 1];
        b->tStart = i * L; b->tEnd = (i + 2) * L - 1; b->score = score; b->isReverse = (uint8_t)isRev; b->req = req;
        winh_push_heap(l, *n);
    }
}

typedef struct cstage { creq_t *v; int n, cap; Seed_t *s; uint64_t ns, caps; wc_t *wbuf; size_t wcap; wc_t *wbuf2; size_t wcap2; } cstage_t;

/* selection of src/LordFAST.cpp:995-1018 (== :659-680) into the worker's chain-request stage */
static int add_chain_request(ctx_t *cx, int tid, int ri, int isRev, uint32_t tStart, uint32_t tEnd)
{
    cstage_t *cs = &cx->cstage[tid];
    const rd_t *r = &cx->reads[ri];
    const uint32_t L = r->len, margin = L >> 1;
    uint32_t cb, ce;
    chr_boundaries(cx->ix, tStart, tEnd, &cb, &ce);
    const int64_t lo = ((int64_t)tStart - (int64_t)margin > (int64_t)cb) ? (int64_t)tStart - (int64_t)margin : (int64_t)cb;
    const int64_t hi = ((int64_t)tEnd + (int64_t)margin < (int64_t)ce) ? (int64_t)tEnd + (int64_t)margin : (int64_t)ce;
    const Seed_t *s = isRev ? r->R : r->F;
    const uint32_t n = isRev ? r->nR : r->nF;
    if (cs->ns + n + 1 > cs->caps) { cs->caps = (cs->ns + n + 1) * 2; cs->s = (Seed_t *)realloc(cs->s, cs->caps * sizeof(Seed_t)); }
    const uint64_t off = cs->ns;
    for (uint32_t i = 0; i < n; i++)
        if ((int64_t)s[i].tPos >= lo && (int64_t)s[i].tPos <= hi) cs->s[cs->ns++] = s[i];
    lf_sort_seeds_by_qpos(cs->s + off, (long)(cs->ns - off));               /* std::sort, src/Chain.cpp:244 */
    if (cs->n == cs->cap) { cs->cap = cs->cap ? cs->cap * 2 : 256; cs->v = (creq_t *)realloc(cs->v, (size_t)cs->cap * sizeof(creq_t)); }
    creq_t *q = &cs->v[cs->n];
    q->read = ri; q->isRev = (uint8_t)isRev; q->tStart = tStart; q->tEnd = tEnd; q->off = off; q->n = (uint32_t)(cs->ns - off);
    return cs->n++;         /* worker-local id; rebased after the merge */
}

static void phase_vote(ctx_t *cx, int tid, int ri)
{
    rd_t *r = &cx->reads[ri];
    const lf_params_t *p = cx->p;
    cstage_t *cs = &cx->cstage[tid];
    r->vote_tid = tid;
    if ((int)r->len < p->min_read_len) { r->mode = 0; return; }
    {   /* this read's hits -> the two SeedLists of the reference (forward / reverse), order kept */
        const lfg_hits_t *h = cx->hits;
        const uint64_t a = h->read_off[r->seed_idx], b = h->read_off[r->seed_idx + 1];
        uint32_t nr = 0;
        for (uint64_t j = a; j < b; j++) nr += h->strand[j];
        r->nR = nr; r->nF = (uint32_t)(b - a) - nr;
        r->F = (Seed_t *)ar_alloc(&cx->arena[tid], ((size_t)(b - a) + 2) * sizeof(Seed_t)); r->R = r->F + r->nF + 1;
        uint32_t f = 0, v = 0;
        for (uint64_t j = a; j < b; j++) {
            Seed_t sd; sd.tPos = h->tpos[j]; sd.qPos = h->qpl[j] & 0xFFFFF; sd.len = h->qpl[j] >> 20;
            if (h->strand[j]) r->R[v++] = sd; else r->F[f++] = sd;
        }
    }
    const uint32_t L = r->len;
    const uint32_t refWinNum = (uint32_t)cx->ix->l_pac / (uint32_t)p->min_read_len;          /* src/LordFAST.cpp:130 */
    uint32_t lim = (uint32_t)cx->ix->l_pac / L + 2;                                          /* :622-624 */
    if (lim > refWinNum) lim = refWinNum;
    const int maxWin = p->max_map;
    r->wins = (win_t *)ar_zalloc(&cx->arena[tid], ((size_t)maxWin + 1) * sizeof(win_t));
    r->nWins = 0;
    wc_t *wF, *wR;
    int dF = vote(p, L, r->F, r->nF, &cs->wbuf, &cs->wcap, &wF);
    for (int k = 0; k < dF && wF[k].win < lim; k++)
        if (local_max(wF, dF, k, refWinNum)) top_push(r->wins, &r->nWins, maxWin, wF[k].win, L, (float)wF[k].cnt, 0, -1);
    int dR = vote(p, L, r->R, r->nR, &cs->wbuf2, &cs->wcap2, &wR);
    for (int k = 0; k < dR && wR[k].win < lim; k++)
        if (local_max(wR, dR, k, refWinNum)) top_push(r->wins, &r->nWins, maxWin, wR[k].win, L, (float)wR[k].cnt, 1, -1);
    if (r->nWins == 0) { r->mode = 1; return; }
    winh_sort_heap(r->wins, r->nWins);                                                        /* :528 */
    const float scoreRatio = 4;
    /* a single candidate is compared with a stale slot in the reference (App. B #1); both branches then
     * align the same window and print the same record */
    if (r->nWins == 1 || r->wins[0].score >= scoreRatio * r->wins[1].score) {
        r->mode = 2;
        r->nWins = 1;
        r->wins[0].req = add_chain_request(cx, tid, ri, r->wins[0].isReverse, r->wins[0].tStart, r->wins[0].tEnd);
    } else {
        r->mode = 3;
        const float minScore = (float)r->wins[0].score / scoreRatio;                          /* :553 */
        r->nWins = 0;
        for (int pass = 0; pass < 2; pass++) {
            const wc_t *w = pass ? wR : wF; const int d = pass ? dR : dF;
            for (int k = 0; k < d && w[k].win < lim; k++) {
                if ((float)w[k].cnt > minScore && local_max(w, d, k, refWinNum)) {                /* :875-877 */
                    if (r->ncand == r->capcand) { int nc = r->capcand ? r->capcand * 2 : 8; r->cands = (struct cand *)ar_grow(&cx->arena[tid], r->cands, (size_t)r->capcand * sizeof(struct cand), (size_t)nc * sizeof(struct cand)); r->capcand = nc; }
                    r->cands[r->ncand].win = w[k].win; r->cands[r->ncand].isRev = (uint8_t)pass;
                    r->cands[r->ncand].req = add_chain_request(cx, tid, ri, pass, w[k].win * L, (w[k].win + 2) * L - 1);
                    r->ncand++;
                }
            }
        }
    }
}

/* device path: the read's mode, windows and fine-mode candidates as lf_vote_select_kernel decided them */
static void phase_select(ctx_t *cx, int tid, int ri)
{
    rd_t *r = &cx->reads[ri];
    const lf_params_t *p = cx->p;
    r->vote_tid = tid;
    if ((int)r->len < p->min_read_len) { r->mode = 0; return; }
    const lfg_vc_t *vc = &cx->vc;
    const int k = r->seed_idx;
    const uint32_t L = r->len;
    r->wins = (win_t *)ar_zalloc(&cx->arena[tid], ((size_t)p->max_map + 1) * sizeof(win_t));
    r->nWins = 0;
    r->mode = vc->mode[k];
    if (r->mode == 2) {
        const int rq = (int)vc->req0[k];
        const uint32_t w = vc->req_win[rq], id = w & 0x7fffffffu;
        win_t *b = &r->wins[0];
        b->tStart = id * L; b->tEnd = (id + 2) * L - 1; b->score = vc->vscore[k]; b->isReverse = (uint8_t)(w >> 31); b->req = rq;
        r->nWins = 1;
    } else if (r->mode == 3) {
        const uint32_t nc = vc->nreq[k];
        r->cands = (struct cand *)ar_alloc(&cx->arena[tid], ((size_t)nc + 1) * sizeof(struct cand));
        r->ncand = (int)nc; r->capcand = (int)nc + 1;
        for (uint32_t c = 0; c < nc; c++) {
            const int rq = (int)(vc->req0[k] + c);
            const uint32_t w = vc->req_win[rq];
            r->cands[c].win = w & 0x7fffffffu; r->cands[c].isRev = (uint8_t)(w >> 31); r->cands[c].req = rq;
        }
    }
}

/* fine mode, after the chain kernel: the heap of src/LordFAST.cpp:879-901 replayed on the chain scores */
static void phase_fine_select(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (r->mode != 3) return;
    for (int c = 0; c < r->ncand; c++)
        top_push(r->wins, &r->nWins, cx->p->max_map, r->cands[c].win, r->len, cx->chain_score[r->cands[c].req], r->cands[c].isRev, r->cands[c].req);
}

/* ================================================================ D: alignChain_edlib as a replay */
typedef struct {
    int ed, end; const uint8_t *ops; uint32_t nops;
    int have;
    int round; uint64_t ops_begin; uint32_t tcons;      /* where the path lives in HBM; reference bases it consumes */
    uint32_t slot, qn; int lazy;
} edres_t;

typedef struct {
    ctx_t *cx; int tid; job_t *job;
    const char *query;        /* read forward or reverse complement */
    uint32_t readLen;
    int missing;              /* edlib results still to come */
    int bail;                 /* a ksw result is missing: stop walking */
    int build;                /* build strings (results complete so far) */
} walk_t;

static int key_eq(const rkey_t *a, const rkey_t *b) { return memcmp(a, b, sizeof(rkey_t)) == 0; }

static memo_t *memo_find(job_t *j, const rkey_t *k)
{
    if (j->hint < j->nmemo && key_eq(&j->memo[j->hint].key, k)) return &j->memo[j->hint++];
    for (int i = 0; i < j->nmemo; i++) if (key_eq(&j->memo[i].key, k)) { j->hint = i + 1; return &j->memo[i]; }
    return NULL;
}
static memo_t *memo_add(job_t *j, const rkey_t *k, arena_t *ar)
{
    if (j->nmemo == j->capmemo) { int nc = j->capmemo ? j->capmemo * 2 : 32; j->memo = (memo_t *)ar_grow(ar, j->memo, (size_t)j->capmemo * sizeof(memo_t), (size_t)nc * sizeof(memo_t)); j->capmemo = nc; }
    memo_t *m = &j->memo[j->nmemo++];
    m->key = *k; m->round = -1; m->slot = -1; m->hops = NULL;
    j->hint = j->nmemo;
    return m;
}

/* lf_map_batch_dev: the host sees a read's bases only where it has to (reads shorter than -l, the replay of the rare
 * chains, entries printed on the host): one small D2H copy on demand.  A read is touched by one worker at a time. */
static void rd_host_bases(ctx_t *cx, rd_t *rd, arena_t *ar)
{
    if (rd->seq || !cx->d_seqs) return;
    char *b = (char *)ar_alloc(ar, ((size_t)rd->len + 1) * (rd->isFq ? 2 : 1));
    if (rd->len && lfg_fetch(cx->ix->device, b, cx->d_seqs + rd->src_off, rd->len) != LF_OK) memset(b, 'N', rd->len);
    b[rd->len] = 0;
    if (rd->isFq) {
        char *q = b + rd->len + 1;
        if (rd->len && lfg_fetch(cx->ix->device, q, cx->d_quals + rd->src_off, rd->len) != LF_OK) memset(q, '!', rd->len);
        q[rd->len] = 0; rd->qual = q;
    }
    rd->seq = b;
}

/* the walk's query string; the reverse complement of a read is only materialised if a byte-string request needs it */
static const char *walk_query(walk_t *w)
{
    if (!w->query) {
        rd_t *rd = &w->cx->reads[w->job->read];
        rd_host_bases(w->cx, rd, &w->cx->arena[w->tid]);
        if (!rd->seq_rev) { rd->seq_rev = (char *)ar_alloc(&w->cx->arena[w->tid], (size_t)rd->len + 1); revcomp_into(rd->seq, rd->seq_rev, rd->len); }
        w->query = rd->seq_rev;
    }
    return w->query;
}
/* bytes of a request: query segment of the walk's query string (optionally reverse-complemented) */
static void put_query(walk_t *w, const rkey_t *k, char *dst)
{
    const char *src = walk_query(w) + k->qs;
    if (!k->qrc) memcpy(dst, src, k->qn);
    else for (uint32_t i = 0; i < k->qn; i++) dst[i] = rc_char(src[k->qseg - 1 - i]);
}
static void put_target(const walk_t *w, const rkey_t *k, char *dst)
{
    const uint8_t *pac = w->cx->ix->pac;
    if (!k->trc) for (uint32_t i = 0; i < k->tn; i++) dst[i] = "ACGT"[pac_base(pac, k->ts + i)];
    else for (uint32_t i = 0; i < k->tn; i++) dst[i] = "ACGT"[3 - pac_base(pac, k->ts + k->tseg - 1 - i)];
}
static uint8_t code_of(char c)
{   /* _pf_char2int, src/LordFAST.cpp:158-164 */
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

/* job owner bookkeeping: parallel arrays */
typedef struct jobvec { job_t **job; int n, cap; } jobvec_t;
static void jv_push(jobvec_t *v, job_t *j) { if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->job = (job_t **)realloc(v->job, (size_t)v->cap * sizeof(job_t *)); } v->job[v->n++] = j; }

static void stage_ksw(walk_t *w, memo_t *m)
{
    stage_t *s = &w->cx->stages[w->tid];
    const rkey_t *k = &m->key;
    if (s->kn == s->kcap) {
        s->kcap = s->kcap ? s->kcap * 2 : 64;
        s->kqoff = (uint64_t *)realloc(s->kqoff, ((size_t)s->kcap + 1) * 8); s->ktoff = (uint64_t *)realloc(s->ktoff, ((size_t)s->kcap + 1) * 8);
        s->kprm = (int32_t *)realloc(s->kprm, (size_t)s->kcap * 7 * 4); s->kowner = (memo_t **)realloc(s->kowner, (size_t)s->kcap * sizeof(memo_t *));
        if (s->kn == 0) { s->kqoff[0] = 0; s->ktoff[0] = 0; }
    }
    if (s->kqn + k->qn + 1 > s->kqcap) { s->kqcap = (s->kqn + k->qn + 1) * 2; s->kq = (uint8_t *)realloc(s->kq, s->kqcap); }
    if (s->ktn + k->tn + 1 > s->ktcap) { s->ktcap = (s->ktn + k->tn + 1) * 2; s->kt = (uint8_t *)realloc(s->kt, s->ktcap); }
    /* codes: convertChar2int / reverseComplementIntStr (src/LordFAST.cpp:1191-1201): 3 - code, so an N (4)
     * becomes 255 in the reference and indexes past its 5x5 matrix; we score any code > 3 as 0 */
    const char *src = walk_query(w) + k->qs;
    for (uint32_t i = 0; i < k->qn; i++) {
        uint8_t c = k->qrc ? code_of(src[k->qseg - 1 - i]) : code_of(src[i]);
        s->kq[s->kqn + i] = k->qrc ? (uint8_t)(c > 3 ? 4 : 3 - c) : c;
    }
    const uint8_t *pac = w->cx->ix->pac;
    for (uint32_t i = 0; i < k->tn; i++)
        s->kt[s->ktn + i] = (uint8_t)(k->trc ? 3 - pac_base(pac, k->ts + k->tseg - 1 - i) : pac_base(pac, k->ts + i));
    s->kqn += k->qn; s->ktn += k->tn;
    s->kqoff[s->kn + 1] = s->kqn; s->ktoff[s->kn + 1] = s->ktn;
    int32_t *pr = s->kprm + 7 * s->kn;
    if (k->mode == 0) { pr[0] = 0; pr[1] = 1; pr[2] = 0; pr[3] = 1; pr[4] = 40; pr[5] = 40; }        /* ksw_extend :1848,:2180 */
    else { pr[0] = 8; pr[1] = 1; pr[2] = 4; pr[3] = 1; pr[4] = 100; pr[5] = 200; }                    /* ksw_extend2 :1971,:1981 */
    pr[6] = (int32_t)k->qn;                                                                           /* h0 = readAlnLen */
    s->kowner[s->kn] = (memo_t *)(uintptr_t)(((uint64_t)(uintptr_t)(m - w->job->memo)));
    s->kn++;
}

/* request -> descriptor (no bytes are copied: the GPU reads the resident read batch and the 2-bit reference) */
static void stage_edlib_desc(walk_t *w, memo_t *m)
{
    stage_t *s = &w->cx->stages[w->tid];
    const rkey_t *k = &m->key;
    const rd_t *rd = &w->cx->reads[w->job->read];
    if (s->dn == s->dcap) {
        s->dcap = s->dcap ? s->dcap * 2 : 1024;
        s->dd = (lf_aln_desc_t *)realloc(s->dd, (size_t)s->dcap * sizeof(lf_aln_desc_t));
        s->dops = (uint64_t *)realloc(s->dops, (size_t)s->dcap * 8);
        s->downer = (uintptr_t *)realloc(s->downer, (size_t)s->dcap * sizeof(uintptr_t));
    }
    lf_aln_desc_t *d = &s->dd[s->dn];
    memset(d, 0, sizeof *d);
    const int64_t roff = (int64_t)w->cx->cat_off[rd->seed_idx], L = (int64_t)rd->len;
    /* the walk's query string is the read (forward chains) or its reverse complement (reverse chains);
     * a request may itself ask for the reverse complement of a segment: compose into (start, direction, complement).
     * complementing twice is the identity for every byte that can match the upper-case reference */
    const int rev_q = (w->job->isRev ? 1 : 0) ^ (k->qrc ? 1 : 0);
    int64_t qstart;
    if (!w->job->isRev) qstart = k->qrc ? roff + k->qs + k->qseg - 1 : roff + k->qs;
    else qstart = k->qrc ? roff + L - k->qs - k->qseg : roff + L - 1 - k->qs;
    d->qstart = qstart;
    d->tstart = k->trc ? (int64_t)k->ts + k->tseg - 1 : (int64_t)k->ts;
    d->n = k->qn; d->m = k->tn; d->mode = k->mode;
    d->flags = (uint8_t)((rev_q ? (LF_F_QREV | LF_F_QCOMP) : 0) | (k->trc ? (LF_F_TREV | LF_F_TCOMP) : 0) | (w->cx->lazy ? LF_F_LAZYX : 0));
    s->dops[s->dn] = s->dops_total; s->dops_total += (uint64_t)k->qn + k->tn;
    s->downer[s->dn] = (uintptr_t)(m - w->job->memo);
    s->dn++;
    s->ext_bytes += (uint64_t)k->qn + (k->tn + 3) / 4 + k->qn + k->tn;      /* SURVEY 8(d) B_ext */
    s->blk_steps += (uint64_t)((k->qn + 63) / 64) * k->tn;
}

/* edlibAlign(query segment, target segment, mode, PATH) through the memo */
static edres_t need_edlib(walk_t *w, int qrc, uint32_t qs, uint32_t qseg, uint32_t qn, int trc, uint32_t ts, uint32_t tseg, uint32_t tn, int mode)
{
    edres_t r; memset(&r, 0, sizeof r);
    rkey_t k; memset(&k, 0, sizeof k);
    k.type = 0; k.qrc = (uint8_t)qrc; k.trc = (uint8_t)trc; k.mode = (uint8_t)mode;
    k.qs = qs; k.qseg = qseg; k.qn = qn; k.ts = ts; k.tseg = tseg; k.tn = tn;
    memo_t *m = memo_find(w->job, &k);
    if (!m) {
        m = memo_add(w->job, &k, &w->cx->arena[w->tid]);
        /* every request is a descriptor: problems above edlib's traceback switch are cut into leaves by the Hirschberg levels
         * on the device (lf_hirsch.hip, any query length), an empty side is a pure run written by the binning kernel */
        stage_edlib_desc(w, m); jv_push(&w->cx->edd_jobs[w->tid], w->job);
    }
    if (m->round < 0) { w->missing++; w->build = 0; r.ed = 0; r.end = (mode == 0) ? (int)tn - 1 : (int)tn - 1; return r; }
    const ed_round_t *R = &w->cx->ed_rounds[m->round];
    r.have = 1; r.ed = R->ed[m->slot]; r.end = R->end[m->slot]; r.nops = R->ops_len[m->slot];
    r.round = m->round; r.ops_begin = R->ops_off[m->slot] + ((uint64_t)qn + tn - r.nops);     /* end-aligned in its region */
    r.tcons = mode == 0 ? tn : (uint32_t)(r.end + 1);                               /* NW: all of it; SHW: up to the end column */
    r.slot = (uint32_t)m->slot; r.qn = qn; r.lazy = R->lazy;
    r.ops = m->hops ? m->hops + ((uint64_t)qn + tn - r.nops) : (R->ops ? R->ops + r.ops_begin : NULL);
    return r;
}

static int need_ksw(walk_t *w, int set, int qrc, uint32_t qs, uint32_t qseg, int trc, uint32_t ts, uint32_t tseg, int *qle, int *tle)
{
    rkey_t k; memset(&k, 0, sizeof k);
    k.type = 1; k.qrc = (uint8_t)qrc; k.trc = (uint8_t)trc; k.mode = (uint8_t)set;
    k.qs = qs; k.qseg = qseg; k.qn = qseg; k.ts = ts; k.tseg = tseg; k.tn = tseg;
    memo_t *m = memo_find(w->job, &k);
    if (!m) { m = memo_add(w->job, &k, &w->cx->arena[w->tid]); stage_ksw(w, m); jv_push(&w->cx->ksw_jobs[w->tid], w->job); }
    if (m->round < 0) { w->bail = 1; w->build = 0; return 0; }
    const ksw_round_t *R = &w->cx->ksw_rounds[m->round];
    *qle = R->qle[m->slot]; *tle = R->tle[m->slot];
    return 1;
}

/* ---- CIGAR / MD tracks (src/LordFAST.cpp:1570-1763) ---- */
static const char OP2CH[4] = { 'M', 'I', 'D', 'M' };
static void ops_back(track_t *cg, track_t *md, const edres_t *r, const uint8_t *pac, int trc, uint32_t ts, uint32_t tseg)
{   /* edlibCigar_pushback + edlibMD_pushback: target base for deletions / mismatches */
    if (!r->have) return;
    tr_room(cg, 0, r->nops); tr_room(md, 0, r->nops);
    uint32_t ti = 0;
    for (uint32_t i = 0; i < r->nops; i++) {
        const uint8_t op = r->ops[i];
        cg->buf[cg->end++] = OP2CH[op];
        char m;
        if (op == 0) { m = '='; ti++; }
        else if (op == 1) m = '-';
        else { m = "ACGT"[trc ? 3 - pac_base(pac, ts + tseg - 1 - ti) : pac_base(pac, ts + ti)]; ti++; }
        md->buf[md->end++] = m;
    }
}
static void ops_front(track_t *cg, track_t *md, const edres_t *r, const uint8_t *pac, uint32_t ts, uint32_t tseg)
{   /* edlibCigar_pushfront + edlibMD_pushfront: the alignment was computed on reverse complements of
       query prefix and reference [ts, ts+tseg); pushing each op to the front restores forward order and
       the MD base is the complement of the (reverse-complemented) target base = the forward base */
    if (!r->have) return;
    tr_room(cg, r->nops, 0); tr_room(md, r->nops, 0);
    uint32_t ti = 0;
    for (uint32_t i = 0; i < r->nops; i++) {
        const uint8_t op = r->ops[i];
        cg->buf[--cg->beg] = OP2CH[op];
        char m;
        if (op == 0) { m = '='; ti++; }
        else if (op == 1) m = '-';
        else { m = "ACGT"[pac_base(pac, ts + tseg - 1 - ti)]; ti++; }   /* complement(rc target[ti]) */
        md->buf[--md->beg] = m;
    }
}

static char *cigar_string(const track_t *c, arena_t *ar)
{   /* edlibCigar_toString: leading / trailing I runs print as S */
    str_t sb; str_init_ar(&sb, ar, 512);
    char ch = 0; unsigned num = 0; int opn = 0;
    const size_t n = tr_size(c);
    for (size_t i = 0; i < n; i++) {
        const char x = c->buf[c->beg + i];
        if (x != ch) {
            if (ch != 0) { str_putu(&sb, num); str_putc(&sb, (opn == 0 && ch == 'I') ? 'S' : ch); opn++; }
            num = 1; ch = x;
        } else num++;
    }
    if (num) { str_putu(&sb, num); str_putc(&sb, ch == 'I' ? 'S' : ch); }
    return sb.s;
}
static char *md_string(const track_t *md, const track_t *cg, arena_t *ar)
{   /* edlibMD_toString */
    str_t sb; str_init_ar(&sb, ar, 512);
    unsigned num = 0; char last = '=';
    const size_t n = tr_size(md);
    for (size_t i = 0; i < n; i++) {
        const char m = md->buf[md->beg + i], c = cg->buf[cg->beg + i];
        if (m == '=') { num++; last = '='; }
        else if (m == '-') last = 'I';
        else if (c == 'M') { str_putu(&sb, num); num = 0; str_putc(&sb, m); last = 'X'; }
        else if (c == 'D') { if (last != 'D') { str_putu(&sb, num); num = 0; str_putc(&sb, '^'); } str_putc(&sb, m); last = 'D'; }
    }
    str_putu(&sb, num);
    return sb.s;
}

/* ---- alignment builder: CIGAR + MD of one SAM record ------------------------------------------------
 * STREAM mode (default) run-length encodes on the fly: it is fed (cigar char, md char) pairs in final order and
 * keeps the state machines of edlibCigar_toString / edlibMD_toString (src/LordFAST.cpp:1596-1626,1717-1763), so no
 * per-base track is materialised.  Front insertions are legal only while nothing has been appended (that is how the
 * reference uses them: left extension first, or right after a clear); they are stacked and flushed first.
 * TRACK mode keeps the per-base deques of the reference and is used when a walk hits the one branch that misaligns
 * MD and CIGAR (src/LordFAST.cpp:2057, App. B #3). */
typedef struct { int kind; char cg, md; uint32_t n; edres_t r; uint32_t ts, tseg; } fseg_t;
typedef struct {
    int track_mode, need_track, active;    /* track_mode: 0 STREAM (host strings), 1 TRACK (per-base), 2 RECIPE (GPU renders) */
    stage_t *rs; uint64_t item_mark;       /* RECIPE mode: the worker's item list; first item of the open record */
    track_t cg, md;                        /* TRACK mode */
    str_t scg, smd; char ch; unsigned run; int opn; unsigned mdnum; char last; int fed;    /* STREAM mode */
    fseg_t front[8]; int nfront;
    const uint8_t *pac; arena_t *ar; size_t hint;
} alnb_t;

static void ab_reset_stream(alnb_t *b) { str_init_ar(&b->scg, b->ar, b->hint); str_init_ar(&b->smd, b->ar, b->hint); b->ch = 0; b->run = 0; b->opn = 0; b->mdnum = 0; b->last = '='; b->fed = 0; b->nfront = 0; }
static void ab_init(alnb_t *b, int track_mode, int active, const uint8_t *pac, size_t hint, arena_t *ar, stage_t *rs)
{
    memset(b, 0, sizeof *b);
    b->track_mode = track_mode; b->active = active; b->pac = pac; b->ar = ar; b->hint = hint / 2 + 128;
    if (!active) return;
    if (track_mode == 2) { b->rs = rs; b->item_mark = rs->rin; b->last = '='; }
    else if (track_mode) { tr_init(&b->cg, hint); tr_init(&b->md, hint); } else ab_reset_stream(b);
}
static void ab_free(alnb_t *b)
{
    if (!b->active) return;
    if (b->track_mode == 1) { free(b->cg.buf); free(b->md.buf); }
}
/* ---- RECIPE mode: the order of the pieces is all the host records ---- */
static inline lf_ritem_t *rc_item(alnb_t *b)
{
    stage_t *s = b->rs;
    if (s->rin == s->ricap) { s->ricap = s->ricap ? s->ricap * 2 : 4096; s->ri = (lf_ritem_t *)realloc(s->ri, s->ricap * sizeof(lf_ritem_t)); }
    lf_ritem_t *it = &s->ri[s->rin++];
    memset(it, 0, sizeof *it);
    return it;
}
static void rc_run(alnb_t *b, int kind, uint32_t n, uint32_t tpos)
{
    if (!n) return;
    b->fed = 1;
    lf_ritem_t *it = rc_item(b); it->kind = (uint8_t)kind; it->n = n; it->tpos = tpos;
}
static void rc_ops(alnb_t *b, const edres_t *r, int kind, uint32_t tpos)
{
    if (r->round >= LF_MAX_ED_ROUNDS) { b->need_track = 1; return; }      /* beyond the rounds kept in HBM: host fallback */
    if (!r->nops) return;
    b->fed = 1;
    lf_ritem_t *it = rc_item(b); it->kind = (uint8_t)kind; it->n = r->nops; it->tpos = tpos; it->round = (uint8_t)r->round; it->ops_begin = r->ops_begin;
    it->slot = r->slot; it->qn = r->qn; it->tcons = r->tcons; it->lazy = (uint8_t)r->lazy;
}
static inline void st_c(alnb_t *b, char c, uint32_t n)
{
    if (c != b->ch) {
        if (b->ch) { str_putu(&b->scg, b->run); str_putc(&b->scg, (b->opn == 0 && b->ch == 'I') ? 'S' : b->ch); b->opn++; }
        b->run = n; b->ch = c;
    } else b->run += n;
}
static inline void st_md_base(alnb_t *b, char base, int is_del)
{
    if (!is_del) { str_putu(&b->smd, b->mdnum); b->mdnum = 0; str_putc(&b->smd, base); b->last = 'X'; }
    else { if (b->last != 'D') { str_putu(&b->smd, b->mdnum); b->mdnum = 0; str_putc(&b->smd, '^'); } str_putc(&b->smd, base); b->last = 'D'; }
}
static void st_run(alnb_t *b, char cg, char md, uint32_t n)
{
    if (!n) return;
    b->fed = 1;
    st_c(b, cg, n);
    if (md == '=') { b->mdnum += n; b->last = '='; } else b->last = 'I';          /* md is '=' or '-' for runs */
}
/* ops in forward order (target base index grows) */
static void st_ops_fwd(alnb_t *b, const edres_t *r, int trc, uint32_t ts, uint32_t tseg)
{
    const uint8_t *pac = b->pac; uint32_t ti = 0;
    if (r->nops) b->fed = 1;
    for (uint32_t i = 0; i < r->nops; ) {
        const uint8_t op = r->ops[i];
        if (op == 0) { uint32_t j = i + 1; while (j < r->nops && r->ops[j] == 0) j++; st_c(b, 'M', j - i); b->mdnum += j - i; b->last = '='; ti += j - i; i = j; continue; }
        if (op == 1) { st_c(b, 'I', 1); b->last = 'I'; i++; continue; }
        const char base = "ACGT"[trc ? 3 - pac_base(pac, ts + tseg - 1 - ti) : pac_base(pac, ts + ti)];
        st_c(b, op == 2 ? 'D' : 'M', 1); st_md_base(b, base, op == 2); ti++; i++;
    }
}
/* ops that the reference pushes to the FRONT one by one (reversed order); target = reverse complement of
 * [ts, ts+tseg): the MD base is the complement of it, i.e. the forward base (edlibMD_pushfront) */
static void st_ops_rev(alnb_t *b, const edres_t *r, uint32_t ts, uint32_t tseg)
{
    const uint8_t *pac = b->pac;
    uint32_t ti = 0;
    for (uint32_t i = 0; i < r->nops; i++) ti += (r->ops[i] != 1);
    if (r->nops) b->fed = 1;
    for (uint32_t i = r->nops; i-- > 0; ) {
        const uint8_t op = r->ops[i];
        if (op == 0) { ti--; st_c(b, 'M', 1); b->mdnum++; b->last = '='; continue; }
        if (op == 1) { st_c(b, 'I', 1); b->last = 'I'; continue; }
        ti--;
        const char base = "ACGT"[pac_base(pac, ts + tseg - 1 - ti)];
        st_c(b, op == 2 ? 'D' : 'M', 1); st_md_base(b, base, op == 2);
    }
}
static void ab_flush_front(alnb_t *b)
{
    while (b->nfront > 0) {
        const fseg_t *f = &b->front[--b->nfront];
        if (b->track_mode == 2) {
            if (f->kind == 0) rc_run(b, f->cg == 'M' ? LF_RI_RUN_M : LF_RI_RUN_I, f->n, 0);
            else rc_ops(b, &f->r, LF_RI_OPS_REV, f->ts + f->tseg - f->r.tcons);
            continue;
        }
        if (f->kind == 0) st_run(b, f->cg, f->md, f->n); else st_ops_rev(b, &f->r, f->ts, f->tseg);
    }
}
static void ab_back_run(alnb_t *b, char cg, char md, size_t n)
{
    if (!b->active) return;
    if (b->track_mode == 1) { tr_back_n(&b->cg, n, cg); tr_back_n(&b->md, n, md); return; }
    ab_flush_front(b);
    if (b->track_mode == 2) { rc_run(b, cg == 'M' ? LF_RI_RUN_M : LF_RI_RUN_I, (uint32_t)n, 0); return; }
    st_run(b, cg, md, (uint32_t)n);
}
static void ab_front_run(alnb_t *b, char cg, char md, size_t n)
{
    if (!b->active) return;
    if (b->track_mode == 1) { tr_front_n(&b->cg, n, cg); tr_front_n(&b->md, n, md); return; }
    if (b->fed || b->nfront >= 8) { b->need_track = 1; return; }
    fseg_t *f = &b->front[b->nfront++]; f->kind = 0; f->cg = cg; f->md = md; f->n = (uint32_t)n;
}
static void ab_back_ops(alnb_t *b, const edres_t *r, int trc, uint32_t ts, uint32_t tseg)
{
    if (!b->active || !r->have) return;
    if (b->track_mode == 1) { ops_back(&b->cg, &b->md, r, b->pac, trc, ts, tseg); return; }
    ab_flush_front(b);
    if (b->track_mode == 2) { rc_ops(b, r, trc ? LF_RI_OPS_FWD_TRC : LF_RI_OPS_FWD, trc ? ts + tseg - 1 : ts); return; }
    st_ops_fwd(b, r, trc, ts, tseg);
}
static void ab_front_ops(alnb_t *b, const edres_t *r, uint32_t ts, uint32_t tseg)
{
    if (!b->active || !r->have) return;
    if (b->track_mode == 1) { ops_front(&b->cg, &b->md, r, b->pac, ts, tseg); return; }
    if (b->fed || b->nfront >= 8) { b->need_track = 1; return; }
    fseg_t *f = &b->front[b->nfront++]; f->kind = 1; f->r = *r; f->ts = ts; f->tseg = tseg;
}
static void ab_back_del(alnb_t *b, uint32_t ts, uint32_t n)
{   /* pure deletion between two anchors (src/LordFAST.cpp:2126-2134) */
    if (!b->active || !n) return;
    if (b->track_mode == 1) {
        tr_back_n(&b->cg, n, 'D'); tr_room(&b->md, 0, n);
        for (uint32_t j = 0; j < n; j++) b->md.buf[b->md.end++] = "ACGT"[pac_base(b->pac, ts + j)];
        return;
    }
    ab_flush_front(b); b->fed = 1;
    if (b->track_mode == 2) { rc_run(b, LF_RI_DEL, n, ts); return; }
    st_c(b, 'D', n);
    for (uint32_t j = 0; j < n; j++) st_md_base(b, "ACGT"[pac_base(b->pac, ts + j)], 1);
}
static void ab_md_front_only(alnb_t *b, size_t n)
{   /* the reference's misplaced padding: MD at the front while the CIGAR got it at the back */
    if (!b->active) return;
    if (b->track_mode == 1) { tr_front_n(&b->md, n, '-'); return; }
    b->need_track = 1;
}
static void ab_cg_back_only(alnb_t *b, size_t n)
{
    if (!b->active) return;
    if (b->track_mode == 1) { tr_back_n(&b->cg, n, 'I'); return; }
    b->need_track = 1;
}
static void ab_clear(alnb_t *b)
{
    if (!b->active) return;
    if (b->track_mode == 1) { tr_clear(&b->cg); tr_clear(&b->md); return; }
    if (b->track_mode == 2) { b->rs->rin = b->item_mark; b->fed = 0; b->nfront = 0; return; }
    ab_reset_stream(b);
}
/* strings of the record built so far (ownership passes to the caller) */
static void ab_take(alnb_t *b, char **cigar, char **md)
{
    if (b->track_mode == 1) { *cigar = cigar_string(&b->cg, b->ar); *md = md_string(&b->md, &b->cg, b->ar); return; }
    ab_flush_front(b);
    if (b->run) { str_putu(&b->scg, b->run); str_putc(&b->scg, b->ch == 'I' ? 'S' : b->ch); }
    str_putu(&b->smd, b->mdnum);
    *cigar = b->scg.s; *md = b->smd.s;
    ab_reset_stream(b);
}

static void emit_sam(walk_t *w, samlist_t *map, const sam_t *tmp, alnb_t *ab)
{
    if (!w->build || !ab->active || ab->need_track) return;
    if (ab->track_mode == 2) {                       /* close the record: its pieces are items [item_mark, rin) */
        ab_flush_front(ab);
        if (ab->need_track) return;
        stage_t *s = ab->rs;
        if (s->rrn == s->rrcap) { s->rrcap = s->rrcap ? s->rrcap * 2 : 1024; s->rr = (lf_rrecord_t *)realloc(s->rr, (size_t)s->rrcap * sizeof(lf_rrecord_t)); }
        s->rr[s->rrn].item0 = (uint32_t)ab->item_mark; s->rr[s->rrn].nitems = (uint32_t)(s->rin - ab->item_mark);
        samlist_push(map, tmp, NULL, NULL, ab->ar);
        map->v[map->n - 1].rec = s->rrn++; map->v[map->n - 1].rtid = w->tid;
        ab->item_mark = s->rin; ab->fed = 0; ab->nfront = 0;
        return;
    }
    char *c, *m;
    ab_take(ab, &c, &m);
    samlist_push(map, tmp, c, m, ab->ar);
}

/* the walk itself.  Returns 1 when every alignment it needed was available (map is then final). */
static int walk_chain_mode(ctx_t *cx, int tid, job_t *job, samlist_t *map, int track_mode, int active, int *need_track)
{
    const struct lf_index *ix = cx->ix;
    const uint8_t *pac = ix->pac;
    rd_t *rd = &cx->reads[job->read];
    const int isRev = job->isRev;
    const Seed_t *s = job->chain;
    const uint32_t chainLen = job->chainLen;
    walk_t W; memset(&W, 0, sizeof W);
    rd_host_bases(cx, rd, &cx->arena[tid]);
    W.cx = cx; W.tid = tid; W.job = job; W.query = isRev ? rd->seq_rev : rd->seq; W.readLen = rd->len; W.build = 1;
    job->hint = 0;
    const int32_t readLen = (int32_t)rd->len;
    stage_t *const rs = &cx->stages[tid];
    const uint64_t rin0 = rs->rin; const int rrn0 = rs->rrn;
    alnb_t ab; ab_init(&ab, track_mode, active, pac, rd->len, &cx->arena[tid], rs);
    sam_t tmp; memset(&tmp, 0, sizeof tmp);
    uint32_t chrBeg, chrEnd, readAlnStart, refAlnStart, readAlnEnd, refAlnEnd, i;
    int32_t readAlnLen, refAlnLen, editScore = 0;
    int qle = 0, tle = 0;
    samlist_clear(map);

    chr_boundaries(ix, s[0].tPos, s[chainLen - 1].tPos, &chrBeg, &chrEnd);                   /* :1799 */
    tmp.flag = isRev ? 16 : 0; tmp.pos = s[0].tPos; tmp.qStart = s[0].qPos;

    /* ---- before the first anchor (:1820-1899) ---- */
    readAlnLen = (int32_t)s[0].qPos; refAlnLen = readAlnLen + 20;
    if (readAlnLen > 0) {
        if ((int64_t)s[0].tPos - refAlnLen >= (int64_t)chrBeg) {
            refAlnStart = s[0].tPos - (uint32_t)refAlnLen;
            edres_t r = need_edlib(&W, 1, 0, (uint32_t)readAlnLen, (uint32_t)readAlnLen, 1, refAlnStart, (uint32_t)refAlnLen, (uint32_t)refAlnLen, 1);
            int realigned = 0;
            if (r.have && readAlnLen > CLIP_LEN && (1 - ((float)r.ed / readAlnLen)) < CLIP_SIM) {
                if (!need_ksw(&W, 0, 1, 0, (uint32_t)readAlnLen, 1, refAlnStart, (uint32_t)refAlnLen, &qle, &tle)) goto bail;
                if (qle > 0 && qle < readAlnLen) {
                    edres_t r2 = need_edlib(&W, 1, 0, (uint32_t)readAlnLen, (uint32_t)qle, 1, refAlnStart, (uint32_t)refAlnLen, (uint32_t)tle, 0);
                    ab_front_ops(&ab, &r2, refAlnStart, (uint32_t)refAlnLen);
                    editScore -= r2.ed;
                    tmp.pos = s[0].tPos - (uint32_t)r2.end - 1;
                    tmp.qStart = s[0].qPos - (uint32_t)qle;
                    ab_front_run(&ab, 'I', '-', (size_t)(readAlnLen - qle));
                    realigned = 1;
                }
            }
            if (!realigned) {
                editScore -= r.ed;
                ab_front_ops(&ab, &r, refAlnStart, (uint32_t)refAlnLen);
                tmp.pos = s[0].tPos - (uint32_t)r.end - 1;
                tmp.qStart = 0;
            }
        } else ab_front_run(&ab, 'I', '-', (size_t)readAlnLen);
    }

    /* ---- between adjacent anchors (:1901-2137) ---- */
    int numAnchorsSoFar = 1;
    for (i = 0; i + 1 < chainLen; i++) {
        ab_back_run(&ab, 'M', '=', s[i].len);
        readAlnStart = s[i].qPos + s[i].len; refAlnStart = s[i].tPos + s[i].len;
        readAlnEnd = s[i + 1].qPos; refAlnEnd = s[i + 1].tPos;
        readAlnLen = (int32_t)(readAlnEnd - readAlnStart); refAlnLen = (int32_t)(refAlnEnd - refAlnStart);
        if (readAlnLen > 0 && refAlnLen > 0) {
            edres_t r = need_edlib(&W, 0, readAlnStart, (uint32_t)readAlnLen, (uint32_t)readAlnLen, 0, refAlnStart, (uint32_t)refAlnLen, (uint32_t)refAlnLen, 0);
            int handled = 0;
            if (r.have && abs(readAlnLen - refAlnLen) >= SPLIT_LEN && (1 - ((float)r.ed / readAlnLen)) < SPLIT_SIM) {
                /* split test: extension from both ends of the gap (:1967-1983) */
                int q1, t1, q2, t2;
                if (!need_ksw(&W, 1, 0, readAlnStart, (uint32_t)readAlnLen, 0, refAlnStart, (uint32_t)refAlnLen, &q1, &t1)) goto bail;
                if (!need_ksw(&W, 1, 1, readAlnStart, (uint32_t)readAlnLen, 1, refAlnStart, (uint32_t)refAlnLen, &q2, &t2)) goto bail;
                const uint32_t rs_new = readAlnStart + (uint32_t)q1, ts_new = refAlnStart + (uint32_t)t1;
                const uint32_t re_new = readAlnEnd - (uint32_t)q2, te_new = refAlnEnd - (uint32_t)t2;
                const int32_t tl_new = (int32_t)(te_new - ts_new), rl_new = (int32_t)(re_new - rs_new);
                if (rs_new < re_new || ts_new < te_new) {                                    /* :1995 */
                    handled = 1;
                    if (rs_new > readAlnStart || ts_new > refAlnStart) {                     /* first part :1998-2007 */
                        edres_t a = need_edlib(&W, 0, readAlnStart, (uint32_t)readAlnLen, rs_new - readAlnStart, 0, refAlnStart, (uint32_t)refAlnLen, ts_new - refAlnStart, 0);
                        ab_back_ops(&ab, &a, 0, refAlnStart, (uint32_t)refAlnLen);
                        editScore -= a.ed;
                    }
                    ab_back_run(&ab, 'I', '-', (size_t)((uint32_t)readLen - rs_new));
                    tmp.posEnd = ts_new; tmp.qEnd = rs_new; tmp.nmCount = editScore;
                    if (numAnchorsSoFar > 1) emit_sam(&W, map, &tmp, &ab);
                    ab_clear(&ab); editScore = 0;
                    if (rs_new < re_new && ts_new < te_new) {                                /* middle part :2033-2077 */
                        edres_t f = need_edlib(&W, 0, rs_new, (uint32_t)rl_new, (uint32_t)rl_new, 0, ts_new, (uint32_t)tl_new, (uint32_t)tl_new, 0);
                        edres_t v = need_edlib(&W, 1, rs_new, (uint32_t)rl_new, (uint32_t)rl_new, 0, ts_new, (uint32_t)tl_new, (uint32_t)tl_new, 0);
                        if (f.have && v.have && (1 - ((double)v.ed / rl_new)) > (1 - ((double)f.ed / rl_new)) && (1 - ((double)v.ed / rl_new)) > REVERSE_SIM) {
                            tmp.flag = isRev ? 0 : 16;
                            tmp.pos = ts_new; tmp.qStart = rs_new; tmp.posEnd = te_new; tmp.qEnd = re_new;
                            ab_back_run(&ab, 'I', '-', rs_new);
                            ab_back_ops(&ab, &v, 0, ts_new, (uint32_t)tl_new);
                            editScore -= v.ed;
                            ab_cg_back_only(&ab, (size_t)((uint32_t)readLen - re_new));
                            ab_md_front_only(&ab, (size_t)((uint32_t)readLen - re_new));          /* sic :2057 (App. B #3) */
                            tmp.nmCount = editScore;
                            emit_sam(&W, map, &tmp, &ab);
                            ab_clear(&ab); editScore = 0;
                        }
                    }
                    if (re_new < readAlnEnd || te_new < refAlnEnd) {                          /* second part :2079-2090 */
                        edres_t b = need_edlib(&W, 1, readAlnStart, (uint32_t)readAlnLen, readAlnEnd - re_new, 1, refAlnStart, (uint32_t)refAlnLen, refAlnEnd - te_new, 0);
                        ab_front_ops(&ab, &b, refAlnStart, (uint32_t)refAlnLen);
                        editScore -= b.ed;
                    }
                    ab_front_run(&ab, 'I', '-', re_new);
                    tmp.flag = isRev ? 16 : 0; tmp.pos = te_new; tmp.qStart = re_new;
                    numAnchorsSoFar = 0;
                }
            }
            if (!handled) { editScore -= r.ed; ab_back_ops(&ab, &r, 0, refAlnStart, (uint32_t)refAlnLen); }
        } else if (readAlnLen > 0) {
            ab_back_run(&ab, 'I', '-', (size_t)readAlnLen);
            editScore -= readAlnLen;
        } else {
            if (refAlnLen > 0) ab_back_del(&ab, refAlnStart, (uint32_t)refAlnLen);
            editScore -= refAlnLen;
        }
        numAnchorsSoFar++;
    }

    /* ---- last anchor and the tail (:2149-2230) ---- */
    ab_back_run(&ab, 'M', '=', s[i].len);
    tmp.posEnd = s[i].tPos + s[i].len - 1; tmp.qEnd = s[i].qPos + s[i].len - 1;
    readAlnStart = s[i].qPos + s[i].len;
    readAlnLen = readLen - (int32_t)readAlnStart; refAlnLen = readAlnLen + 20;
    if (readAlnLen > 0) {
        if (s[i].tPos + s[i].len + (uint32_t)refAlnLen - 1 <= chrEnd) {
            refAlnStart = s[i].tPos + s[i].len;
            edres_t r = need_edlib(&W, 0, readAlnStart, (uint32_t)readAlnLen, (uint32_t)readAlnLen, 0, refAlnStart, (uint32_t)refAlnLen, (uint32_t)refAlnLen, 1);
            int realigned = 0;
            if (r.have && readAlnLen > CLIP_LEN && (1 - ((float)r.ed / readAlnLen)) < CLIP_SIM) {
                if (!need_ksw(&W, 0, 0, readAlnStart, (uint32_t)readAlnLen, 0, refAlnStart, (uint32_t)refAlnLen, &qle, &tle)) goto bail;
                if (qle > 0 && qle < readAlnLen) {
                    edres_t r2 = need_edlib(&W, 0, readAlnStart, (uint32_t)readAlnLen, (uint32_t)qle, 0, refAlnStart, (uint32_t)refAlnLen, (uint32_t)tle, 0);
                    ab_back_ops(&ab, &r2, 0, refAlnStart, (uint32_t)refAlnLen);
                    editScore -= r2.ed;
                    tmp.posEnd = refAlnStart + (uint32_t)r2.end;
                    tmp.qEnd = readAlnStart + (uint32_t)qle;
                    ab_back_run(&ab, 'I', '-', (size_t)(readAlnLen - qle));
                    realigned = 1;
                }
            }
            if (!realigned) {
                editScore -= r.ed;
                ab_back_ops(&ab, &r, 0, refAlnStart, (uint32_t)refAlnLen);
                tmp.posEnd = refAlnStart + (uint32_t)r.end;
                tmp.qEnd = (uint32_t)readLen;
            }
        } else { ab_back_run(&ab, 'I', '-', (size_t)readAlnLen); }
    }
    tmp.nmCount = editScore;
    emit_sam(&W, map, &tmp, &ab);
bail:
    *need_track = ab.need_track;
    ab_free(&ab);
    job->complete = (W.missing == 0 && !W.bail);
    if (!job->complete || ab.need_track) { samlist_clear(map); rs->rin = rin0; rs->rrn = rrn0; }
    else if (track_mode == 2 && active) rs->rin = ab.item_mark;          /* pieces after the last record are dropped */
    return job->complete;
}

/* per-base fallback only: a lazy path (op 0 on every diagonal move) copied back from HBM gets its mismatches here,
 * by the comparison the edlib kernels make: raw bytes of the request's query and target strings */
static void resolve_lazy_ops(ctx_t *cx, int tid, job_t *job, memo_t *m, uint32_t nops)
{
    const rkey_t *k = &m->key;
    rd_t *rd = &cx->reads[job->read];
    walk_t W; memset(&W, 0, sizeof W);
    rd_host_bases(cx, rd, &cx->arena[tid]);
    W.cx = cx; W.tid = tid; W.job = job; W.query = job->isRev ? rd->seq_rev : rd->seq; W.readLen = rd->len;
    char *q = (char *)ar_alloc(&cx->arena[tid], (size_t)k->qn + k->tn + 2), *t = q + k->qn + 1;
    put_query(&W, k, q); put_target(&W, k, t);
    uint8_t *ops = m->hops + ((size_t)k->qn + k->tn - nops);
    uint32_t qi = 0, ti = 0;
    for (uint32_t i = 0; i < nops; i++) {
        const uint8_t op = ops[i];
        if (op == 1) { qi++; continue; }
        if (op == 2) { ti++; continue; }
        if (op == 0 && q[qi] != t[ti]) ops[i] = 3;
        qi++; ti++;
    }
}

static int walk_chain(ctx_t *cx, int tid, job_t *job, samlist_t *map)
{
    int need_track = 0;
    const int active = job->nmemo > 0;        /* a first walk has no results yet: it only registers requests */
    const int mode = cx->host_cigar ? 0 : 2;
    int done = walk_chain_mode(cx, tid, job, map, mode, active, &need_track);
    if (done && !active) done = walk_chain_mode(cx, tid, job, map, mode, 1, &need_track);   /* chain without any alignment */
    if (done && need_track) {                                                               /* rare: per-base tracks on the host */
        for (int k = 0; k < job->nmemo; k++) {          /* bring this job's edit paths back from HBM */
            memo_t *m = &job->memo[k];
            if (m->key.type != 0 || m->round < 0 || m->hops) continue;
            const ed_round_t *R = &cx->ed_rounds[m->round];
            if (!R->lazy && (R->ops || !R->d_ops)) continue;
            const size_t region = (size_t)m->key.qn + m->key.tn;
            m->hops = (uint8_t *)ar_alloc(&cx->arena[tid], region + 1);
            if (R->ops) memcpy(m->hops, R->ops + R->ops_off[m->slot], region);
            else if (lfg_fetch(cx->ix->device, m->hops, R->d_ops + R->ops_off[m->slot], region) != LF_OK) { m->hops = NULL; return 0; }
            if (R->lazy) resolve_lazy_ops(cx, tid, job, m, R->ops_len[m->slot]);
        }
        done = walk_chain_mode(cx, tid, job, map, 1, 1, &need_track);
    }
    return done;
}


/* alignWin's scoring tail (src/LordFAST.cpp:1063-1090,1148-1175) */
static void score_mapping(const lf_params_t *p, samlist_t *map, int isReverse, uint32_t rLen, uint32_t chainLen)
{
    if (chainLen > 1) {
        map->totalScore = 0;
        for (int i = 0; i < map->n; i++) {
            map->v[i].alnScore = (int32_t)((uint32_t)map->v[i].nmCount + (map->v[i].qEnd - map->v[i].qStart));
            map->totalScore += map->v[i].nmCount;
        }
        const double gp = isReverse ? p->gap_penalty : 0.15;                                  /* :1077 vs :1162 */
        for (int i = 0; i + 1 < map->n; i++) {
            int64_t a = (int64_t)map->v[i + 1].pos - (int64_t)map->v[i].posEnd, b = (int64_t)map->v[i + 1].qStart - (int64_t)map->v[i].qEnd;
            uint32_t diff = (uint32_t)((a < 0 ? -a : a) + (b < 0 ? -b : b));
            map->totalScore = (int32_t)((double)map->totalScore - gp * (double)diff);
        }
        map->totalScore = (int32_t)((uint32_t)map->totalScore - map->v[0].qStart);
        map->totalScore = (int32_t)((uint32_t)map->totalScore - (rLen - map->v[map->n - 1].qEnd));
    } else map->totalScore = (int32_t)((uint32_t)-2 * rLen);
}

/* ================================================================ E: printSamEntry (src/LordFAST.cpp:318-459) */
static void intv_info(const struct lf_index *ix, uint32_t pos, uint32_t posEnd, const char **name, uint32_t *cbeg)
{
    int rid = pos2rid(ix, (int64_t)(((uint64_t)pos + (uint64_t)posEnd) >> 1));
    *cbeg = (uint32_t)((uint64_t)pos - (uint64_t)ix->contigs[rid].offset);
    *name = ix->contigs[rid].name;
}

static void sam_line(str_t *o, const ctx_t *cx, const rd_t *r, const sam_t *s, int flag, const char *rname, uint32_t rstart, int mapq)
{
    str_puts(o, r->name); str_putc(o, '\t'); str_puti(o, flag); str_putc(o, '\t'); str_puts(o, rname); str_putc(o, '\t');
    str_putu(o, rstart + 1); str_putc(o, '\t'); str_puti(o, mapq >= 0 ? mapq : 0); str_putc(o, '\t');
    str_puts(o, s->cigar); str_puts(o, "\t*\t0\t0\t");
    if (s->flag & 16) { str_put_rc(o, r->seq, r->len); str_putc(o, '\t'); str_put_rev(o, r->qual, r->isFq ? r->len : 1); }
    else { str_putn(o, r->seq, r->len); str_putc(o, '\t'); str_puts(o, r->qual); }
    str_puts(o, "\tAS:i:"); str_puti(o, s->alnScore); str_puts(o, "\tXS:i:0\tNM:i:"); str_puti(o, abs(s->nmCount));
    str_puts(o, "\tMD:Z:"); str_puts(o, s->md);
    if (cx->p->read_group_id[0]) { str_puts(o, "\tRG:Z:"); str_puts(o, cx->p->read_group_id); }
}

static void print_sam_entry(ctx_t *cx, rd_t *r, int num)
{
    str_t *o = &r->out;
    rd_host_bases(cx, r, &cx->arena[0]);      /* called from the lane driver's serial loop (or, host SAM path, never in device-input mode) */
    const samlist_t *mp = r->maps;
    const int readLen = (int)r->len, maxWin = cx->p->max_map;
    const double bestEdit = (num > 0 ? (double)(-1 * mp[0].totalScore) / readLen : 1);
    const double mapqPortion = 50.0 / (maxWin - 1);
    int x1 = 0, x2 = 0;
    for (int i = 0; i < num; i++) if (mp[i].n > 0) { x1++; if ((double)(-1 * mp[i].totalScore) / readLen * 0.95 < bestEdit) x2++; }
    const double mapq = (x2 > 1 ? 2.1 : (maxWin - x1) * mapqPortion);
    int32_t mapq_int;
    for (int i = 0; i < num; i++) {
        if (i == 0) {
            if (mp[0].n > 0) {
                const double e0 = (double)(-1 * mp[0].totalScore) / readLen;
                if (num == 1 || (num > 1 && e0 < 0.15 && e0 < 0.95 * (double)(-1 * mp[1].totalScore) / readLen)) mapq_int = 60;
                else mapq_int = (int32_t)(mapq + 5 * (0.2 - e0) / 0.2);
                const int ns = mp[0].n;
                str_t *sa = (str_t *)calloc((size_t)ns, sizeof(str_t));
                const char **rn = (const char **)calloc((size_t)ns, sizeof(char *));
                uint32_t *rs = (uint32_t *)calloc((size_t)ns, sizeof(uint32_t));
                for (int j = 0; j < ns; j++) {
                    const sam_t *s = &mp[0].v[j];
                    intv_info(cx->ix, s->pos, s->posEnd, &rn[j], &rs[j]);
                    if (ns > 1) {
                        str_init(&sa[j]);
                        str_puts(&sa[j], rn[j]); str_putc(&sa[j], ','); str_putu(&sa[j], rs[j] + 1); str_putc(&sa[j], ',');
                        str_puts(&sa[j], (s->flag & 16) ? "-," : "+,"); str_puts(&sa[j], s->cigar); str_putc(&sa[j], ',');
                        str_puti(&sa[j], mapq_int); str_putc(&sa[j], ','); str_puti(&sa[j], abs(s->nmCount)); str_putc(&sa[j], ';');
                    }
                }
                for (int j = 0; j < ns; j++) {
                    const sam_t *s = &mp[0].v[j];
                    sam_line(o, cx, r, s, j > 0 ? (s->flag | 2048) : s->flag, rn[j], rs[j], mapq_int);
                    if (ns > 1) { str_puts(o, "\tSA:Z:"); for (int z = 0; z < ns; z++) if (z != j) str_putn(o, sa[z].s, sa[z].n); }
                    str_putc(o, '\n');
                }
                if (ns > 1) for (int j = 0; j < ns; j++) free(sa[j].s);
                free(sa); free(rn); free(rs);
            } else {
                str_puts(o, r->name); str_puts(o, "\t4\t*\t0\t0\t*\t*\t0\t0\t"); str_putn(o, r->seq, r->len); str_putc(o, '\t'); str_puts(o, r->qual);
                if (cx->p->read_group_id[0]) { str_puts(o, "\tRG:Z:"); str_puts(o, cx->p->read_group_id); }
                str_putc(o, '\n');
            }
        } else if (mp[i].n > 0) {
            mapq_int = (int32_t)(mapq + 5 * (0.2 - (double)(-1 * mp[i].totalScore) / readLen) / 0.2);
            for (int j = 0; j < mp[i].n; j++) {
                const sam_t *s = &mp[i].v[j];
                const char *rn; uint32_t rs;
                intv_info(cx->ix, s->pos, s->posEnd, &rn, &rs);
                sam_line(o, cx, r, s, s->flag | 256, rn, rs, mapq_int);
                str_putc(o, '\n');
            }
        }
    }
}

/* ---- the same decisions as print_sam_entry, as 48-byte line descriptors for lf_sam.hip (which writes the text) ---- */
typedef struct {
    lf_samline_t *ln; int *rd; int n, cap;  /* rd: the read (index in the chunk) a line belongs to */
    char *blob; uint64_t nb, capb;          /* SA:Z values and literal lines */
    char *names; uint64_t nn, capn;
    int cur_rd;
} linevec_t;
static lf_samline_t *lv_line(linevec_t *v)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8192; v->ln = (lf_samline_t *)realloc(v->ln, (size_t)v->cap * sizeof(lf_samline_t)); v->rd = (int *)realloc(v->rd, (size_t)v->cap * sizeof(int)); }
    v->rd[v->n] = v->cur_rd;
    lf_samline_t *l = &v->ln[v->n++]; memset(l, 0, sizeof *l);
    return l;
}
static uint64_t lv_blob(linevec_t *v, const char *s, size_t n)
{
    if (v->nb + n + 1 > v->capb) { v->capb = (v->nb + n + 1) * 2 + 4096; v->blob = (char *)realloc(v->blob, v->capb); }
    memcpy(v->blob + v->nb, s, n); v->nb += n;
    return v->nb - n;
}
static size_t rec_index(const ctx_t *cx, const sam_t *s) { return s->rtid == -2 ? (size_t)s->rec : (size_t)cx->rrbase[s->rtid] + (size_t)s->rec; }
/* CIGAR of a record as a C string (SA:Z values of split alignments need a few of them on the host) */
static char *rec_cigar_host(ctx_t *cx, const sam_t *s, arena_t *ar)
{
    if (s->cigar) return s->cigar;
    const size_t g = rec_index(cx, s);
    const uint32_t len = cx->rlens[2 * g];                                  /* incl. NUL */
    char *buf = (char *)ar_alloc(ar, (size_t)len + 1);
    if (lfg_fetch(cx->ix->device, buf, (const char *)cx->rtext_dev.d_text + cx->roffs[2 * g], len) != LF_OK) buf[0] = 0;
    buf[len ? len - 1 : 0] = 0;
    return buf;
}
static int rid_of_record(const struct lf_index *ix, const sam_t *s, uint32_t *cbeg)
{
    const int rid = pos2rid(ix, (int64_t)(((uint64_t)s->pos + (uint64_t)s->posEnd) >> 1));     /* bwt_get_intv_info: contig of the midpoint */
    *cbeg = (uint32_t)((uint64_t)s->pos - (uint64_t)ix->contigs[rid].offset);
    return rid;
}
static void line_mapped(ctx_t *cx, linevec_t *v, const rd_t *r, uint32_t name_off, const sam_t *s, int flag, int rid, uint32_t rstart, int mapq, uint32_t sa_off, uint32_t sa_len)
{
    lf_samline_t *l = lv_line(v);
    l->kind = LF_SL_MAPPED; l->name_off = name_off; l->name_len = (uint16_t)strlen(r->name); l->flag = (uint16_t)flag;
    l->read = (uint32_t)r->seed_idx; l->rname = rid; l->pos1 = rstart + 1; l->mapq = mapq >= 0 ? mapq : 0;
    l->as = s->alnScore; l->nm = (uint32_t)abs(s->nmCount); l->rec = (uint32_t)rec_index(cx, s);
    l->sa_off = sa_off; l->sa_len = sa_len; l->is_fq = (uint8_t)r->isFq;
}
static void lines_sam_entry(ctx_t *cx, linevec_t *v, rd_t *r, uint32_t name_off, int num, arena_t *ar)
{   /* src/LordFAST.cpp:318-459; the arithmetic is print_sam_entry's, expression by expression */
    const samlist_t *mp = r->maps;
    const int readLen = (int)r->len, maxWin = cx->p->max_map;
    const double bestEdit = (num > 0 ? (double)(-1 * mp[0].totalScore) / readLen : 1);
    const double mapqPortion = 50.0 / (maxWin - 1);
    int x1 = 0, x2 = 0;
    for (int i = 0; i < num; i++) if (mp[i].n > 0) { x1++; if ((double)(-1 * mp[i].totalScore) / readLen * 0.95 < bestEdit) x2++; }
    const double mapq = (x2 > 1 ? 2.1 : (maxWin - x1) * mapqPortion);
    int32_t mapq_int;
    for (int i = 0; i < num; i++) {
        if (i == 0) {
            if (mp[0].n > 0) {
                const double e0 = (double)(-1 * mp[0].totalScore) / readLen;
                if (num == 1 || (num > 1 && e0 < 0.15 && e0 < 0.95 * (double)(-1 * mp[1].totalScore) / readLen)) mapq_int = 60;
                else mapq_int = (int32_t)(mapq + 5 * (0.2 - e0) / 0.2);
                const int ns = mp[0].n;
                if (ns == 1) {
                    uint32_t rs; const int rid = rid_of_record(cx->ix, &mp[0].v[0], &rs);
                    line_mapped(cx, v, r, name_off, &mp[0].v[0], mp[0].v[0].flag, rid, rs, mapq_int, 0, 0);
                } else {
                    /* split alignment: every record carries the others in SA:Z (rname,pos,strand,CIGAR,mapQ,NM;) (:358-373) */
                    str_t *sa = (str_t *)calloc((size_t)ns, sizeof(str_t));
                    int *rid = (int *)calloc((size_t)ns, sizeof(int)); uint32_t *rs = (uint32_t *)calloc((size_t)ns, sizeof(uint32_t));
                    for (int j = 0; j < ns; j++) {
                        const sam_t *s = &mp[0].v[j];
                        rid[j] = rid_of_record(cx->ix, s, &rs[j]);
                        str_init(&sa[j]);
                        str_puts(&sa[j], cx->ix->contigs[rid[j]].name); str_putc(&sa[j], ','); str_putu(&sa[j], rs[j] + 1); str_putc(&sa[j], ',');
                        str_puts(&sa[j], (s->flag & 16) ? "-," : "+,"); str_puts(&sa[j], rec_cigar_host(cx, s, ar)); str_putc(&sa[j], ',');
                        str_puti(&sa[j], mapq_int); str_putc(&sa[j], ','); str_puti(&sa[j], abs(s->nmCount)); str_putc(&sa[j], ';');
                    }
                    for (int j = 0; j < ns; j++) {
                        const sam_t *s = &mp[0].v[j];
                        uint64_t o0 = v->nb; uint32_t ln = 0;
                        for (int z = 0; z < ns; z++) if (z != j) { const uint64_t o = lv_blob(v, sa[z].s, sa[z].n); if (!ln) o0 = o; ln += (uint32_t)sa[z].n; }
                        line_mapped(cx, v, r, name_off, s, j > 0 ? (s->flag | 2048) : s->flag, rid[j], rs[j], mapq_int, (uint32_t)o0, ln);
                    }
                    for (int j = 0; j < ns; j++) free(sa[j].s);
                    free(sa); free(rid); free(rs);
                }
            } else {
                lf_samline_t *l = lv_line(v);
                l->kind = LF_SL_UNMAPPED; l->name_off = name_off; l->name_len = (uint16_t)strlen(r->name); l->read = (uint32_t)r->seed_idx; l->is_fq = (uint8_t)r->isFq;
            }
        } else if (mp[i].n > 0) {
            mapq_int = (int32_t)(mapq + 5 * (0.2 - (double)(-1 * mp[i].totalScore) / readLen) / 0.2);
            for (int j = 0; j < mp[i].n; j++) {
                const sam_t *s = &mp[i].v[j];
                uint32_t rs; const int rid = rid_of_record(cx->ix, s, &rs);
                line_mapped(cx, v, r, name_off, s, s->flag | 256, rid, rs, mapq_int, 0, 0);
            }
        }
    }
}

/* ================================================================ phases driven by lf_map_batch */
static void phase_prepare(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    memset(&r->out, 0, sizeof r->out);
    if ((int)r->len < cx->p->min_read_len) return;
    r->seq_rev = NULL; r->qual_rev = NULL;     /* built on demand: walk_query (rare byte-string requests); SAM prints them in place */
}

static void phase_concat(ctx_t *cx, int tid, int k)
{
    (void)tid;
    const rd_t *r = &cx->reads[cx->seed_map[k]];
    memcpy(cx->cat + cx->cat_off[k], r->seq, r->len);
}

static void phase_make_jobs(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (r->mode < 2) return;
    r->jobs = (job_t *)ar_zalloc(&cx->arena[tid], ((size_t)r->nWins + 1) * sizeof(job_t));
    r->maps = (samlist_t *)ar_zalloc(&cx->arena[tid], ((size_t)cx->p->max_map + 1) * sizeof(samlist_t));
    /* clasp, fine mode: chain_seeds_clasp called for a window WITHOUT seeds (all of them in the neighbouring contig) sets the
     * score to -1 and leaves the previous call's chain in place (src/Chain.cpp:68,92); alignWin (src/LordFAST.cpp:1028-1046)
     * then extends that stale chain -- the read's previous window, or the last candidate calcChainScore looked at -- on the
     * window's own strand, and its record is printed once more as a secondary.  Restated: such a window takes the request of
     * the most recent chain call of this read that had seeds (none before it in the read: thread history in the reference,
     * empty here; positions above 2 * 10^9, where the reference also adds its shift to the stale chain once more: not restated). */
    int last_req = -1;
    if (cx->p->chain_alg == 1 && r->mode == 3 && !cx->host_vote)
        for (int c = 0; c < r->ncand; c++) if (cx->chain_len[r->cands[c].req] > 0) last_req = r->cands[c].req;
    for (int w = 0; w < r->nWins; w++) {
        job_t *j = &r->jobs[w];
        int rq = r->wins[w].req;
        if (cx->p->chain_alg == 1 && r->mode == 3 && !cx->host_vote) { if (cx->chain_len[rq] == 0 && last_req >= 0) rq = last_req; else if (cx->chain_len[rq] > 0) last_req = rq; }
        j->req = rq;
        j->read = ri; j->widx = w; j->isRev = r->wins[w].isReverse;
        j->chainLen = cx->chain_len[rq];
        j->chain = (Seed_t *)ar_alloc(&cx->arena[tid], ((size_t)j->chainLen + 1) * sizeof(Seed_t));
        if (!cx->host_vote) memcpy(j->chain, cx->vc.chain_seeds + cx->vc.chain_off[rq], (size_t)j->chainLen * sizeof(Seed_t));
        else {
            const creq_t *cq = &cx->creq[rq];
            for (uint32_t k = 0; k < j->chainLen; k++) j->chain[k] = cx->cseeds[cq->off + cx->chain_idx[cq->off + k]];
        }
        j->complete = (j->chainLen <= 1);                   /* nothing to extend: totalScore = -2L (:1089) */
    }
}

static void phase_walk(ctx_t *cx, int tid, int ri)
{
    rd_t *r = &cx->reads[ri];
    if (r->mode < 2) return;
    for (int w = 0; w < r->nWins; w++) {
        job_t *j = &r->jobs[w];
        if (j->complete) continue;
        walk_chain(cx, tid, j, &r->maps[w]);
    }
}

/* the records' strings are in the rendered text now */
static void phase_bind_text(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (r->mode < 2) return;
    for (int w = 0; w < r->nWins; w++)
        for (int j = 0; j < r->maps[w].n; j++) {
            sam_t *s = &r->maps[w].v[j];
            if (s->rec < 0) continue;
            const size_t g = s->rtid == -2 ? (size_t)s->rec : (size_t)cx->rrbase[s->rtid] + (size_t)s->rec;
            s->cigar = cx->rtext + cx->roffs[2 * g]; s->md = cx->rtext + cx->roffs[2 * g + 1];
        }
}

static void phase_strlen(ctx_t *cx, int tid, int i) { (void)tid; cx->len_out[i] = (uint32_t)strlen(cx->len_seqs[i]); }
static void phase_checklen(ctx_t *cx, int tid, int i)
{
    (void)tid;
    const char *s = cx->len_seqs[i]; const uint32_t l = cx->len_out[i];
    if (s[l] != 0 || (l > 0 && s[l - 1] == 0)) __sync_lock_test_and_set(&cx->len_bad, i);
}

static void phase_merge_desc(ctx_t *cx, int tid, int t)
{
    (void)tid;
    stage_t *s = &cx->stages[t];
    int g = cx->mg_gbase[t];
    const uint64_t ob = cx->mg_qbase[t];
    for (int k = 0; k < s->dn; k++, g++) {
        cx->mg_desc[g] = s->dd[k];
        cx->mg_R->ops_off[g] = ob + s->dops[k];
        memo_t *m = &cx->edd_jobs[t].job[k]->memo[s->downer[k]];
        m->round = cx->mg_round; m->slot = g;
    }
    s->dn = 0; s->dops_total = 0; cx->edd_jobs[t].n = 0;
}


/* E is three passes so that the SAM text is formatted straight into its final place:
 * score + order the mappings; print in COUNT mode (exact record sizes); print again into the final buffer */
static void phase_sam_score(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (r->mode < 2) {
        r->maps = (samlist_t *)ar_zalloc(&cx->arena[tid], 2 * sizeof(samlist_t));
        if (cx->stage_sink) { int32_t *o = (int32_t *)malloc(8); o[0] = r->mode; o[1] = 0; cx->stage_sink[cx->stage_i0 + ri] = o; }
        return;
    }
    for (int w = 0; w < r->nWins; w++) score_mapping(cx->p, &r->maps[w], r->wins[w].isReverse, r->len, r->jobs[w].chainLen);
    if (cx->stage_sink) {
        /* the stage view (lf_map_stages_batch): what mapSeq holds between alignWin and the sort / print (src/LordFAST.cpp:547, :561):
         * { mode, n_wins, then per window tStart, tEnd, isReverse, score bits, totalScore, n_records, 7 ints per record } */
        size_t words = 2;
        for (int w = 0; w < r->nWins; w++) words += 6 + 7 * (size_t)r->maps[w].n;
        int32_t *o = (int32_t *)malloc(words * 4); size_t k = 0;
        o[k++] = r->mode; o[k++] = r->nWins;
        for (int w = 0; w < r->nWins; w++) {
            const win_t *wn = &r->wins[w]; const samlist_t *m = &r->maps[w];
            o[k++] = (int32_t)wn->tStart; o[k++] = (int32_t)wn->tEnd; o[k++] = wn->isReverse; memcpy(&o[k++], &wn->score, 4); o[k++] = m->totalScore; o[k++] = m->n;
            for (int j = 0; j < m->n; j++) { const sam_t *x = &m->v[j]; o[k++] = (int32_t)x->pos; o[k++] = (int32_t)x->posEnd; o[k++] = (int32_t)x->qStart; o[k++] = (int32_t)x->qEnd; o[k++] = x->flag; o[k++] = x->alnScore; o[k++] = x->nmCount; }
        }
        cx->stage_sink[cx->stage_i0 + ri] = o;
    }
    if (r->mode == 3) samsort_sort(r->maps, r->nWins);                     /* std::sort(compareSam) :565 */
}
static void phase_sam_print(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (cx->out_base) { r->out.s = cx->out_base + cx->out_off[ri]; r->out.cap = r->out.n; r->out.n = 0; r->out.mode = 2; }
    else { r->out.s = NULL; r->out.cap = 0; r->out.n = 0; r->out.mode = 1; }
    print_sam_entry(cx, r, r->mode < 2 ? 1 : (r->mode == 2 ? 1 : r->nWins));
}

static const char *volatile g_lane_mark[LF_MAX_LANES];        /* LF_WATCHDOG: the last stage mark of every lane */
static inline void tmark(ctx_t *cx, const char *label)
{
    if (cx->lane >= 0 && cx->lane < LF_MAX_LANES) g_lane_mark[cx->lane] = label;
    if (cx->timing && cx->n_marks < 96) { cx->marks[cx->n_marks].label = label; cx->marks[cx->n_marks].t = now_ms(); cx->n_marks++; }
}
static void tmark_dump(ctx_t *cx, double t_begin)
{
    if (!cx->timing) return;
    char line[4096]; int o = 0; double prev = t_begin;
    for (int k = 0; k < cx->n_marks && o < 3900; k++) { o += snprintf(line + o, sizeof line - (size_t)o, " %s %.1f", cx->marks[k].label, cx->marks[k].t - prev); prev = cx->marks[k].t; }
    fprintf(stderr, "[lf] lane %d timeline (ms per step, start t=%.1f):%s\n", cx->lane, t_begin, line);
}

/* ---------------------------------------------------------------- one chunk of reads through all stages */
static int map_chunk(ctx_t *cx)
{
    const int n = cx->n_reads, nt = cx->n_threads;
    lf_stats_t *st = cx->st;
    int rc = LF_OK;
    double t0 = now_ms(), t1;
    const int timing = getenv("LF_TIMING") != NULL, timing0 = timing;
    cx->timing = timing; cx->n_marks = 0;
    const double t_begin = t0;

    parallel_for(cx, n, phase_prepare);
    tmark(cx, "prepare");
    if (getenv("LF_TIMING")) fprintf(stderr, "[lf] prepare %.1f ms\n", now_ms() - t0);

    /* ---- A: seeds ---- */
    lfg_hits_t hits; memset(&hits, 0, sizeof hits);
    {
        int *map = (int *)malloc((size_t)n * sizeof(int)); int m = 0;
        uint64_t bases = 0;
        for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) { cx->reads[i].seed_idx = m; map[m++] = i; bases += cx->reads[i].len; }
        if (m) {
            char *cat = cx->d_seqs ? NULL : (char *)lfg_pin_slot(LF_PS_READS, bases + 64);
            uint64_t *off = (uint64_t *)lfg_pin_slot(LF_PS_READOFF, ((size_t)m + 1) * 8 * (cx->d_seqs ? 2 : 1));
            if ((!cat && !cx->d_seqs) || !off) { free(map); return LF_ERR_NOMEM; }
            uint64_t o = 0;
            for (int k = 0; k < m; k++) { off[k] = o; o += cx->reads[map[k]].len; }
            off[m] = o;
            cx->seed_map = map; cx->cat = cat; cx->cat_off = off;
            double tc0 = now_ms();
            if (cx->d_seqs) {      /* bases already in HBM: a device gather replaces the host concatenation + H2D copy */
                uint64_t *so = off + m + 1;
                for (int k = 0; k < m; k++) so[k] = cx->reads[map[k]].src_off;
                rc = lfg_seed_src(cx->ix, cx->p, m, NULL, cx->d_seqs, so, off, cx->host_vote, &hits);
            } else {
            parallel_for(cx, m, phase_concat);
            tmark(cx, "concat");
            if (getenv("LF_TIMING")) fprintf(stderr, "[lf] concat %.1f ms\n", now_ms() - tc0);
            tc0 = now_ms();
            rc = lfg_seed(cx->ix, cx->p, m, cat, off, cx->host_vote, &hits);
            }
            tmark(cx, "SEED");
            if (getenv("LF_TIMING")) fprintf(stderr, "[lf] lfg_seed %.1f ms (search %.1f locate %.1f), %llu hits\n", now_ms() - tc0, hits.ms_search, hits.ms_locate, (unsigned long long)hits.n_hits);
            if (rc != LF_OK) { free(map); return rc; }
            /* the vote stage sorts 2 keys per hit with 31-bit indices: a chunk above the limit is cut in two by the
             * caller and mapped again (the reference has no such limit; it must stay an implementation detail) */
            if (hits.n_hits >= cx->max_chunk_hits && n > 1) { free(map); cx->seed_map = NULL; return LF_RC_SPLIT; }
            st->n_seeds += hits.n_hits; st->n_cache += hits.counters[0]; st->n_occblk += hits.counters[1]; st->n_sa += hits.counters[2]; st->n_readbytes += hits.counters[3];
            st->ms_k_search += hits.ms_search; st->ms_k_accept += hits.ms_accept; st->ms_k_locate += hits.ms_locate;
            st->search_launches++; st->locate_launches++;
        }
        free(map); cx->seed_map = NULL;
        cx->hits = &hits;
    }
    double tstage[8] = { 0 };
    t1 = now_ms(); st->ms_seed += t1 - t0; tstage[0] = t1 - t0; t0 = t1;

    const int host_vote = cx->host_vote;
    if (!host_vote) {
        /* ---- B + C on the device: votes -> candidate windows -> sorted requests -> chains; only chains come back ---- */
        uint32_t max_len = 0;
        for (int i = 0; i < n; i++) if (cx->reads[i].len > max_len) max_len = cx->reads[i].len;
        int m = 0;
        for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) m++;
        rc = lfg_vote_chain(cx->ix, cx->p, m, hits.n_hits, max_len, &cx->vc);
        if (timing0) fprintf(stderr, "[lf] lfg_vote_chain %.1f ms (vote %.1f chain %.1f), %d requests, %llu request seeds, %llu chain seeds\n", now_ms() - t0,
                             cx->vc.ms_vote, cx->vc.ms_chain, cx->vc.n_req, (unsigned long long)cx->vc.n_req_seeds, (unsigned long long)cx->vc.n_chain_seeds);
        tmark(cx, "VOTECHAIN");
        if (rc != LF_OK) return rc;
        cx->n_creq = cx->vc.n_req; cx->chain_len = cx->vc.chain_len; cx->chain_score = cx->vc.chain_score;
        st->ms_k_vote += cx->vc.ms_vote; st->ms_k_chain += cx->vc.ms_chain; st->n_chain_problems += (uint64_t)cx->vc.n_req;
        st->n_req_seeds += cx->vc.n_req_seeds; st->n_tie_requests += cx->vc.n_tie_req;
        parallel_for(cx, n, phase_select);
        t1 = now_ms(); st->ms_vote += t1 - t0; tstage[1] = t1 - t0; t0 = t1;
        parallel_for(cx, n, phase_fine_select);
        parallel_for(cx, n, phase_make_jobs);
        tmark(cx, "select+jobs");
        t1 = now_ms(); st->ms_chain += t1 - t0; tstage[2] = t1 - t0; t0 = t1;
        goto extend;
    }
    /* ---- B: vote + chain requests ---- */
    cx->cstage = (cstage_t *)calloc((size_t)nt, sizeof(cstage_t));
    parallel_for(cx, n, phase_vote);
    {   /* merge the per-worker chain requests; rebase request ids */
        int total = 0; uint64_t seeds = 0;
        int *base = (int *)malloc((size_t)nt * sizeof(int)); uint64_t *sbase = (uint64_t *)malloc((size_t)nt * 8);
        for (int t = 0; t < nt; t++) { base[t] = total; sbase[t] = seeds; total += cx->cstage[t].n; seeds += cx->cstage[t].ns; }
        cx->n_creq = total; cx->n_cseeds = seeds;
        cx->creq = (creq_t *)malloc(((size_t)total + 1) * sizeof(creq_t));
        cx->cseeds = (Seed_t *)malloc((seeds + 1) * sizeof(Seed_t));
        for (int t = 0; t < nt; t++) {
            memcpy(cx->cseeds + sbase[t], cx->cstage[t].s, cx->cstage[t].ns * sizeof(Seed_t));
            for (int k = 0; k < cx->cstage[t].n; k++) { creq_t q = cx->cstage[t].v[k]; q.off += sbase[t]; cx->creq[base[t] + k] = q; }
        }
        /* request ids handed out during the vote were worker-local: rebase them */
        for (int i = 0; i < n; i++) {
            rd_t *r = &cx->reads[i];
            if (r->mode == 2) r->wins[0].req += base[r->vote_tid];
            else if (r->mode == 3) for (int c = 0; c < r->ncand; c++) r->cands[c].req += base[r->vote_tid];
        }
        free(base); free(sbase);
        for (int t = 0; t < nt; t++) { free(cx->cstage[t].v); free(cx->cstage[t].s); free(cx->cstage[t].wbuf); free(cx->cstage[t].wbuf2); }
        free(cx->cstage); cx->cstage = NULL;
    }
    t1 = now_ms(); st->ms_vote += t1 - t0; tstage[1] = t1 - t0; t0 = t1;

    /* ---- C: chains ---- */
    {
        uint64_t *off = (uint64_t *)malloc(((size_t)cx->n_creq + 1) * 8);
        for (int g = 0; g < cx->n_creq; g++) off[g] = cx->creq[g].off;
        off[cx->n_creq] = cx->n_cseeds;
        cx->chain_idx = (uint32_t *)malloc((cx->n_cseeds + 1) * 4);
        cx->chain_len = (uint32_t *)calloc((size_t)cx->n_creq + 1, 4);
        cx->chain_score = (float *)calloc((size_t)cx->n_creq + 1, 4);
        float ms = 0;
        rc = lfg_chain_n2(cx->ix->device, cx->p, cx->n_creq, cx->cseeds, off, cx->chain_idx, cx->chain_len, cx->chain_score, &ms);
        free(off);
        if (rc != LF_OK) return rc;
        st->ms_k_chain += ms; st->n_chain_problems += (uint64_t)cx->n_creq;
    }
    parallel_for(cx, n, phase_fine_select);
    parallel_for(cx, n, phase_make_jobs);
    t1 = now_ms(); st->ms_chain += t1 - t0; tstage[2] = t1 - t0; t0 = t1;

extend:
    /* ---- D: extension rounds ---- */
    cx->stages = (stage_t *)calloc((size_t)nt, sizeof(stage_t));
    cx->ed_jobs = (jobvec_t *)calloc((size_t)nt, sizeof(jobvec_t));
    cx->ksw_jobs = (jobvec_t *)calloc((size_t)nt, sizeof(jobvec_t));
    cx->edd_jobs = (jobvec_t *)calloc((size_t)nt, sizeof(jobvec_t));
    /* ---- D0: the common path of alignChain_edlib on the device (lf_walk.hip): pieces -> descriptors -> alignments ->
     * clip / split triggers -> record fields + CIGAR recipe, all in HBM.  Chains that leave the common path (and everything
     * when a host cross-check mode is on) are replayed by the host walk below, which then finds them incomplete. ---- */
    cx->n_dev_recs = 0; cx->n_dev_items = 0; cx->d_dev_recs = NULL; cx->d_dev_items = NULL;
    if (!host_vote && !cx->host_cigar && !(g_crosscheck & LF_XC_HOST_WALK)) {
        int nj = 0;
        for (int i = 0; i < n; i++) { const rd_t *r = &cx->reads[i]; if (r->mode >= 2) for (int w = 0; w < r->nWins; w++) nj += r->jobs[w].chainLen > 1; }
        if (nj > 0) {
            lf_wjob_t *wj = (lf_wjob_t *)lfg_pin_slot(LF_PS_WALK0 + 2, (size_t)nj * sizeof(lf_wjob_t));
            job_t **owner = (job_t **)malloc((size_t)nj * sizeof(job_t *));
            if (!wj || !owner) { free(owner); return LF_ERR_NOMEM; }
            int k = 0;
            for (int i = 0; i < n; i++) {
                rd_t *r = &cx->reads[i];
                if (r->mode < 2) continue;
                for (int w = 0; w < r->nWins; w++) {
                    job_t *j = &r->jobs[w];
                    if (j->chainLen <= 1) continue;
                    wj[k].req = (uint32_t)j->req; wj[k].read = (uint32_t)r->seed_idx; wj[k].chain_len = j->chainLen; wj[k].is_rev = (uint8_t)j->isRev;
                    wj[k].pad[0] = wj[k].pad[1] = wj[k].pad[2] = 0;
                    owner[k++] = j;
                }
            }
            lfg_walk_t W;
            rc = lfg_walk_plan(cx->ix, nj, wj, &cx->vc, cx->lazy, &W);
            tmark(cx, "WALKPLAN");
            if (rc != LF_OK) { free(owner); return rc; }
            void *dops = NULL, *d_ed = NULL, *d_end = NULL, *d_len = NULL; float ms = 0;
            if (W.n_desc) rc = lfg_edlib_desc_dev(cx->ix, (int)W.n_desc, W.d_desc, W.d_opsoff, W.ops_total, &W.hc, LF_DS_RND0 + 0, &dops, &d_ed, &d_end, &d_len, &ms);
            tmark(cx, "EDLIB0");
            if (rc != LF_OK) { free(owner); return rc; }
            lf_wrec_t *wrec = NULL;
            rc = lfg_walk_emit(cx->ix, &cx->vc, cx->lazy, &W, d_ed, d_end, d_len, &wrec);
            tmark(cx, "WALKEMIT");
            if (rc != LF_OK) { free(owner); return rc; }
            {   /* extension round 0 lives on the device only */
                ed_round_t R; memset(&R, 0, sizeof R);
                R.n = (int)W.n_desc; R.pinned = 1; R.ops_bytes = W.ops_total; R.d_ops = (uint8_t *)dops; R.d_desc = W.d_desc; R.lazy = cx->lazy;
                cx->ed_rounds = (ed_round_t *)realloc(cx->ed_rounds, ((size_t)cx->n_ed_rounds + 1) * sizeof(ed_round_t));
                cx->ed_rounds[cx->n_ed_rounds++] = R;
            }
            st->ms_k_edlib += ms; st->n_edlib_problems += W.n_desc; st->edlib_launches += 1; st->ops_bytes += W.ops_total;
            if (W.n_desc) { float bd[4]; lfg_edlib_breakdown(bd); st->ms_k_rsweep += bd[0]; st->ms_k_tb += bd[1]; st->ms_k_hirsch += bd[2]; st->ms_k_bin += bd[3]; }
            st->ext_bytes += W.ext_bytes; st->dp_block_steps += W.block_steps;
            int n_rare = 0;
            for (k = 0; k < nj; k++) {
                job_t *j = owner[k];
                if (wrec[k].rare) { n_rare++; continue; }
                rd_t *r = &cx->reads[j->read];
                samlist_t *map = &r->maps[j->widx];
                sam_t tmp; memset(&tmp, 0, sizeof tmp);
                tmp.flag = j->isRev ? 16 : 0; tmp.pos = wrec[k].pos; tmp.posEnd = wrec[k].posEnd; tmp.qStart = wrec[k].qStart; tmp.qEnd = wrec[k].qEnd; tmp.nmCount = wrec[k].nm;
                samlist_clear(map);
                samlist_push(map, &tmp, NULL, NULL, &cx->arena[0]);
                map->v[0].rec = k; map->v[0].rtid = -2;                 /* record k of the device-planned recipe */
                j->complete = 1;
            }
            if (timing) fprintf(stderr, "[lf] device walk: %d jobs, %llu pieces aligned, %d jobs (%.1f %%) left to the host replay\n", nj, (unsigned long long)W.n_desc, n_rare, 100.0 * n_rare / nj);
            cx->n_dev_recs = nj; cx->n_dev_items = W.n_items; cx->d_dev_recs = W.d_recs; cx->d_dev_items = W.d_items;
            free(owner);
        }
    }
    for (int round = 0; round < 64; round++) {
        double tw0 = now_ms();
        parallel_for(cx, n, phase_walk);
        tmark(cx, "walk");
        if (timing) fprintf(stderr, "[lf] round %d walk %.1f ms\n", round, now_ms() - tw0);
        int nk = 0, nd = 0;
        for (int t = 0; t < nt; t++) { nk += cx->stages[t].kn; nd += cx->stages[t].dn; }
        if (nk == 0 && nd == 0) break;
        if (nd) {
            /* descriptor requests: nothing but 32-byte descriptors goes to the GPU */
            uint64_t *obase = (uint64_t *)malloc((size_t)nt * 8); int *gbase = (int *)malloc((size_t)nt * sizeof(int));
            uint64_t ops_total = 0;
            { int g0 = 0; for (int t = 0; t < nt; t++) { obase[t] = ops_total; gbase[t] = g0; ops_total += cx->stages[t].dops_total; g0 += cx->stages[t].dn; st->ext_bytes += cx->stages[t].ext_bytes; cx->stages[t].ext_bytes = 0; st->dp_block_steps += cx->stages[t].blk_steps; cx->stages[t].blk_steps = 0; } }
            const int ridx = cx->n_ed_rounds;
            const int pin = ridx < 16;
            ed_round_t R; memset(&R, 0, sizeof R);
            R.n = nd; R.pinned = pin; R.ops_bytes = ops_total;
            /* the edit paths stay in HBM (slot of this round) unless the host builds the strings itself */
            const int host_ops = cx->host_cigar || ridx >= LF_MAX_ED_ROUNDS;
            if (pin) {
                R.ed = (int32_t *)lfg_pin_slot(LF_PS_ROUND0 + 4 * ridx, (size_t)nd * 4); R.end = (int32_t *)lfg_pin_slot(LF_PS_ROUND0 + 4 * ridx + 1, (size_t)nd * 4);
                R.ops_len = (uint32_t *)lfg_pin_slot(LF_PS_ROUND0 + 4 * ridx + 2, (size_t)nd * 4);
                if (host_ops) R.ops = (uint8_t *)lfg_pin_slot(LF_PS_ROUND0 + 4 * ridx + 3, ops_total + 1);
            } else {
                R.ed = (int32_t *)malloc((size_t)nd * 4); R.end = (int32_t *)malloc((size_t)nd * 4);
                R.ops_len = (uint32_t *)malloc((size_t)nd * 4); R.ops = (uint8_t *)malloc(ops_total + 1);
            }
            R.ops_off = (uint64_t *)malloc((size_t)nd * 8);
            lf_aln_desc_t *desc = (lf_aln_desc_t *)lfg_pin_slot(LF_PS_ALN_PROB, (size_t)nd * sizeof(lf_aln_desc_t));
            if (!desc || !R.ed || !R.end || !R.ops_len || (host_ops && !R.ops)) return LF_ERR_NOMEM;
            cx->mg_desc = desc; cx->mg_R = &R; cx->mg_qbase = obase; cx->mg_gbase = gbase; cx->mg_round = ridx;
            double tm0 = now_ms();
            parallel_for(cx, nt, phase_merge_desc);
            tmark(cx, "merge");
            free(obase); free(gbase);
            float ms = 0;
            void *dops = NULL, *ddesc = NULL;
            rc = lfg_edlib_desc(cx->ix, nd, desc, R.ops_off, ops_total, R.ed, R.end, R.ops, R.ops_len,
                                LF_DS_RND0 + 2 * (ridx < LF_MAX_ED_ROUNDS ? ridx : 0), &dops, &ddesc, &ms);
            if (!host_ops) { R.d_ops = (uint8_t *)dops; R.d_desc = ddesc; }
            R.lazy = cx->lazy;
            tmark(cx, "EDLIB");
            if (timing) fprintf(stderr, "[lf] round %d: %d descriptor problems, merge+solve %.1f ms (kernels %.1f ms), ops %.1f MB\n", round, nd, now_ms() - tm0, ms, ops_total / 1e6);
            cx->ed_rounds = (ed_round_t *)realloc(cx->ed_rounds, ((size_t)cx->n_ed_rounds + 1) * sizeof(ed_round_t));
            cx->ed_rounds[cx->n_ed_rounds++] = R;
            if (rc != LF_OK) return rc;
            st->ms_k_edlib += ms; st->n_edlib_problems += (uint64_t)nd; st->edlib_launches += 1; st->ops_bytes += ops_total;
            { float bd[4]; lfg_edlib_breakdown(bd); st->ms_k_rsweep += bd[0]; st->ms_k_tb += bd[1]; st->ms_k_hirsch += bd[2]; st->ms_k_bin += bd[3]; }
        }
        if (nk) {
            uint64_t qn = 0, tn = 0;
            for (int t = 0; t < nt; t++) { qn += cx->stages[t].kqn; tn += cx->stages[t].ktn; }
            uint8_t *qb = (uint8_t *)malloc(qn + 1), *tb = (uint8_t *)malloc(tn + 1);
            uint64_t *qoff = (uint64_t *)malloc(((size_t)nk + 1) * 8), *toff = (uint64_t *)malloc(((size_t)nk + 1) * 8);
            int32_t *prm = (int32_t *)malloc((size_t)nk * 7 * 4);
            ksw_round_t R; memset(&R, 0, sizeof R);
            R.n = nk; R.score = (int32_t *)malloc((size_t)nk * 4); R.qle = (int32_t *)malloc((size_t)nk * 4); R.tle = (int32_t *)malloc((size_t)nk * 4);
            const int ridx = cx->n_ksw_rounds;
            int g = 0; uint64_t qo = 0, to = 0;
            for (int t = 0; t < nt; t++) {
                stage_t *s = &cx->stages[t];
                memcpy(qb + qo, s->kq, s->kqn); memcpy(tb + to, s->kt, s->ktn);
                for (int k = 0; k < s->kn; k++, g++) {
                    qoff[g] = qo + s->kqoff[k]; toff[g] = to + s->ktoff[k];
                    memcpy(prm + 7 * g, s->kprm + 7 * k, 28);
                    memo_t *m = &cx->ksw_jobs[t].job[k]->memo[(uintptr_t)s->kowner[k]];
                    m->round = ridx; m->slot = g;
                }
                qo += s->kqn; to += s->ktn;
                s->kn = 0; s->kqn = 0; s->ktn = 0; cx->ksw_jobs[t].n = 0;
            }
            qoff[nk] = qo; toff[nk] = to;
            float ms = 0;
            rc = lfg_ksw(cx->ix->device, nk, qb, qoff, tb, toff, prm, R.score, R.qle, R.tle, &ms);
            tmark(cx, "KSW");
            free(qb); free(tb); free(qoff); free(toff); free(prm);
            cx->ksw_rounds = (ksw_round_t *)realloc(cx->ksw_rounds, ((size_t)cx->n_ksw_rounds + 1) * sizeof(ksw_round_t));
            cx->ksw_rounds[cx->n_ksw_rounds++] = R;
            if (rc != LF_OK) return rc;
            st->ms_k_ksw += ms; st->n_ksw_problems += (uint64_t)nk; st->ksw_bytes += qn + tn + 12ull * (uint64_t)nk;
        }
    }
    t1 = now_ms(); st->ms_extend += t1 - t0; tstage[3] = t1 - t0; t0 = t1;

    /* ---- D': CIGAR / MD text of every record, rendered on the GPU from the paths in HBM ---- */
    if (!cx->host_cigar) {
        uint64_t n_items = 0; int n_recs = 0;
        cx->rrbase = (int *)malloc((size_t)nt * sizeof(int));
        uint64_t *ibase = (uint64_t *)malloc((size_t)nt * 8);
        for (int t = 0; t < nt; t++) { cx->rrbase[t] = cx->n_dev_recs + n_recs; ibase[t] = cx->n_dev_items + n_items; n_recs += cx->stages[t].rrn; n_items += cx->stages[t].rin; }
        if (n_recs + cx->n_dev_recs) {
            if (n_items + cx->n_dev_items >= 0xffffffffull) { lf_set_error("lf_map_batch: too many CIGAR pieces in one chunk"); free(ibase); return LF_ERR_ARG; }
            lf_ritem_t *items = (lf_ritem_t *)lfg_pin_slot(LF_PS_RENDER0 + 3, (n_items + 1) * sizeof(lf_ritem_t));
            lf_rrecord_t *recs = (lf_rrecord_t *)lfg_pin_slot(LF_PS_RENDER0 + 4, ((size_t)n_recs + 1) * sizeof(lf_rrecord_t));
            if (!items || !recs) { free(ibase); return LF_ERR_NOMEM; }
            for (int t = 0; t < nt; t++) {
                const stage_t *s = &cx->stages[t];
                if (s->rin) memcpy(items + (ibase[t] - cx->n_dev_items), s->ri, s->rin * sizeof(lf_ritem_t));
                for (int k = 0; k < s->rrn; k++) { lf_rrecord_t q = s->rr[k]; q.item0 += (uint32_t)ibase[t]; recs[cx->rrbase[t] - cx->n_dev_recs + k] = q; }
            }
            /* rounds whose paths were computed through the host (Hirschberg-size problems): put them into HBM too */
            const void *round_ops[LF_MAX_ED_ROUNDS], *round_desc[LF_MAX_ED_ROUNDS]; memset(round_ops, 0, sizeof round_ops); memset(round_desc, 0, sizeof round_desc);
            for (int k = 0; k < cx->n_ed_rounds && k < LF_MAX_ED_ROUNDS; k++) {
                ed_round_t *Rk = &cx->ed_rounds[k];
                if (!Rk->d_ops && Rk->ops && Rk->ops_bytes) {
                    void *dp = lfg_dev_slot(cx->ix->device, LF_DS_RND0 + 2 * k, Rk->ops_bytes + 64);
                    if (!dp) { free(ibase); return LF_ERR_NOMEM; }
                    rc = lfg_upload(cx->ix->device, dp, Rk->ops, Rk->ops_bytes);
                    if (rc != LF_OK) { free(ibase); return rc; }
                    Rk->d_ops = (uint8_t *)dp;
                }
                round_ops[k] = Rk->d_ops; round_desc[k] = Rk->d_desc;
            }
            float ms = 0; uint64_t tbytes = 0;
            tmark(cx, "recipe");
            rc = lfg_render(cx->ix, cx->n_dev_recs, cx->d_dev_recs, cx->n_dev_items, cx->d_dev_items, n_recs, recs, n_items, items, round_ops, round_desc, &cx->rtext, &cx->roffs, &tbytes, &ms,
                            cx->dev_sam ? &cx->rtext_dev : NULL, &cx->rlens);
            tmark(cx, "RENDER");
            if (timing) fprintf(stderr, "[lf] render: %d records, %llu pieces, %.1f MB text, kernels %.1f ms, total %.1f ms\n", n_recs, (unsigned long long)n_items, tbytes / 1e6, ms, now_ms() - t0);
            if (rc != LF_OK) { free(ibase); return rc; }
            st->ms_k_render += ms; st->render_bytes += tbytes; st->render_launches += 1;
            if (!cx->dev_sam) parallel_for(cx, n, phase_bind_text);
        }
        free(ibase);
    }
    t1 = now_ms(); st->ms_render += t1 - t0; tstage[4] = t1 - t0; t0 = t1;

    /* ---- E: SAM (score + count here; the text is written by lf_map_batch straight into the output buffer) ---- */
    parallel_for(cx, n, phase_sam_score);
    cx->out_base = NULL;
    if (cx->dev_sam) {
        /* SAM lines on the device: the host only says what is printed (48 bytes per line) */
        linevec_t V; memset(&V, 0, sizeof V);
        arena_t *ar = &cx->arena[0];
        int any_fq = 0;
        for (int i = 0; i < n; i++) {
            rd_t *r = &cx->reads[i];
            V.cur_rd = i;
            const size_t nl = strlen(r->name);
            if (V.nn + nl + 1 > V.capn) { V.capn = (V.nn + nl + 1) * 2 + 65536; V.names = (char *)realloc(V.names, V.capn); }
            memcpy(V.names + V.nn, r->name, nl);
            const uint32_t name_off = (uint32_t)V.nn; V.nn += nl;
            if (r->mode == 0) {                 /* shorter than -l: not in the resident batch; the whole line is literal text */
                str_init(&r->out); print_sam_entry(cx, r, 1);
                lf_samline_t *l = lv_line(&V); l->kind = LF_SL_LITERAL; l->sa_off = (uint32_t)lv_blob(&V, r->out.s, r->out.n); l->sa_len = (uint32_t)r->out.n;
                free(r->out.s); memset(&r->out, 0, sizeof r->out);
                continue;
            }
            any_fq |= r->isFq;
            const int num = r->mode == 3 ? r->nWins : 1;
            int host_strings = nl > 65535;       /* a name longer than the line descriptor's 16-bit length, or a record whose CIGAR / MD were built per base on the host (the reference's misaligned-MD branch) */
            if (r->mode >= 2) for (int w = 0; w < num; w++) for (int j = 0; j < r->maps[w].n; j++) host_strings |= r->maps[w].v[j].rec < 0;
            if (host_strings) {
                /* rare: print the whole entry on the host (its other records' text is fetched from the device) */
                for (int w = 0; w < num; w++) for (int j = 0; j < r->maps[w].n; j++) {
                    sam_t *sr = &r->maps[w].v[j];
                    if (sr->rec < 0) continue;
                    const size_t g = rec_index(cx, sr);
                    sr->cigar = rec_cigar_host(cx, sr, ar);
                    const uint32_t ml = cx->rlens[2 * g + 1];
                    sr->md = (char *)ar_alloc(ar, (size_t)ml + 1);
                    if (lfg_fetch(cx->ix->device, sr->md, (const char *)cx->rtext_dev.d_text + cx->roffs[2 * g + 1], ml) != LF_OK) sr->md[0] = 0;
                    sr->md[ml ? ml - 1 : 0] = 0;
                }
                str_init(&r->out); print_sam_entry(cx, r, num);
                lf_samline_t *l = lv_line(&V); l->kind = LF_SL_LITERAL; l->sa_off = (uint32_t)lv_blob(&V, r->out.s, r->out.n); l->sa_len = (uint32_t)r->out.n;
                free(r->out.s); memset(&r->out, 0, sizeof r->out);
                continue;
            }
            lines_sam_entry(cx, &V, r, name_off, num, ar);
        }
        if (V.nb >= 0xffffffffull || V.nn >= 0xffffffffull) { free(V.ln); free(V.rd); free(V.blob); free(V.names); lf_set_error("lf_map_batch: chunk too large for the SAM writer"); return LF_ERR_ARG; }
        char *qcat = NULL; uint64_t qbytes = 0; int n_batch = 0;
        for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) n_batch++;
        if (any_fq && cx->d_quals) { for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) qbytes += cx->reads[i].len; }
        else if (any_fq && cx->holes) qcat = (char *)"";      /* the host prints the qualities itself: the device only has to know that there are some */
        else if (any_fq) {                            /* FASTQ: the qualities in the layout of the resident read batch */
            for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) qbytes += cx->reads[i].len;
            qcat = (char *)malloc(qbytes + 1);
            uint64_t o = 0;
            for (int i = 0; i < n; i++) { const rd_t *r = &cx->reads[i]; if ((int)r->len < cx->p->min_read_len) continue; if (r->isFq) memcpy(qcat + o, r->qual, r->len); else memset(qcat + o, '*', r->len); o += r->len; }
        }
        const uint64_t *h_offs = NULL; const uint32_t *h_hole = NULL;
        rc = lfg_sam_build(cx->ix, cx->p, V.n, V.ln, V.names, V.nn, V.blob, V.nb, qcat, qbytes, (any_fq && cx->d_quals) ? cx->d_quals : NULL, n_batch,
                           &cx->rtext_dev, cx->sam_parity, cx->holes, &cx->sam_total, &h_offs, &h_hole);
        cx->fill = NULL; cx->n_fill = 0;
        if (rc == LF_OK && cx->holes && V.n > 0) {
            /* where the host puts SEQ (/ QUAL): one entry per line that has a hole */
            cx->fill = (fill_t *)malloc(((size_t)V.n + 1) * sizeof(fill_t));
            for (int k = 0; k < V.n; k++) {
                if (!h_hole[2 * (size_t)k + 1]) continue;
                const rd_t *r = &cx->reads[V.rd[k]];
                fill_t *f = &cx->fill[cx->n_fill++];
                f->pos = h_offs[k] + h_hole[2 * (size_t)k]; f->seq = r->seq; f->qual = r->qual; f->len = r->len;
                f->rev = (uint8_t)(V.ln[k].kind == LF_SL_MAPPED && (V.ln[k].flag & 16)); f->fq = (uint8_t)(h_hole[2 * (size_t)k + 1] > r->len);
            }
        }
        free(V.ln); free(V.rd); free(V.blob); free(V.names); if (!(any_fq && cx->holes)) free(qcat);
        if (rc != LF_OK) return rc;
    } else parallel_for(cx, n, phase_sam_print);
    t1 = now_ms(); st->ms_sam += t1 - t0; tstage[5] = t1 - t0;
    tmark(cx, "samcount");
    tmark_dump(cx, t_begin);
    if (timing) fprintf(stderr, "[lf] lane %d chunk of %d reads: seed %.1f vote %.1f chain %.1f extend %.1f render %.1f sam-count %.1f ms (t=%.1f)\n",
                        cx->lane, n, tstage[0], tstage[1], tstage[2], tstage[3], tstage[4], tstage[5], now_ms());
    return LF_OK;
}

static void chunk_free(ctx_t *cx)
{
    for (int t = 0; t < cx->n_threads; t++) ar_reset(&cx->arena[t]);      /* every per-read object at once */
    free(cx->creq); free(cx->cseeds); free(cx->chain_idx);
    if (cx->host_vote) { free(cx->chain_len); free(cx->chain_score); }          /* device path: pinned slots of lfg_vote_chain */
    for (int k = 0; k < cx->n_ed_rounds; k++) { ed_round_t *R = &cx->ed_rounds[k]; if (!R->pinned) { free(R->ed); free(R->end); free(R->ops_len); free(R->ops); } free(R->ops_off); }
    for (int k = 0; k < cx->n_ksw_rounds; k++) { ksw_round_t *R = &cx->ksw_rounds[k]; free(R->score); free(R->qle); free(R->tle); }
    free(cx->ed_rounds); free(cx->ksw_rounds);
    if (cx->stages) {
        for (int t = 0; t < cx->n_threads; t++) {
            stage_t *s = &cx->stages[t];
            free(s->qb); free(s->tb); free(s->qoff); free(s->toff); free(s->mode); free(s->owner);
            free(s->kq); free(s->kt); free(s->kqoff); free(s->ktoff); free(s->kprm); free(s->kowner);
            free(s->dd); free(s->dops); free(s->downer); free(s->ri); free(s->rr);
            free(cx->ed_jobs[t].job); free(cx->ksw_jobs[t].job); free(cx->edd_jobs[t].job);
        }
        free(cx->stages); free(cx->ed_jobs); free(cx->ksw_jobs); free(cx->edd_jobs); cx->ed_jobs = cx->ksw_jobs = cx->edd_jobs = NULL;
    }
    cx->creq = NULL; cx->cseeds = NULL; cx->chain_idx = NULL; cx->chain_len = NULL; cx->chain_score = NULL;
    cx->ed_rounds = NULL; cx->ksw_rounds = NULL; cx->n_ed_rounds = cx->n_ksw_rounds = 0; cx->stages = NULL;
    free(cx->rrbase); cx->rrbase = NULL; cx->rtext = NULL; cx->roffs = NULL;
}

/* Several batches may be mapped at once (calls from different threads): a call's lane drivers take LANE IDS -- the key of
 * the per-lane device slots, streams, arenas and pool job slot -- from one process-wide allocator, lowest free id first, and
 * give them back when they run out of chunks.  `cap` bounds the ids in use at a time (8 per device: every id owns a few GB of
 * grow-only working memory in HBM), so a second large batch waits for lanes of the first instead of doubling the working
 * set, while small batches (a rank's 12 k-read shards under strong scaling) overlap: the launch / sync chain of one hides
 * behind the kernels of the others. */
static pthread_mutex_t g_lanes_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_lanes_cv = PTHREAD_COND_INITIALIZER;
static unsigned g_lanes_used;                    /* bit l: lane id l is taken */
static int g_active_calls;                       /* batches inside map_batch_core (the pool is only resized when there is none) */
/* next / n0: the batch's chunk cursor -- a driver that would only find its batch's chunks all taken gives up (-1) instead of
 * waiting for a lane another batch holds */
static int lane_acquire(int cap, volatile int *next, int n0)
{
    pthread_mutex_lock(&g_lanes_mu);
    for (;;) {
        if (next && *next >= n0) { pthread_mutex_unlock(&g_lanes_mu); return -1; }
        if (__builtin_popcount(g_lanes_used) < cap) for (int l = 0; l < LF_MAX_LANES; l++) if (!(g_lanes_used & (1u << l))) { g_lanes_used |= 1u << l; pthread_mutex_unlock(&g_lanes_mu); return l; }
        pthread_cond_wait(&g_lanes_cv, &g_lanes_mu);
    }
}
static void lane_release(int lane)
{
    pthread_mutex_lock(&g_lanes_mu);
    g_lanes_used &= ~(1u << lane);
    pthread_cond_broadcast(&g_lanes_cv);
    pthread_mutex_unlock(&g_lanes_mu);
}

/* ---- a batch is cut into chunks; two lane threads pull chunks and run them through map_chunk.  While one lane waits
 * for the GPU the other lane's host phases keep the cores busy.  SAM text is written in chunk order. ---- */
typedef struct { int i0, i1; uint64_t size; int sized; } chunk_t;
typedef struct {
    const lf_index_t *const *ixs; int n_ix;       /* one replica of the index per device; lane l works on device l % n_ix */
    const lf_params_t *p;
    const char *const *names, *const *seqs, *const *quals; const uint32_t *lens;
    const unsigned char *d_seqs, *d_quals; const uint64_t *src_off; int dev_out;     /* lf_map_batch_dev: bases / qualities / SAM text in HBM */
    int32_t **stage_sink;                       /* lf_map_stages_batch */
    int slots;                                  /* per-worker scratch slots = pool workers + lane ids */
    int lane_cap;                               /* lane ids this process may have in use while this batch takes one (lane_acquire) */
    chunk_t *chunks; int n_chunks, n_chunks0; volatile int next_chunk;   /* n_chunks0: entries cut up front; n_chunks grows when a lane cuts a chunk */
    pthread_mutex_t mu; pthread_cond_t cv;      /* chunk sizes become known in any order */
    pthread_rwlock_t grow;                      /* writers of SAM text hold it shared; growing the buffer exclusive */
    int host_cigar, host_vote; str_t all; int fixed_out;                   /* fixed_out: caller-provided buffer, never reallocated */
    int holes;                                  /* the output buffer is pinned host memory and the reads are host strings: SEQ-less egress (lf_sam.hip) */
    volatile int rc; char err[1024];
    lf_stats_t st[LF_MAX_LANES];
} batch_t;

static void merge_stats(lf_stats_t *d, const lf_stats_t *a)
{
    d->ms_seed += a->ms_seed; d->ms_vote += a->ms_vote; d->ms_chain += a->ms_chain; d->ms_extend += a->ms_extend; d->ms_sam += a->ms_sam;
    d->ms_k_search += a->ms_k_search; d->ms_k_accept += a->ms_k_accept; d->ms_k_locate += a->ms_k_locate; d->ms_k_chain += a->ms_k_chain;
    d->ms_k_edlib += a->ms_k_edlib; d->ms_k_ksw += a->ms_k_ksw;
    d->n_reads += a->n_reads; d->n_bases += a->n_bases; d->n_seeds += a->n_seeds; d->n_chain_problems += a->n_chain_problems;
    d->n_edlib_problems += a->n_edlib_problems; d->n_ksw_problems += a->n_ksw_problems; d->n_cache += a->n_cache; d->n_occblk += a->n_occblk;
    d->n_sa += a->n_sa; d->n_readbytes += a->n_readbytes; d->ext_bytes += a->ext_bytes; d->edlib_launches += a->edlib_launches;
    d->search_launches += a->search_launches; d->locate_launches += a->locate_launches;
    d->dp_block_steps += a->dp_block_steps; d->ms_render += a->ms_render; d->ms_k_render += a->ms_k_render; d->ms_k_vote += a->ms_k_vote; d->ops_bytes += a->ops_bytes; d->n_req_seeds += a->n_req_seeds; d->n_tie_requests += a->n_tie_requests; d->render_bytes += a->render_bytes; d->render_launches += a->render_launches;
    d->ksw_bytes += a->ksw_bytes;
    d->ms_k_rsweep += a->ms_k_rsweep; d->ms_k_tb += a->ms_k_tb; d->ms_k_hirsch += a->ms_k_hirsch; d->ms_k_bin += a->ms_k_bin;
}

/* base offset of a chunk's text = the sizes of all chunks of earlier reads.  block == 0: returns 0 when one of them has not
 * published its size yet (entries can be added while we wait: rescan after every wake-up) */
static int chunk_base(batch_t *B, const chunk_t *C, int block, uint64_t *base_out)
{
    pthread_mutex_lock(&B->mu);
    uint64_t base;
    for (;;) {
        int waiting = 0; base = 0;
        for (int j = 0; j < B->n_chunks; j++) {
            if (B->chunks[j].i1 > C->i0) continue;
            if (!B->chunks[j].sized) { waiting = 1; break; }
            base += B->chunks[j].size;
        }
        if (!waiting) break;
        if (!block) { pthread_mutex_unlock(&B->mu); return 0; }
        pthread_cond_wait(&B->cv, &B->mu);
    }
    pthread_mutex_unlock(&B->mu);
    *base_out = base;
    return 1;
}
/* makes room for [base, base + tot] in the batch's output and holds the read lock on return (rc: B->rc) */
static void out_reserve(batch_t *B, uint64_t base, uint64_t tot)
{
    pthread_rwlock_rdlock(&B->grow);
    if (base + tot + 1 > B->all.cap) {          /* rare: the up-front estimate was too small */
        pthread_rwlock_unlock(&B->grow);
        pthread_rwlock_wrlock(&B->grow);
        if (base + tot + 1 > B->all.cap) {
            if (B->fixed_out) { snprintf(B->err, sizeof B->err, "lf_map_batch_into: output buffer too small (need more than %llu bytes)", (unsigned long long)(base + tot + 1)); B->rc = LF_ERR_NOMEM; }
            else { size_t nc = (size_t)((base + tot + 1) * 1.25) + 4096; B->all.s = (char *)realloc(B->all.s, nc); B->all.cap = nc; }
        }
        pthread_rwlock_unlock(&B->grow);
        pthread_rwlock_rdlock(&B->grow);
    }
}
/* a chunk whose SAM text is complete in one of the lane's two device buffers but whose place in the output is not known
 * yet (an earlier chunk is still being mapped by another lane): the lane maps its next chunk first */
typedef struct { const chunk_t *C; const lf_index_t *ix; uint64_t tot; int parity, active; fill_t *fill; int n_fill; } pending_t;
/* HOLES mode: SEQ (/ QUAL) of line k goes to out_base + fill[k].pos -- the strings printSamEntry prints (src/LordFAST.cpp:377-402):
 * the read as given, or its reverse complement / reversed qualities for a record on the reverse strand (:501-502) */
static void phase_fill(ctx_t *cx, int tid, int k)
{
    (void)tid;
    const fill_t *f = &cx->fill[k];
    char *d = cx->out_base + f->pos;
    if (!f->rev) memcpy(d, f->seq, f->len); else rc_copy(d, f->seq, f->len);
    if (f->fq) {
        d[f->len] = '\t';
        char *q = d + f->len + 1;
        if (!f->rev) memcpy(q, f->qual, f->len); else for (uint32_t i = 0; i < f->len; i++) q[i] = f->qual[f->len - 1 - i];
    }
}
static void fetch_dev_sam(batch_t *B, const lf_index_t *ix, uint64_t base, uint64_t tot, int parity, lf_stats_t *st, fill_t *fill, int n_fill, int lane)
{
    if (B->rc != LF_OK) { free(fill); return; }
    const double t0 = now_ms();
    out_reserve(B, base, tot);
    if (B->rc == LF_OK) {               /* one D2H copy of the chunk's text straight into its place */
        /* a caller-provided buffer never moves: the copy runs behind the lane's back (lfg_sam_fetch_wait at the lane's end);
         * a growable one may be reallocated by another lane, so the copy completes under the read lock */
        const int frc = B->fixed_out ? lfg_sam_fetch_async(ix, B->all.s + base, tot, parity) : lfg_sam_fetch(ix, B->all.s + base, tot, parity);
        if (frc != LF_OK) { snprintf(B->err, sizeof B->err, "%s", lf_last_error()); B->rc = frc; }
        else if (n_fill > 0) {          /* the holes, while the scatter kernel moves the rest over the link */
            ctx_t fx; memset(&fx, 0, sizeof fx);
            fx.lane = lane; fx.n_threads = B->slots; fx.fill = fill; fx.n_fill = n_fill; fx.out_base = B->all.s + base;
            parallel_for(&fx, n_fill, phase_fill);
        }
    }
    pthread_rwlock_unlock(&B->grow);
    free(fill);
    st->ms_sam += now_ms() - t0;
}

static void *lane_main(void *arg_)
{
    batch_t *B = (batch_t *)((void **)arg_)[0];
    const int held = (int)(intptr_t)((void **)arg_)[1];
    const int lane = held ? held - 1 : lane_acquire(B->lane_cap, &B->next_chunk, B->n_chunks0);
    if (lane < 0) return NULL;
    const int timing = getenv("LF_TIMING") != NULL;
    lfg_set_lane(lane);
    const long long lane_c0 = g_phase_on ? thread_cpu_ns() : 0;
    lf_stats_t *st = &B->st[lane];
    uint64_t max_hits = 1ull << 30;
    if (getenv("LF_MAX_CHUNK_HITS")) { max_hits = strtoull(getenv("LF_MAX_CHUNK_HITS"), NULL, 10); if (max_hits < 1) max_hits = 1; }   /* test hook */
    int todo[66], n_todo = 0;                       /* second halves of chunks this lane had to cut */
    pending_t pend; memset(&pend, 0, sizeof pend);
    int parity = 0;
    for (;;) {
        int k;
        if (n_todo > 0) k = todo[--n_todo];
        else {
            k = __sync_fetch_and_add(&B->next_chunk, 1);
            if (k >= B->n_chunks0) break;
        }
        chunk_t *C = &B->chunks[k];
        if (B->rc != LF_OK) {
            /* another lane failed after this chunk was claimed: publish it as empty, or a lane that claimed a later chunk
             * just before the error would wait for this chunk's size forever */
            pthread_mutex_lock(&B->mu);
            C->size = 0; C->sized = 1;
            pthread_cond_broadcast(&B->cv);
            pthread_mutex_unlock(&B->mu);
            continue;                               /* drain the remaining chunk ids the same way */
        }
        ctx_t cx; memset(&cx, 0, sizeof cx);
        cx.ix = B->ixs[lane % B->n_ix]; cx.p = B->p; cx.n_threads = B->slots; cx.st = st; cx.lane = lane; cx.arena = g_arena[lane]; cx.host_cigar = B->host_cigar; cx.host_vote = B->host_vote; cx.lazy = 0;      /* the traceback kernel classifies diagonal moves itself (from registers); LF_F_LAZYX stays for the kernels' stage users */
        cx.max_chunk_hits = max_hits;
        cx.dev_sam = !B->host_cigar && !B->host_vote && !(g_crosscheck & LF_XC_HOST_SAM);
        cx.holes = B->holes && cx.dev_sam;
        cx.sam_parity = parity;
        cx.d_seqs = B->d_seqs; cx.d_quals = B->d_quals; cx.stage_sink = B->stage_sink; cx.stage_i0 = C->i0;
        cx.n_reads = C->i1 - C->i0;
        cx.reads = (rd_t *)calloc((size_t)cx.n_reads, sizeof(rd_t));
        uint64_t chunk_bases = 0;
        for (int i = C->i0; i < C->i1; i++) {
            rd_t *r = &cx.reads[i - C->i0];
            r->name = B->names[i]; r->len = B->lens[i];
            if (B->d_seqs) { r->seq = NULL; r->src_off = B->src_off[i]; r->isFq = B->d_quals != NULL; r->qual = r->isFq ? NULL : "*"; }
            else {
                r->seq = B->seqs[i];
                r->isFq = (B->quals && B->quals[i] && B->quals[i][0]);
                r->qual = r->isFq ? B->quals[i] : "*";
            }
            chunk_bases += r->len;
        }
        double tch = now_ms();
        int rc = map_chunk(&cx);
        if (rc == LF_RC_SPLIT) {
            /* cut the chunk: this entry keeps the first half, the second half becomes a new entry that this lane maps
             * next.  The new entry is registered before the first half publishes its size, so every chunk behind it
             * sees it when it adds up its base offset. */
            chunk_free(&cx); free(cx.reads);
            if (n_todo >= 63) {     /* every entry still in todo[] (registered second halves) and k itself are then published as empty by the branch above */
                snprintf(B->err, sizeof B->err, "lf_map_batch: a chunk could not be cut below the seed-hit limit"); B->rc = LF_ERR_ARG; todo[n_todo++] = k; continue;
            }
            pthread_mutex_lock(&B->mu);
            const int mid = C->i0 + (C->i1 - C->i0) / 2, nk = B->n_chunks++;
            B->chunks[nk].i0 = mid; B->chunks[nk].i1 = C->i1; B->chunks[nk].size = 0; B->chunks[nk].sized = 0;
            C->i1 = mid;
            pthread_mutex_unlock(&B->mu);
            if (timing) fprintf(stderr, "[lf] lane %d chunk %d: too many seed hits, cut at read %d\n", lane, k, mid);
            todo[n_todo++] = nk; todo[n_todo++] = k;    /* first half first */
            continue;
        }
        st->n_bases += chunk_bases;
        st->n_reads += (uint64_t)cx.n_reads;
        if (timing) fprintf(stderr, "[lf] lane %d chunk %d (%d reads): map_chunk %.1f ms\n", lane, k, cx.n_reads, now_ms() - tch);
        uint64_t tot = 0, *ooff = NULL;
        if (rc == LF_OK) {
            ooff = (uint64_t *)malloc(((size_t)cx.n_reads + 1) * 8);
            if (cx.dev_sam) tot = cx.sam_total;
            else for (int i = 0; i < cx.n_reads; i++) { ooff[i] = tot; tot += cx.reads[i].out.n; }
        } else { snprintf(B->err, sizeof B->err, "%s", lf_last_error()); B->rc = rc; }
        /* publish this chunk's size */
        pthread_mutex_lock(&B->mu);
        C->size = tot; C->sized = 1;
        pthread_cond_broadcast(&B->cv);
        pthread_mutex_unlock(&B->mu);
        if (cx.dev_sam) {
            /* The text sits in device buffer `parity`.  Its place in the output is known once every chunk of earlier reads has
             * published its size; lanes finish out of order, so instead of waiting here the lane keeps ONE chunk pending and
             * maps the next one (into the other buffer).  The older pending chunk must leave its buffer first. */
            uint64_t base;
            if (pend.active) { (void)chunk_base(B, pend.C, 1, &base); fetch_dev_sam(B, pend.ix, base, pend.tot, pend.parity, st, pend.fill, pend.n_fill, lane); pend.active = 0; }
            if (rc == LF_OK) {
                if (chunk_base(B, C, 0, &base)) fetch_dev_sam(B, cx.ix, base, tot, parity, st, cx.fill, cx.n_fill, lane);
                else {
                    pend.C = C; pend.ix = cx.ix; pend.tot = tot; pend.parity = parity; pend.active = 1; pend.fill = cx.fill; pend.n_fill = cx.n_fill;
                    /* the writer kernel still reads this chunk's buffers: the next chunk's first stream waits for it (lfg_sam_build) */
                }
                cx.fill = NULL; cx.n_fill = 0;
                parity ^= 1;
            } else { free(cx.fill); cx.fill = NULL; }
        } else {
            uint64_t base;
            (void)chunk_base(B, C, 1, &base);
            if (rc == LF_OK && B->rc == LF_OK) {
                tch = now_ms();
                out_reserve(B, base, tot);
                if (B->rc == LF_OK) {
                    cx.out_base = B->all.s + base; cx.out_off = ooff;
                    parallel_for(&cx, cx.n_reads, phase_sam_print);
                }
                pthread_rwlock_unlock(&B->grow);
                st->ms_sam += now_ms() - tch;
            }
        }
        free(ooff);
        tch = now_ms();
        chunk_free(&cx);
        free(cx.reads);
        if (timing) fprintf(stderr, "[lf] lane %d chunk %d: chunk_free %.1f ms\n", lane, k, now_ms() - tch);
    }
    if (pend.active) { uint64_t base; (void)chunk_base(B, pend.C, 1, &base); fetch_dev_sam(B, pend.ix, base, pend.tot, pend.parity, st, pend.fill, pend.n_fill, lane); }
    if (B->fixed_out && !B->host_cigar && !B->host_vote) {      /* the asynchronous copies of this lane */
        const double t0 = now_ms();
        const int wrc = lfg_sam_fetch_wait(B->ixs[lane % B->n_ix]);
        if (wrc != LF_OK && B->rc == LF_OK) { snprintf(B->err, sizeof B->err, "%s", lf_last_error()); B->rc = wrc; }
        st->ms_sam += now_ms() - t0;
    }
    if (g_phase_on) phase_account("(lane driver threads, incl. their share of the phases)", (thread_cpu_ns() - lane_c0) / 1e6, 0);
    lane_release(lane);
    return NULL;
}

typedef struct { volatile int stop; int limit_s; } wdog_t;
static void *wdog_main(void *arg)
{
    wdog_t *w = (wdog_t *)arg;
    for (int ms = 0; !w->stop; ms += 50) {
        struct timespec ts = { 0, 50 * 1000000 }; nanosleep(&ts, NULL);
        if (ms >= w->limit_s * 1000) {
            fprintf(stderr, "[lf watchdog] batch still running after %d s\n", w->limit_s);
            for (int l = 0; l < LF_MAX_LANES; l++) if (g_lane_mark[l]) fprintf(stderr, "[lf watchdog] lane %d: last stage mark %s\n", l, g_lane_mark[l]);
            lfg_phase_dump();
            fflush(stderr);
            abort();
        }
    }
    return NULL;
}

typedef struct { const void *d_seqs, *d_quals; const uint64_t *seq_off; int dev_out; int32_t **stage_sink; } devio_t;
static int map_batch_core(const lf_index_t *const *ixs, int n_ix, const lf_params_t *p, int n, const char *const *names,
                          const char *const *seqs, const char *const *quals, const uint32_t *seq_lens, char *ext_buf, size_t ext_cap,
                          char **sam, size_t *sam_len, lf_stats_t *stats, const devio_t *dio)
{
    if (!ixs || n_ix < 1 || n_ix > 16 || !p || n < 0 || (!sam && !ext_buf)) { lf_set_error("lf_map_batch: bad argument"); return LF_ERR_ARG; }
    if (dio && !dio->stage_sink && (n_ix != 1 || !dio->d_seqs || !dio->seq_off || !seq_lens || !ext_buf || g_crosscheck)) {
        lf_set_error("lf_map_batch_dev: needs one index, device bases with offsets and lengths, an output buffer, and no cross-check mode (lf_debug_crosscheck)"); return LF_ERR_ARG;
    }
    for (int d = 0; d < n_ix; d++) {
        if (!ixs[d]) { lf_set_error("lf_map_batch: bad argument"); return LF_ERR_ARG; }
        if (ixs[d]->l_pac != ixs[0]->l_pac || ixs[d]->seq_len != ixs[0]->seq_len || ixs[d]->n_seqs != ixs[0]->n_seqs) { lf_set_error("lf_map_batch_multi: the index replicas differ"); return LF_ERR_ARG; }
    }
    if (p->chain_alg != 0 && p->chain_alg != 1) { lf_set_error("lf_map_batch: chain_alg must be 0 (dp-n2) or 1 (clasp)"); return LF_ERR_ARG; }
    if (p->chain_alg == 1 && (g_crosscheck & LF_XC_HOST_VOTE)) { lf_set_error("lf_map_batch: the host-vote cross-check only knows dp-n2; clasp runs on the device vote path"); return LF_ERR_ARG; }
    if (p->min_anchor_len < 12 || p->min_anchor_len > 20 || p->sampling_count <= 0 || p->max_map < 2 || p->max_ref_hits <= 0 || p->min_read_len < 100) {
        lf_set_error("lf_map_batch: option out of range (k in [12,20], c > 0, n >= 2, m > 0, l >= 100)"); return LF_ERR_ARG;
    }
    lf_stats_t local; memset(&local, 0, sizeof local);
    lf_stats_t *st = stats ? stats : &local;
    memset(st, 0, sizeof *st);
    int nt = p->threads;
    long online = sysconf(_SC_NPROCESSORS_ONLN);
    if (nt <= 0 || nt > online) {                                      /* "all CPUs", src/CommandLineParser.cpp:181-185 ... */
        nt = (int)online;
        /* ... but not more than the cgroup CPU quota grants: oversubscribing a throttled container only adds
         * context switches (cpu.max = "<quota> <period>" on cgroup v2) */
        FILE *fq = fopen("/sys/fs/cgroup/cpu.max", "r");
        if (fq) {
            long long quota = 0, period = 0; char qs[64];
            if (fscanf(fq, "%63s %lld", qs, &period) == 2 && strcmp(qs, "max") != 0 && period > 0) {
                quota = atoll(qs);
                int lim = (int)((quota + period - 1) / period);
                if (lim >= 1 && lim < nt) nt = lim;
            }
            fclose(fq);
        }
    }
    if (nt > 255) nt = 255;
    if (nt < 1) nt = 1;
    const double T0 = now_ms();
    pthread_once(&g_rc_once, rc_tab_init);
    /* chunks in flight: the host phases of one overlap the GPU phases of the others */
    /* drivers sleep while they wait for the GPU (blocking waits), so small thread budgets still get several chunks in flight */
    /* a lane driver spends most of a chunk blocked on the GPU (the chain walk runs on the device now), so the number of
     * chunks in flight is not tied to the thread budget any more: eight from four threads up */
    int n_lanes = nt >= 4 ? 8 : nt;
    if (getenv("LF_LANES")) { n_lanes = atoi(getenv("LF_LANES")); if (n_lanes < 1) n_lanes = 1; if (n_lanes > LF_MAX_LANES) n_lanes = LF_MAX_LANES; }
    if (getenv("LF_ONE_LANE")) n_lanes = 1;
    if (n_ix > 1) {
        /* several devices: LF_LANES / the default is per device (capped by LF_MAX_LANES and the thread budget); every
         * device gets at least one lane.  Lanes pull chunks from one shared counter, so the devices balance themselves. */
        int per = n_lanes; if (per * n_ix > LF_MAX_LANES) per = LF_MAX_LANES / n_ix; if (per < 1) per = 1;
        n_lanes = per * n_ix;
    }
    int nw = nt - n_lanes;                             /* pool workers; the lane drivers work too */
    if (nw < nt / 2) nw = nt / 2;                      /* few threads, many (mostly sleeping) drivers: keep half the budget as workers */
    if (nw > 220) nw = 220;                            /* worker ids: pool threads, then one per lane id (< 260 in all) */
    pthread_mutex_lock(&g_lanes_mu);
    if (g_active_calls == 0 || !g_pool.started) { g_phase_on = getenv("LF_PHASES") != NULL; pool_ensure(nw); }
    else nw = g_pool.nw;                               /* another batch is being mapped: the pool keeps its size */
    g_active_calls++;
    pthread_mutex_unlock(&g_lanes_mu);
    const int lane_cap = n_lanes > 8 ? n_lanes : 8;
    const int lane0 = lane_acquire(lane_cap, NULL, 0); /* this thread's lane id: the set-up passes below, then its chunks */

    batch_t B; memset(&B, 0, sizeof B);
    B.host_cigar = (g_crosscheck & LF_XC_HOST_CIGAR) != 0;
    B.host_vote = (g_crosscheck & LF_XC_HOST_VOTE) != 0;          /* diagnostic cross-check only; the device stage is the product path */
    B.ixs = ixs; B.n_ix = n_ix; B.p = p; B.names = names; B.seqs = seqs; B.quals = quals; B.slots = nw + LF_MAX_LANES; B.rc = LF_OK; B.lane_cap = lane_cap;
    if (dio && dio->stage_sink) B.stage_sink = dio->stage_sink;
    else if (dio) { B.d_seqs = (const unsigned char *)dio->d_seqs; B.d_quals = (const unsigned char *)dio->d_quals; B.src_off = dio->seq_off; B.dev_out = dio->dev_out; }
    pthread_mutex_init(&B.mu, NULL); pthread_cond_init(&B.cv, NULL); pthread_rwlock_init(&B.grow, NULL);
    if (ext_buf) { B.all.s = ext_buf; B.all.cap = ext_cap; B.all.n = 0; B.all.mode = 2; B.fixed_out = 1; }
    else str_init(&B.all);
    /* SEQ-less egress: the caller's reads are host strings (we can print SEQ / QUAL ourselves) and its output buffer is pinned host
     * memory that kernels of every device can store into.  LF_SAM_FULL=1 keeps the whole line on the device (A / B measurements). */
    if (ext_buf && !dio && seqs && !getenv("LF_SAM_FULL")) {
        B.holes = 1;
        for (int d = 0; d < n_ix; d++) if (!lfg_host_mapped(ixs[d]->device, ext_buf, ext_cap)) B.holes = 0;
    }
    uint32_t *lens = (uint32_t *)malloc(((size_t)n + 1) * 4);
    B.lens = lens;
    {   /* read lengths once, in parallel; one allocation for the SAM text (~2 x bases + per-record overhead) */
        ctx_t c0; memset(&c0, 0, sizeof c0);
        c0.n_threads = nw + LF_MAX_LANES; c0.lane = lane0; c0.len_seqs = seqs; c0.len_out = lens;
        if (seq_lens) {                                               /* the caller knows them (Read.length, src/Reads.h): no pass over the bases */
            memcpy(lens, seq_lens, (size_t)n * 4);
            /* a wrong length would make the device read past a string: the terminator of every read is checked (best effort: the
             * check itself trusts lens[i] to stay inside the caller's allocation) */
            int bad = -1;
            if (!dio || dio->stage_sink) { c0.len_bad = -1; parallel_for(&c0, n, phase_checklen); bad = c0.len_bad; }      /* one cold cache line per read: all workers */
            if (bad >= 0) { const int i = bad;
                lf_set_error("lf_map_batch_into_lens: seq_lens[%d] = %u is not the length of seqs[%d]", i, lens[i], i);
                free(lens); pthread_rwlock_destroy(&B.grow); pthread_mutex_destroy(&B.mu); pthread_cond_destroy(&B.cv);
                if (!ext_buf) free(B.all.s);
                lane_release(lane0);
                pthread_mutex_lock(&g_lanes_mu); g_active_calls--; pthread_mutex_unlock(&g_lanes_mu);
                return LF_ERR_ARG;
            }
        }
        else parallel_for(&c0, n, phase_strlen);
        uint64_t est = 4096;
        for (int i = 0; i < n; i++) est += 2 * (uint64_t)lens[i] + 640;
        if (!ext_buf) str_room(&B.all, est + est / 8);
    }
    if (getenv("LF_TIMING")) fprintf(stderr, "[lf] setup (strlen + SAM buffer) %.1f ms, %d lanes, %d pool workers\n", now_ms() - T0, n_lanes, nw);
    /* chunks bound the device + host working set; reads stay in input order */
    uint64_t CHUNK_BASES = 400ull << 20;
    if (getenv("LF_CHUNK_BASES")) { CHUNK_BASES = strtoull(getenv("LF_CHUNK_BASES"), NULL, 10); if (CHUNK_BASES < 1) CHUNK_BASES = 1; }      /* measurement hook: bench.py's exclusive pass maps the whole batch as ONE chunk */
    int CHUNK_READS = 32768;
    if (getenv("LF_CHUNK_READS")) { CHUNK_READS = atoi(getenv("LF_CHUNK_READS")); if (CHUNK_READS < 1) CHUNK_READS = 1; }   /* test hook */
    else if (n_lanes >= 2 && n > 2048) {
        /* Chunks per lane.  Reads already in HBM (lf_map_batch_dev): ONE -- nothing of a chunk waits for a bus, the lanes only
         * overlap each other's host phases, and larger chunks fill the GPU better with fewer launches (100 k reads, 8 lanes, chunks
         * of 3125 / 6250 / 12500 / 16667 / 25000 / 100000 reads: 0.92 / 1.06 / 1.15-1.22 / 1.19 / 1.19 / 1.13 M reads/s).  Host
         * buffers: THREE -- the 1.5 GB of bases going up and the 4 GB of SAM text coming down per 100 k reads overlap the other
         * chunks' kernels better in smaller pieces (same sweep: 768 / 727 / 673-704 / 733 / 705 / 675 k reads/s). */
        /* round 4 (packed k-mer tables, SEQ-less egress: 1.0 GB instead of 2.6 GB of text comes down per 100 k reads): two per lane
         * for host batches whose output buffer is pinned (chunks of 4167 / 6250 / 8334 / 12500 / 25000 reads: 123 / 110 / 113 /
         * 113 / 117 ms per 100 k reads; whole lines: 122 ms at 4167, 136 at 12500).  A chunk's launch / sync chain does not
         * shrink with the chunk, so small batches get FEWER chunks, not smaller ones: at least 6250 reads each (HBM-resident
         * 12.5 k reads as 8 / 4 / 2 / 1 chunks: 17.1 / 17.5 / 15.2 / 15.6 ms; 25 k: 27.5 (8) / 26.5 (4) / 27.6 (2); 50 k: 42.0 (8) / 43.9 (4)). */
        const int dev_in = dio && !dio->stage_sink;
        const int per_lane = dev_in ? 1 : (B.holes ? 2 : 3);
        const int min_chunk = (dev_in || B.holes) ? 6250 : 1024;
        int want = (n + per_lane * n_lanes - 1) / (per_lane * n_lanes); if (want < min_chunk) want = min_chunk;
        if (want < CHUNK_READS) CHUNK_READS = want;
    }
    /* reads x sampling positions is a 31-bit index in the seed stage */
    { const long long cap = (1ll << 30) / (p->sampling_count > 0 ? p->sampling_count : 1); if (cap < CHUNK_READS) CHUNK_READS = cap < 1 ? 1 : (int)cap; }
    B.chunks = (chunk_t *)calloc((size_t)n + 1, sizeof(chunk_t));
    for (int i0 = 0; i0 < n; ) {
        int i1 = i0; uint64_t bases = 0;
        while (i1 < n && i1 - i0 < CHUNK_READS && bases < CHUNK_BASES) { bases += lens[i1]; i1++; }
        B.chunks[B.n_chunks].i0 = i0; B.chunks[B.n_chunks].i1 = i1; B.n_chunks++;
        i0 = i1;
    }
    B.n_chunks0 = B.n_chunks;
    /* LF_WATCHDOG=<seconds>: a batch that takes longer reports where every lane is and aborts (tests set it: a hang
     * becomes a failure with a location) */
    wdog_t wd; memset(&wd, 0, sizeof wd); pthread_t wdt; int have_wd = 0;
    if (getenv("LF_WATCHDOG") && atoi(getenv("LF_WATCHDOG")) > 0) { wd.limit_s = atoi(getenv("LF_WATCHDOG")); have_wd = pthread_create(&wdt, NULL, wdog_main, &wd) == 0; }
    void *la[LF_MAX_LANES][2]; pthread_t lt[LF_MAX_LANES]; int have[LF_MAX_LANES] = { 0 };
    for (int l = 0; l < LF_MAX_LANES; l++) { la[l][0] = &B; la[l][1] = (void *)(intptr_t)(l == 0 ? lane0 + 1 : 0); }      /* [1]: lane id + 1 already held, 0: take one */
    for (int l = 1; l < n_lanes && l < B.n_chunks; l++) have[l] = pthread_create(&lt[l], NULL, lane_main, la[l]) == 0;
    lane_main(la[0]);
    for (int l = 1; l < n_lanes; l++) if (have[l]) pthread_join(lt[l], NULL);
    if (have_wd) { wd.stop = 1; pthread_join(wdt, NULL); }
    lfg_set_lane(0);
    if (g_phase_on) { fprintf(stderr, "[lf] batch of %d reads: %.1f ms wall, %d threads\n", n, now_ms() - T0, nt); phase_report(); }
    pthread_mutex_lock(&g_lanes_mu); g_active_calls--; pthread_mutex_unlock(&g_lanes_mu);

    uint64_t total = 0;
    for (int k = 0; k < B.n_chunks; k++) total += B.chunks[k].size;
    for (int l = 0; l < LF_MAX_LANES; l++) merge_stats(st, &B.st[l]);
    free(lens); free(B.chunks);
    pthread_mutex_destroy(&B.mu); pthread_cond_destroy(&B.cv); pthread_rwlock_destroy(&B.grow);
    st->ms_total = now_ms() - T0;
    if (getenv("LF_TIMING")) fprintf(stderr, "[lf] lf_map_batch total %.1f ms\n", st->ms_total);
    if (B.rc != LF_OK) { lf_set_error("%s", B.err); if (!ext_buf) free(B.all.s); return B.rc; }
    if (!B.dev_out) B.all.s[total] = 0;
    if (sam) *sam = B.all.s;
    if (sam_len) *sam_len = total;
    return LF_OK;
}

int lf_map_batch(const lf_index_t *ix, const lf_params_t *p, int n, const char *const *names,
                 const char *const *seqs, const char *const *quals, char **sam, size_t *sam_len, lf_stats_t *stats)
{
    return map_batch_core(&ix, 1, p, n, names, seqs, quals, NULL, NULL, 0, sam, sam_len, stats, NULL);
}

/* same, into a caller-owned buffer (e.g. pinned and reused across batches: a fresh multi-GB malloc per batch costs
 * page faults on first touch and an munmap on free).  LF_ERR_NOMEM if it is too small; 2.5 x bases + 1 KiB per read
 * is a safe size for error rates up to ~20 %. */
int lf_map_batch_into(const lf_index_t *ix, const lf_params_t *p, int n, const char *const *names,
                      const char *const *seqs, const char *const *quals, char *out, size_t out_cap, size_t *sam_len, lf_stats_t *stats)
{
    if (!out || out_cap < 2) { lf_set_error("lf_map_batch_into: no output buffer"); return LF_ERR_ARG; }
    return map_batch_core(&ix, 1, p, n, names, seqs, quals, NULL, out, out_cap, NULL, sam_len, stats, NULL);
}

/* same with the read lengths supplied (seq_lens[i] == strlen(seqs[i]); the strings stay NUL-terminated): the reference's
 * Read records carry `length` (src/Reads.h), so its callers never measure a read twice either */
int lf_map_batch_into_lens(const lf_index_t *ix, const lf_params_t *p, int n, const char *const *names,
                           const char *const *seqs, const char *const *quals, const uint32_t *seq_lens,
                           char *out, size_t out_cap, size_t *sam_len, lf_stats_t *stats)
{
    if (!out || out_cap < 2) { lf_set_error("lf_map_batch_into_lens: no output buffer"); return LF_ERR_ARG; }
    if (!seq_lens && n > 0) { lf_set_error("lf_map_batch_into_lens: no lengths"); return LF_ERR_ARG; }
    return map_batch_core(&ix, 1, p, n, names, seqs, quals, seq_lens, out, out_cap, NULL, sam_len, stats, NULL);
}

/* Device-resident form: the bases (and qualities) of the batch are already in HBM of idx's device and the SAM text is left
 * there -- nothing of the bulk data crosses PCIe.  What a rank of the N-GPU deployment receives over xGMI is mapped where it
 * landed, and its records leave over xGMI again (lordfast_amd/dist.py).
 *   d_seqs / d_quals: device pointers; read i = d_seqs[seq_off[i] .. seq_off[i] + seq_lens[i]) (anything may sit between two
 *   reads: NULs, names); d_quals NULL = FASTA ("*"), else same layout.  names, seq_off, seq_lens: host arrays.
 *   out: device buffer when out_is_device (else host memory, e.g. pinned), out_cap bytes; no terminating NUL is written to a
 *   device buffer.  A chunk's reads are gathered inside HBM (lf_reads_gather_kernel) instead of concatenated and uploaded. */
int lf_map_batch_dev(const lf_index_t *ix, const lf_params_t *p, int n, const char *const *names, const void *d_seqs,
                     const uint64_t *seq_off, const uint32_t *seq_lens, const void *d_quals, void *out, size_t out_cap,
                     int out_is_device, size_t *sam_len, lf_stats_t *stats)
{
    if (!out || out_cap < 2) { lf_set_error("lf_map_batch_dev: no output buffer"); return LF_ERR_ARG; }
    devio_t dio; dio.d_seqs = d_seqs; dio.d_quals = d_quals; dio.seq_off = seq_off; dio.dev_out = out_is_device != 0; dio.stage_sink = NULL;
    return map_batch_core(&ix, 1, p, n, names, NULL, NULL, seq_lens, (char *)out, out_cap, NULL, sam_len, stats, &dio);
}

/* Stage view of mapSeq for a batch (the reference's findTopWins_coarse / _fine and alignWin are only visible through the SAM
 * records; this entry point shows what lies between them): per read the decision, the windows alignWin is called with --
 * coarse: the best window; fine: the heap array of src/LordFAST.cpp:553-562 in array order -- and per window alignWin's
 * totalScore and records (before the sort and MAPQ of printSamEntry).  Same kernels, same host glue as lf_map_batch. */
int lf_map_stages_batch(const lf_index_t *ix, const lf_params_t *p, int n, const char *const *seqs, lf_stages_t **out)
{
    if (!out || n < 0) { lf_set_error("lf_map_stages_batch: bad argument"); return LF_ERR_ARG; }
    *out = NULL;
    int32_t **sink = (int32_t **)calloc((size_t)n + 1, sizeof(int32_t *));
    const char **names = (const char **)malloc(((size_t)n + 1) * sizeof(char *));
    for (int i = 0; i < n; i++) names[i] = "r";
    devio_t dio; memset(&dio, 0, sizeof dio); dio.stage_sink = sink;
    char *sam = NULL; size_t sl = 0;
    const int rc = map_batch_core(&ix, 1, p, n, names, seqs, NULL, NULL, NULL, 0, &sam, &sl, NULL, &dio);
    free(sam); free(names);
    if (rc != LF_OK) { for (int i = 0; i < n; i++) free(sink[i]); free(sink); return rc; }
    lf_stages_t *S = (lf_stages_t *)calloc(1, sizeof *S);
    S->n_reads = n; S->mode = (uint8_t *)calloc((size_t)n + 1, 1); S->win0 = (uint32_t *)calloc((size_t)n + 2, 4);
    size_t nw = 0, nr = 0;
    for (int i = 0; i < n; i++) if (sink[i]) { const int32_t *o = sink[i]; size_t k = 2; for (int w = 0; w < o[1]; w++) { nr += (size_t)o[k + 5]; k += 6 + 7 * (size_t)o[k + 5]; } nw += (size_t)o[1]; }
    S->wins = (lf_stage_win_t *)calloc(nw + 1, sizeof(lf_stage_win_t)); S->recs = (lf_stage_rec_t *)calloc(nr + 1, sizeof(lf_stage_rec_t));
    nw = 0; nr = 0;
    for (int i = 0; i < n; i++) {
        S->win0[i] = (uint32_t)nw;
        if (!sink[i]) continue;
        const int32_t *o = sink[i]; size_t k = 2;
        S->mode[i] = (uint8_t)o[0];
        for (int w = 0; w < o[1]; w++) {
            lf_stage_win_t *W = &S->wins[nw++];
            W->tStart = (uint32_t)o[k]; W->tEnd = (uint32_t)o[k + 1]; W->isReverse = (uint32_t)o[k + 2]; memcpy(&W->score, &o[k + 3], 4); W->totalScore = o[k + 4]; W->n_records = (uint32_t)o[k + 5]; W->rec0 = (uint32_t)nr;
            k += 6;
            for (uint32_t j = 0; j < W->n_records; j++, k += 7) { lf_stage_rec_t *R = &S->recs[nr++]; R->pos = (uint32_t)o[k]; R->posEnd = (uint32_t)o[k + 1]; R->qStart = (uint32_t)o[k + 2]; R->qEnd = (uint32_t)o[k + 3]; R->flag = o[k + 4]; R->alnScore = o[k + 5]; R->nmCount = o[k + 6]; }
        }
        free(sink[i]);
    }
    S->win0[n] = (uint32_t)nw; S->n_wins = (uint32_t)nw; S->n_recs = (uint32_t)nr;
    free(sink);
    *out = S;
    return LF_OK;
}
void lf_stages_free(lf_stages_t *S) { if (!S) return; free(S->mode); free(S->win0); free(S->wins); free(S->recs); free(S); }

/* one batch over SEVERAL devices of this process: idx[d] is a replica of the same index on its own device
 * (lf_index_load(prefix, device d, ...)).  The batch is cut into chunks that the devices' lanes pull from one counter
 * (the reference's pthread pool pulls reads from a shared cursor the same way, src/LordFAST.cpp:295-303), so the
 * devices balance themselves; SAM records come out in input order -- byte-identical to the one-device output.
 * (Two replicas may share a device -- that is how the one-GPU test box exercises this path.)
 * seq_lens may be NULL; out == NULL: *sam is malloc'd. */
int lf_map_batch_multi(const lf_index_t *const *idx, int n_idx, const lf_params_t *p, int n, const char *const *names,
                       const char *const *seqs, const char *const *quals, const uint32_t *seq_lens,
                       char *out, size_t out_cap, char **sam, size_t *sam_len, lf_stats_t *stats)
{
    if (out && out_cap < 2) { lf_set_error("lf_map_batch_multi: output buffer too small"); return LF_ERR_ARG; }
    return map_batch_core(idx, n_idx, p, n, names, seqs, quals, seq_lens, out, out ? out_cap : 0, out ? NULL : sam, sam_len, stats, NULL);
}

/* printSamHeader (src/BWT.cpp:668-681) */
char *lf_sam_header(const lf_index_t *ix, const lf_params_t *p, const char *cmdline)
{
    str_t sb; str_init(&sb);
    str_puts(&sb, "@HD\tVN:1.5\tSO:unsorted\n");
    for (int i = 0; i < ix->n_seqs; i++) { str_puts(&sb, "@SQ\tSN:"); str_puts(&sb, ix->contigs[i].name); str_puts(&sb, "\tLN:"); str_puti(&sb, ix->contigs[i].len); str_putc(&sb, '\n'); }
    if (p && p->read_group_id[0] && p->read_group[0]) { str_puts(&sb, p->read_group); str_putc(&sb, '\n'); }      /* src/BWT.cpp:676-679 */
    str_puts(&sb, "@PG\tID:lordfast\tPN:lordfast\tVN:0.0.10\tCL:"); str_puts(&sb, cmdline ? cmdline : ""); str_putc(&sb, '\n');
    return sb.s;
}
