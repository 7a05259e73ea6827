/* lf_sched.c -- lanes and batches of lf_map_batch: the persistent worker pool, the lane allocator, the order of a batch's chunks
 * and the place of their SAM text in the output, and the lf_map_batch* entry points (the reference's pthread pool takes reads
 * from a shared cursor the same way, src/LordFAST.cpp:295-303).  What a chunk goes through is lf_pipeline.c. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include "lf_pipe.h"
#include "lf_batch.h"
#include <errno.h>

volatile unsigned g_crosscheck = 0;
lf_xc_hooks_t g_xc;
void lf_xc_register(const lf_xc_hooks_t *h) { if (h) g_xc = *h; }
/* the cross-check implementations live in liblfxcheck.so next to this library (tests / bench only): loaded the first time a non-zero mask is asked for */
unsigned lf_debug_crosscheck(unsigned mask)
{
    const unsigned old = g_crosscheck;
    mask &= 15u;
    if (mask && !g_xc.vote_chain) {
        Dl_info di; char path[4096];
        if (dladdr((void *)lf_debug_crosscheck, &di) && di.dli_fname) {
            snprintf(path, sizeof path, "%s", di.dli_fname);
            char *sl = strrchr(path, '/'); if (sl) sl[1] = 0; else path[0] = 0;
            strncat(path, "liblfxcheck.so", sizeof path - strlen(path) - 1);
            void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
            if (h) { void (*inst)(void) = (void (*)(void))dlsym(h, "lf_xcheck_install"); if (inst) inst(); }
        }
        if (!g_xc.vote_chain) { lf_set_error("lf_debug_crosscheck: the cross-check library (liblfxcheck.so, a test artefact) is not beside liblfgpu.so"); return ~0u; }
    }
    g_crosscheck = mask;
    return old;
}
double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
arena_t g_arena[LF_MAX_LANES][260];
const char *volatile g_lane_mark[LF_MAX_LANES];
static pthread_once_t g_rc_once = PTHREAD_ONCE_INIT;

/* ---------------------------------------------------------------- parallel for on a persistent thread pool
 * Up to eight chunks ("lanes") are in flight at once so that the host phases of one overlap the GPU phases of the others.
 * The pool therefore serves one job per lane concurrently; each lane's driver thread also works on its own job.
 * Worker ids: pool threads 0..nw-1, lane drivers nw..nw+lanes-1 (per-worker scratch arrays have nw+lanes entries). */
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
static int g_phase_n; int g_phase_on;
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
void parallel_for_named(ctx_t *cx, int n, pf_fn fn, const char *name)
{
    pool_t *P = &g_pool;
    if (n <= 0) return;
    pjob_t *J = &P->job[cx->lane];
    const int self = P->nw + cx->lane;
    const double w0 = g_phase_on ? now_ms() : 0;
    if (n == 1) {                                           /* ONE item (the replay of a chunk's one rare chain): nothing to spread; two to four items used to run here too, but an
                                                             * item can be the replay of a 50 - 100 kb read with -n 30 chains (C4 / C5), and the lane driver holding a GPU lane was serialized behind them */
        const long long c0 = g_phase_on ? thread_cpu_ns() : 0;
        for (int i = 0; i < n; i++) fn(cx, self, i);
        if (g_phase_on) phase_account(name, (thread_cpu_ns() - c0) / 1e6, now_ms() - w0);
        return;
    }
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

static void phase_strlen(ctx_t *cx, int tid, int i) { (void)tid; cx->len_out[i] = (uint32_t)strlen(cx->len_seqs[i]); }
static void phase_checklen(ctx_t *cx, int tid, int i)
{
    (void)tid;
    const char *s = cx->len_seqs[i]; const uint32_t l = cx->len_out[i];
    if (s[l] != 0 || (l > 0 && s[l - 1] == 0)) __sync_lock_test_and_set(&cx->len_bad, i);
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
    const lf_prepack_t *pre;                    /* lf_map_batch_from: the batch's planes were made when it was created (lf_batch.h) */
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
    double t0;                                  /* LF_TIMING: the batch's start */
} batch_t;

#ifndef LF_CHUNK_RAMP_DEFAULT
#define LF_CHUNK_RAMP_DEFAULT 0.25
#endif
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
    d->hirsch_bytes += a->hirsch_bytes; d->n_host_waits += a->n_host_waits; d->n_chunks += a->n_chunks; d->n_stale_first_windows += a->n_stale_first_windows;
    if (a->hirsch_max_rows > d->hirsch_max_rows) d->hirsch_max_rows = a->hirsch_max_rows;
    d->hirsch_banded_nodes += a->hirsch_banded_nodes; d->hirsch_unbanded_nodes += a->hirsch_unbanded_nodes;
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
static void fetch_dev_sam(batch_t *B, const lf_index_t *ix, uint64_t base, uint64_t tot, int parity, lf_stats_t *st, fill_t *fill, int n_fill, int lane)
{
    if (B->rc != LF_OK) { free(fill); return; }
    const double t0 = now_ms();
    static int timing = -1; if (timing < 0) timing = lf_env_set("LF_TIMING");
    out_reserve(B, base, tot);
    if (B->rc == LF_OK) {               /* one D2H copy of the chunk's text straight into its place */
        /* a caller-provided buffer never moves: the copy runs behind the lane's back (lfg_sam_fetch_wait at the lane's end);
         * a growable one may be reallocated by another lane, so the copy completes under the read lock */
        /* (A chunk's place in the output is known when every earlier chunk has its size, so the chunks behind a late one are released together and their
         * scatter kernels share the link.  Letting them go one at a time measured equal -- 86.8 / 91.4 against 86.8 / 87.4 ms per step, profiles/r04_waits/ --
         * and is not in the tree.) */
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
    if (timing) fprintf(stderr, "[lf] lane %d egress: %.1f MB placed at offset %.1f MB; issued at t=%.1f ms of the batch, host fill done at t=%.1f\n", lane, tot / 1e6, base / 1e6, t0 - B->t0, now_ms() - B->t0);
}

static void *lane_main(void *arg_)
{
    batch_t *B = (batch_t *)((void **)arg_)[0];
    const int held = (int)(intptr_t)((void **)arg_)[1];
    const int lane = held ? held - 1 : lane_acquire(B->lane_cap, &B->next_chunk, B->n_chunks0);
    if (lane < 0) return NULL;
    const int timing = lf_env_set("LF_TIMING");
    lfg_set_lane(lane);
    const long long lane_c0 = g_phase_on ? thread_cpu_ns() : 0;
    lf_stats_t *st = &B->st[lane];
    uint64_t max_hits = 1ull << 30;
    if (lf_env_set("LF_MAX_CHUNK_HITS")) { max_hits = strtoull(lf_env("LF_MAX_CHUNK_HITS"), NULL, 10); if (max_hits < 1) max_hits = 1; }   /* test hook */
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
        cx.pre = B->pre; cx.pre_i0 = C->i0;
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
        (void)lfg_take_waits();
        int rc = map_chunk(&cx);
        st->n_host_waits += lfg_take_waits(); st->n_chunks += 1;
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
        if (timing) fprintf(stderr, "[lf] lane %d chunk %d (%d reads): map_chunk %.1f ms, done at t=%.1f ms of the batch\n", lane, k, cx.n_reads, now_ms() - tch, now_ms() - B->t0);
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
                    parallel_for(&cx, cx.n_reads, g_xc.sam_print);
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
        if (timing) fprintf(stderr, "[lf] lane %d: its scatter kernels done at t=%.1f ms of the batch (waited %.1f ms)\n", lane, now_ms() - B->t0, now_ms() - t0);
    }
    if (g_phase_on) phase_account("(lane driver threads, incl. their share of the phases)", (thread_cpu_ns() - lane_c0) / 1e6, 0);
    lane_release(lane);
    return NULL;
}

/* (sleeps on a condition variable: the batch's end wakes it -- a plain 50 ms sleep made every batch that ran under LF_WATCHDOG last a multiple of 50 ms) */
typedef struct { int stop; int limit_s; pthread_mutex_t mu; pthread_cond_t cv; } wdog_t;
static void *wdog_main(void *arg)
{
    wdog_t *w = (wdog_t *)arg;
    struct timespec dl; clock_gettime(CLOCK_REALTIME, &dl); dl.tv_sec += w->limit_s;
    pthread_mutex_lock(&w->mu);
    while (!w->stop) {
        if (pthread_cond_timedwait(&w->cv, &w->mu, &dl) == ETIMEDOUT && !w->stop) {
            fprintf(stderr, "[lf watchdog] batch still running after %d s\n", w->limit_s);
            for (int l = 0; l < LF_MAX_LANES; l++) if (g_lane_mark[l]) fprintf(stderr, "[lf watchdog] lane %d: last stage mark %s\n", l, g_lane_mark[l]);
            lfg_phase_dump();
            fflush(stderr);
            abort();
        }
    }
    pthread_mutex_unlock(&w->mu);
    return NULL;
}

/* One chunk per lane (reads resident in HBM, or a pinned host batch): the chunks are cut by BASES -- equal shares (a lane's time goes with its bases, not
 * its reads), times the linear ramp for host batches (chunk k about 1 + ramp (2 k / (L - 1) - 1) times the mean: the lanes finish in chunk order) -- and
 * there are as many as lanes, or as the bounds ask for:
 *   - a lane's working set: 768 MB of bases per chunk (~ 13 bytes of HBM per base);
 *   - reads per chunk: reads x sampling positions is a 31-bit index in the seed stage and sizes its buffers (lf_seed.hip), so at most
 *     2^30 / sampling_count reads and never more than LF_CHUNK_READS_MAX -- a batch of millions of short reads is cut into more chunks, not failed
 *     ("seed batch too large") or given gigabytes of sample buffers per lane;
 *   - no chunk below 6250 reads when fewer chunks than lanes would do (a chunk's launch / sync chain does not shrink with it).
 * ends[k] = one past the last read of chunk k; returns the number of chunks (<= cap), 0 if cap is too small. */
#define LF_CHUNK_READS_MAX 65536
int lf_cut_chunks_by_bases(const uint32_t *lens, int n, int n_lanes, double ramp, int sampling_count, int *ends, int cap)
{
    if (n <= 0 || cap <= 0) return 0;
    uint64_t total_b = 0; for (int i = 0; i < n; i++) total_b += lens[i];
    int nck = n_lanes < 1 ? 1 : n_lanes; if (nck > n / 6250) nck = n / 6250 > 0 ? n / 6250 : 1;
    const uint64_t cap_b = 768ull << 20;
    if ((uint64_t)nck * cap_b < total_b) nck = (int)((total_b + cap_b - 1) / cap_b);
    long long cap_r = (1ll << 30) / (sampling_count > 0 ? sampling_count : 1); if (cap_r > LF_CHUNK_READS_MAX) cap_r = LF_CHUNK_READS_MAX; if (cap_r < 1) cap_r = 1;
    if ((long long)nck * cap_r < n) nck = (int)((n + cap_r - 1) / cap_r);
    if (nck > 1 && ramp > 0) {          /* the largest chunk of the ramp holds (1 + ramp) times the mean */
        while ((long long)((double)n / nck * (1.0 + ramp)) + 1 > cap_r && nck < n) nck++;
    }
#define CHUNK_W(k) (1.0 + (nck > 1 ? ramp * (2.0 * (k) / (nck - 1) - 1.0) : 0.0))
    double wsum = 0; for (int k = 0; k < nck; k++) wsum += CHUNK_W(k);
    int i0 = 0, out = 0; double acc_w = 0; uint64_t acc_b = 0;
    for (int k = 0; k < nck && i0 < n; k++) {
        acc_w += CHUNK_W(k);
        const uint64_t goal = k == nck - 1 ? total_b : (uint64_t)((double)total_b * acc_w / wsum);
        int i1 = i0;
        while (i1 < n && (acc_b < goal || i1 == i0) && i1 - i0 < cap_r) { acc_b += lens[i1]; i1++; }
        if (k == nck - 1 && n - i0 <= cap_r) i1 = n;
        if (out >= cap) return 0;
        ends[out++] = i1;
        i0 = i1;
    }
    while (i0 < n) {                     /* (reads the read cap pushed behind the last planned chunk) */
        int i1 = i0 + (int)cap_r; if (i1 > n) i1 = n;
        if (out >= cap) return 0;
        ends[out++] = i1; i0 = i1;
    }
#undef CHUNK_W
    return out;
}

typedef struct { const void *d_seqs, *d_quals; const uint64_t *seq_off; int dev_out; int32_t **stage_sink; } devio_t;
static __thread const lf_prepack_t *t_pre;      /* set by lf_map_batch_from around its call of map_batch_core */
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
    /* FOUR chunks in flight (round 5; eight before): with the search kernel 30 % shorter the step is the sum of its kernels sooner, four 25 k-read
     * chunks run at least as fast as eight 12.5 k ones (63.7 - 66.4 against 67.6 - 70.3 ms per 100 k reads, same box; three: 69.9, two: 74.1)
     * and cost half the launches, waits and driver threads: 0.15 - 0.17 instead of 0.34 host core-s per step (profiles/r05_lanes/) */
    /* (host batches keep eight: their step time is set by the chunk that happens to hold a read with a long replay -- every chunk behind it waits
     * for its size -- and with four 25 k-read chunks that one chunk is a quarter of the batch: 85.6 / 103.0 ms against 84.1 / 88.1 with eight,
     * same box, at equal host CPU: profiles/r05_lanes/) */
    int n_lanes = nt >= 4 ? ((dio && !dio->stage_sink) ? 4 : 8) : nt;
    if (lf_env_set("LF_LANES")) { n_lanes = (int)lf_env_long("LF_LANES", n_lanes); if (n_lanes < 1) n_lanes = 1; if (n_lanes > LF_MAX_LANES) n_lanes = LF_MAX_LANES; }
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
    if (g_active_calls == 0 || !g_pool.started) { g_phase_on = lf_env_set("LF_PHASES"); pool_ensure(nw); }
    else nw = g_pool.nw;                               /* another batch is being mapped: the pool keeps its size */
    g_active_calls++;
    pthread_mutex_unlock(&g_lanes_mu);
    const int lane_cap = n_lanes > 8 ? n_lanes : 8;
    const int lane0 = lane_acquire(lane_cap, NULL, 0); /* this thread's lane id: the set-up passes below, then its chunks */

    batch_t B; memset(&B, 0, sizeof B);
    B.t0 = T0;
    B.host_cigar = (g_crosscheck & LF_XC_HOST_CIGAR) != 0;
    B.host_vote = (g_crosscheck & LF_XC_HOST_VOTE) != 0;          /* diagnostic cross-check only; the device stage is the product path */
    B.ixs = ixs; B.n_ix = n_ix; B.p = p; B.names = names; B.seqs = seqs; B.quals = quals; B.slots = nw + LF_MAX_LANES; B.rc = LF_OK; B.lane_cap = lane_cap;
    B.pre = t_pre; t_pre = NULL;
    if (dio && dio->stage_sink) B.stage_sink = dio->stage_sink;
    else if (dio) { B.d_seqs = (const unsigned char *)dio->d_seqs; B.d_quals = (const unsigned char *)dio->d_quals; B.src_off = dio->seq_off; B.dev_out = dio->dev_out; }
    pthread_mutex_init(&B.mu, NULL); pthread_cond_init(&B.cv, NULL); pthread_rwlock_init(&B.grow, NULL);
    if (ext_buf) { B.all.s = ext_buf; B.all.cap = ext_cap; B.all.n = 0; B.all.mode = 2; B.fixed_out = 1; }
    else str_init(&B.all);
    /* SEQ-less egress: the caller's reads are host strings (we can print SEQ / QUAL ourselves) and its output buffer is pinned host
     * memory that kernels of every device can store into.  LF_SAM_FULL=1 keeps the whole line on the device (A / B measurements). */
    if (ext_buf && !dio && seqs && !lf_env_set("LF_SAM_FULL")) {
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
            if ((!dio || dio->stage_sink) && !B.pre) { c0.len_bad = -1; parallel_for(&c0, n, phase_checklen); bad = c0.len_bad; }      /* one cold cache line per read: all workers (a prepacked batch measured its reads itself) */
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
    if (lf_env_set("LF_TIMING")) fprintf(stderr, "[lf] setup (strlen + SAM buffer) %.1f ms, %d lanes, %d pool workers\n", now_ms() - T0, n_lanes, nw);
    /* chunks bound the device + host working set; reads stay in input order */
    uint64_t CHUNK_BASES = 400ull << 20;
    if (lf_env_set("LF_CHUNK_BASES")) { CHUNK_BASES = strtoull(lf_env("LF_CHUNK_BASES"), NULL, 10); if (CHUNK_BASES < 1) CHUNK_BASES = 1; }      /* measurement hook: bench.py's exclusive pass maps the whole batch as ONE chunk */
    int CHUNK_READS = 32768;
    if (lf_env_set("LF_CHUNK_READS")) { CHUNK_READS = (int)lf_env_long("LF_CHUNK_READS", CHUNK_READS); if (CHUNK_READS < 1) CHUNK_READS = 1; }   /* test hook */
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
        /* (with the lanes' uploads taking turns, lf_seed.hip: ONE chunk per lane for pinned host batches too -- 6250 / 8334 / 12500
         * reads per chunk: 99.4 / 97.5 / 96.0 ms per 100 k reads) */
        int per_lane = (dev_in || B.holes) ? 1 : 3;
        const int min_chunk = (dev_in || B.holes) ? 6250 : 1024;
        int want = (n + per_lane * n_lanes - 1) / (per_lane * n_lanes); if (want < min_chunk) want = min_chunk;
        if (want < CHUNK_READS) CHUNK_READS = want;
    }
    /* reads x sampling positions is a 31-bit index in the seed stage */
    { const long long cap = (1ll << 30) / (p->sampling_count > 0 ? p->sampling_count : 1); if (cap < CHUNK_READS) CHUNK_READS = cap < 1 ? 1 : (int)cap; }
    B.chunks = (chunk_t *)calloc((size_t)n + 1, sizeof(chunk_t));
    /* (Small first-round chunks, so that the kernels start while the bulk of the bases still crosses the link, were measured and are not in the tree: no ramp
     * 109.5 ms, a third of a chunk 115.0, 1 024 reads 111.1 -- the small chunks' fixed costs ate the gain; the lanes' uploads take turns instead, lf_seed.hip.) */
    /* Pinned host batches, one chunk per lane: the chunks GROW along the batch.  A chunk's place in the output is the sum of the sizes of
     * the chunks in front of it; with equal chunks the lanes finish together, in any order, and the SAM text of all of them crosses the
     * link at the very end (17 ms of an 88 ms step, profiles/r04_waits/).  With chunk k about (1 + ramp (2 k / (L - 1) - 1)) times the mean
     * the lanes finish in chunk order: every chunk but the last leaves while the larger ones are still being mapped. */
    double ramp = 0.0;
    const int n_ramp = n_lanes;                          /* chunks of the ramp: one per lane */
    if (B.holes && !lf_env_set("LF_CHUNK_READS") && n_lanes >= 2 && n > 2048 && (n + n_lanes - 1) / n_lanes >= 6250) ramp = LF_CHUNK_RAMP_DEFAULT;
    /* One chunk per lane (reads resident in HBM, or a pinned host batch): the chunks are cut by BASES -- equal shares (a lane's time goes with
     * its bases, not its reads), times the ramp for host batches -- and there are exactly as many as lanes (or as the working-set bound asks
     * for): with four lanes a cut by read count left a fifth, tiny chunk behind the four large ones. */
    int by_bases = 0;
    if (!lf_env_set("LF_CHUNK_READS") && !lf_env_set("LF_CHUNK_BASES") && n_lanes >= 2 && n > 2048 && ((dio && !dio->stage_sink) || B.holes)) {
        int *ends = (int *)malloc(((size_t)n + 1) * sizeof(int));
        const int nck = ends ? lf_cut_chunks_by_bases(lens, n, n_lanes, ramp, p->sampling_count, ends, n + 1) : 0;
        for (int k = 0, i0 = 0; k < nck; k++) { B.chunks[B.n_chunks].i0 = i0; B.chunks[B.n_chunks].i1 = ends[k]; B.n_chunks++; i0 = ends[k]; }
        free(ends);
        if (nck <= 0) { free(lens); free(B.chunks); lf_set_error("out of memory (chunk table)"); return LF_ERR_NOMEM; }
        by_bases = 1;
    }
    for (int i0 = 0; i0 < n && !by_bases; ) {
        int i1 = i0; uint64_t bases = 0;
        int lim = CHUNK_READS;
        if (ramp > 0) {
            const int k = B.n_chunks;
            if (k >= n_ramp - 1) lim = n - i0;                          /* the last chunk takes what is left */
            else lim = (int)((double)n / n_ramp * (1.0 + ramp * (2.0 * k / (n_ramp - 1) - 1.0)) + 0.5);
            if (lim < 1) lim = 1;
        }
        while (i1 < n && i1 - i0 < lim && bases < CHUNK_BASES) { bases += lens[i1]; i1++; }
        B.chunks[B.n_chunks].i0 = i0; B.chunks[B.n_chunks].i1 = i1; B.n_chunks++;
        i0 = i1;
    }
    B.n_chunks0 = B.n_chunks;
    /* LF_WATCHDOG=<seconds>: a batch that takes longer reports where every lane is and aborts (tests set it: a hang
     * becomes a failure with a location) */
    wdog_t wd; memset(&wd, 0, sizeof wd); pthread_t wdt; int have_wd = 0;
    if (lf_env_long("LF_WATCHDOG", 0) > 0) {
        wd.limit_s = (int)lf_env_long("LF_WATCHDOG", 0); pthread_mutex_init(&wd.mu, NULL); pthread_cond_init(&wd.cv, NULL);
        have_wd = pthread_create(&wdt, NULL, wdog_main, &wd) == 0;
    }
    void *la[LF_MAX_LANES][2]; pthread_t lt[LF_MAX_LANES]; int have[LF_MAX_LANES] = { 0 };
    for (int l = 0; l < LF_MAX_LANES; l++) { la[l][0] = &B; la[l][1] = (void *)(intptr_t)(l == 0 ? lane0 + 1 : 0); }      /* [1]: lane id + 1 already held, 0: take one */
    for (int l = 1; l < n_lanes && l < B.n_chunks; l++) have[l] = pthread_create(&lt[l], NULL, lane_main, la[l]) == 0;
    lane_main(la[0]);
    for (int l = 1; l < n_lanes; l++) if (have[l]) pthread_join(lt[l], NULL);
    if (have_wd) { pthread_mutex_lock(&wd.mu); wd.stop = 1; pthread_cond_signal(&wd.cv); pthread_mutex_unlock(&wd.mu); pthread_join(wdt, NULL); }
    lfg_set_lane(0);
    if (g_phase_on) { fprintf(stderr, "[lf] batch of %d reads: %.1f ms wall, %d threads\n", n, now_ms() - T0, nt); phase_report(); }
    pthread_mutex_lock(&g_lanes_mu); g_active_calls--; pthread_mutex_unlock(&g_lanes_mu);

    uint64_t total = 0;
    for (int k = 0; k < B.n_chunks; k++) total += B.chunks[k].size;
    for (int l = 0; l < LF_MAX_LANES; l++) merge_stats(st, &B.st[l]);
    free(lens); free(B.chunks);
    pthread_mutex_destroy(&B.mu); pthread_cond_destroy(&B.cv); pthread_rwlock_destroy(&B.grow);
    st->ms_total = now_ms() - T0;
    if (lf_env_set("LF_TIMING")) fprintf(stderr, "[lf] lf_map_batch total %.1f ms\n", st->ms_total);
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

/* ---- the mapper-ready batch (lf_batch.h, include/lordfast_amd.h) ---- */
typedef struct { struct lf_read_batch *b; lf_prepack_t *P; int n; int nthr; uint64_t xcap; volatile int overflow; } ppk_t;
static void *prepack_main(void *arg)
{
    void **a = (void **)arg; ppk_t *K = (ppk_t *)a[0]; const int t = (int)(intptr_t)a[1];
    /* reads in contiguous shares by index: neighbouring threads meet in at most one plane word per share boundary (lf_pack_read ORs its first and last word in atomically) */
    const int i0 = (int)((int64_t)K->n * t / K->nthr), i1 = (int)((int64_t)K->n * (t + 1) / K->nthr);
    for (int i = i0; i < i1; i++) {
        if ((int)K->b->lens[i] < K->P->min_read_len) continue;
        if (!lf_pack_read(K->P->planes, K->P->QW, K->P->boff[i], K->b->seqs[i], K->b->lens[i], K->P->exc_pos, K->P->exc_byte, K->xcap, &K->P->n_exc)) K->overflow = 1;
    }
    return NULL;
}
static int exc_cmp(const void *x, const void *y) { const uint64_t a = ((const uint64_t *)x)[0], b = ((const uint64_t *)y)[0]; return a < b ? -1 : a > b; }
void lf_prepack_free(lf_prepack_t *P)
{
    if (!P) return;
    if (P->pinned) lfg_host_free(P->planes); else free(P->planes);
    free(P->boff); free(P->exc_pos); free(P->exc_byte); free(P);
}
int lf_read_batch_prepack(struct lf_read_batch *b, int min_read_len, int threads)
{
    if (!b || min_read_len < 1) { lf_set_error("lf_read_batch_prepack: bad argument"); return LF_ERR_ARG; }
    if (b->pre && b->pre->min_read_len == min_read_len) return LF_OK;
    lf_prepack_free(b->pre); b->pre = NULL;
    lf_prepack_t *P = (lf_prepack_t *)calloc(1, sizeof *P);
    if (!P) return LF_ERR_NOMEM;
    P->min_read_len = min_read_len;
    P->boff = (uint64_t *)malloc(((size_t)b->n + 1) * 8);
    if (!P->boff) { lf_prepack_free(P); return LF_ERR_NOMEM; }
    uint64_t o = 0;
    for (int i = 0; i < b->n; i++) { P->boff[i] = o; if ((int)b->lens[i] >= min_read_len) o += b->lens[i]; }
    P->boff[b->n] = o; P->bases = o;
    P->QW = (o + 63) / 64 + 8;
    const size_t bytes = 3 * P->QW * 8;
    P->planes = lfg_device_count() > 0 ? (uint64_t *)lfg_host_alloc(bytes) : NULL;
    P->pinned = P->planes != NULL;
    if (!P->planes) P->planes = (uint64_t *)malloc(bytes);
    if (!P->planes) { lf_prepack_free(P); return LF_ERR_NOMEM; }
    memset(P->planes, 0, bytes);
    uint64_t xcap = o / 64 + 4096;
    for (int attempt = 0; attempt < 2; attempt++) {
        P->exc_pos = (uint64_t *)malloc(xcap * 8); P->exc_byte = (uint8_t *)malloc(xcap); P->n_exc = 0;
        if (!P->exc_pos || !P->exc_byte) { lf_prepack_free(P); return LF_ERR_NOMEM; }
        int nthr = threads; long online = sysconf(_SC_NPROCESSORS_ONLN);
        if (nthr <= 0 || nthr > online) nthr = (int)online;
        if (nthr > 64) nthr = 64; if (nthr > b->n) nthr = b->n > 0 ? b->n : 1;
        ppk_t K; K.b = b; K.P = P; K.n = b->n; K.nthr = nthr; K.xcap = xcap; K.overflow = 0;
        pthread_t th[64]; void *args[64][2]; int started[64];
        for (int t = 0; t < nthr; t++) { args[t][0] = &K; args[t][1] = (void *)(intptr_t)t; started[t] = t > 0 && pthread_create(&th[t], NULL, prepack_main, args[t]) == 0; }
        prepack_main(args[0]);
        for (int t = 1; t < nthr; t++) { if (started[t]) pthread_join(th[t], NULL); else prepack_main(args[t]); }
        if (!K.overflow) break;
        /* more bytes outside ACGT than one in 64 (lower-case reads): a list for all of them */
        free(P->exc_pos); free(P->exc_byte); P->exc_pos = NULL; P->exc_byte = NULL;
        if (attempt == 1) { lf_prepack_free(P); lf_set_error("lf_read_batch_prepack: exception list overflow"); return LF_ERR_NOMEM; }
        xcap = o + 64; memset(P->planes, 0, bytes);
    }
    if (P->n_exc > 1) {          /* ascending positions: a chunk's exceptions are a contiguous range */
        uint64_t *pair = (uint64_t *)malloc(P->n_exc * 16);
        if (!pair) { lf_prepack_free(P); return LF_ERR_NOMEM; }
        for (uint64_t k = 0; k < P->n_exc; k++) { pair[2 * k] = P->exc_pos[k]; pair[2 * k + 1] = P->exc_byte[k]; }
        qsort(pair, P->n_exc, 16, exc_cmp);
        for (uint64_t k = 0; k < P->n_exc; k++) { P->exc_pos[k] = pair[2 * k]; P->exc_byte[k] = (uint8_t)pair[2 * k + 1]; }
        free(pair);
    }
    b->pre = P;
    return LF_OK;
}
lf_read_batch_t *lf_batch_create(int n, const char *const *names, const char *const *seqs, const char *const *quals, const uint32_t *seq_lens, int min_read_len, int threads)
{
    if (n < 0 || (n > 0 && (!names || !seqs))) { lf_set_error("lf_batch_create: bad argument"); return NULL; }
    struct lf_read_batch *b = (struct lf_read_batch *)calloc(1, sizeof *b);
    if (!b) { lf_set_error("out of memory"); return NULL; }
    b->n = n; b->rcap = n;
    b->names = (const char **)malloc(((size_t)n + 1) * sizeof(char *)); b->seqs = (const char **)malloc(((size_t)n + 1) * sizeof(char *));
    b->quals = (const char **)malloc(((size_t)n + 1) * sizeof(char *)); b->lens = (uint32_t *)malloc(((size_t)n + 1) * 4);
    if (!b->names || !b->seqs || !b->quals || !b->lens) { lf_read_batch_free(b); lf_set_error("out of memory"); return NULL; }
    for (int i = 0; i < n; i++) {
        b->names[i] = names[i]; b->seqs[i] = seqs[i]; b->quals[i] = (quals && quals[i]) ? quals[i] : "";
        b->lens[i] = seq_lens ? seq_lens[i] : (uint32_t)strlen(seqs[i]);
        b->bases += b->lens[i];
    }
    if (lf_read_batch_prepack(b, min_read_len > 0 ? min_read_len : 1000, threads) != LF_OK) { lf_read_batch_free(b); return NULL; }
    return b;
}
void lf_batch_free(lf_read_batch_t *b) { lf_read_batch_free(b); }
int lf_batch_size(const lf_read_batch_t *b) { return b ? b->n : 0; }
int lf_map_batch_from(const lf_index_t *ix, const lf_params_t *p, const lf_read_batch_t *b, char *out, size_t out_cap, size_t *sam_len, lf_stats_t *stats)
{
    if (!b || !p) { lf_set_error("lf_map_batch_from: no batch"); return LF_ERR_ARG; }
    if (!out || out_cap < 2) { lf_set_error("lf_map_batch_from: no output buffer"); return LF_ERR_ARG; }
    /* the planes serve when they were made for this -l (reads below it are not in them); otherwise the call packs like lf_map_batch_into_lens */
    t_pre = (b->pre && b->pre->min_read_len == p->min_read_len) ? b->pre : NULL;
    const int rc = map_batch_core(&ix, 1, p, b->n, b->names, b->seqs, b->quals, b->lens, out, out_cap, NULL, sam_len, stats, NULL);
    t_pre = NULL;
    return rc;
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

