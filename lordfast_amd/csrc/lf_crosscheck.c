/* lf_crosscheck.c -- TEST LIBRARY (liblfxcheck.so), not part of liblfgpu.so: host re-implementations of device stages, kept as CROSS-CHECKS for the tests.  They are reached through
 * lf_debug_crosscheck() only (bit 0: vote / selection / std::sort, bit 1: CIGAR / MD strings -- the STREAM / TRACK builders live
 * with the replay in lf_replay.c --, bit 3: SAM line assembly) and work on data copied back from the device; nothing in a
 * production environment can select them. */
#include "lf_pipe.h"

void lf_sort_seeds_by_qpos(Seed_t *s, long n);
int  crosscheck_vote_chain(ctx_t *cx);
void phase_bind_text(ctx_t *cx, int tid, int ri);
void phase_sam_print(ctx_t *cx, int tid, int ri);

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

/* B + C with the host vote: per-worker chain requests are merged, the chains themselves still run on the device */
int crosscheck_vote_chain(ctx_t *cx)
{
    const int n = cx->n_reads, nt = cx->n_threads;
    lf_stats_t *st = cx->st;
    int rc = LF_OK;
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
    return LF_OK;
}

/* the records' strings are in the rendered text now */
void phase_bind_text(ctx_t *cx, int tid, int ri)
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

void phase_sam_print(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (cx->out_base) { r->out.s = cx->out_base + cx->out_off[ri]; r->out.cap = r->out.n; r->out.n = 0; r->out.mode = 2; }
    else { r->out.s = NULL; r->out.cap = 0; r->out.n = 0; r->out.mode = 1; }
    print_sam_entry(cx, r, r->mode < 2 ? 1 : (r->mode == 2 ? 1 : r->nWins));
}


/* liblfxcheck.so's entry point: lf_debug_crosscheck() (liblfgpu.so) calls it once */
void lf_xcheck_install(void)
{
    lf_xc_hooks_t h; h.vote_chain = crosscheck_vote_chain; h.bind_text = phase_bind_text; h.sam_print = phase_sam_print;
    lf_xc_register(&h);
}
