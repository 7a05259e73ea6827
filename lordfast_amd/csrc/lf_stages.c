/*
 * lf_stages.c -- stage-level batch entry points of the C ABI: chaining, edlib-equivalent alignment
 * (including the Hirschberg orchestration for problems above edlib's 1 MiB traceback switch) and ksw.
 * All DP runs in HIP kernels (lf_chain.hip, lf_align.hip); the host only orders, splits and stitches.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lf_internal.h"
#include "lf_stdsort.h"

/* compare_seed (src/Chain.cpp:227-230): by qPos only -> ties keep introsort's permutation */
#define SEED_QLESS(a, b) ((a)->qPos < (b)->qPos)
LF_DEFINE_STDSORT(seedq, Seed_t, SEED_QLESS)

void lf_sort_seeds_by_qpos(Seed_t *s, long n) { seedq_sort(s, n); }

int lf_chain_n2_batch(const lf_params_t *p, int n_windows, Seed_t *seeds, const uint64_t *off,
                      uint32_t *chain_idx, uint32_t *chain_len, float *score, int device)
{
    if (!p || n_windows < 0) { lf_set_error("lf_chain_n2_batch: bad argument"); return LF_ERR_ARG; }
    if (p->chain_alg != 0) { lf_set_error("lf_chain_n2_batch: only dp-n2 is implemented on this path"); return LF_ERR_ARG; }
    for (int w = 0; w < n_windows; w++) seedq_sort(seeds + off[w], (long)(off[w + 1] - off[w]));   /* std::sort, src/Chain.cpp:244 */
    return lfg_chain_n2(device, p, n_windows, seeds, off, chain_idx, chain_len, score, NULL);
}

int lf_chain_clasp_batch(int n_windows, const Seed_t *seeds, const uint64_t *off,
                         Seed_t *chain_out, uint32_t *chain_len, float *score, int device)
{
    if (n_windows < 0 || (n_windows && (!seeds || !off || !chain_out || !chain_len || !score))) { lf_set_error("lf_chain_clasp_batch: bad argument"); return LF_ERR_ARG; }
    return lfg_chain_clasp(device, n_windows, seeds, off, chain_out, chain_len, score, NULL);
}

int lf_ksw_extend2_batch(int n, const uint8_t *q, const uint64_t *qoff, const uint8_t *t, const uint64_t *toff,
                         const int32_t *prm, int32_t *score, int32_t *qle, int32_t *tle, int device)
{
    if (n < 0) { lf_set_error("lf_ksw_extend2_batch: bad argument"); return LF_ERR_ARG; }
    for (int i = 0; i < n; i++) if (prm[7 * i + 6] <= 0 || prm[7 * i + 1] <= 0 || prm[7 * i + 3] <= 0) {
        lf_set_error("lf_ksw_extend2_batch: h0, e_del, e_ins must be positive (lib/bwa/ksw.c:385)"); return LF_ERR_ARG;
    }
    return lfg_ksw(device, n, q, qoff, t, toff, prm, score, qle, tle, NULL);
}

/* ------------------------------------------------------------------------------------------------
 * edlib: leaf problems go straight to the kernel; larger ones follow obtainAlignmentHirschberg
 * (lib/edlib/edlib.cpp:1161-1330): two column-score kernels per split, split row = first match in the
 * order rows 0..n-2, then -1, then n-1 (:1263-1289), children re-decide leaf/split by their own size.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int prob;                 /* owning problem */
    uint64_t qo, to;          /* absolute offsets into q / t */
    uint32_t n, m;
    int best;
    int left, right;          /* children (node indices) or -1 */
    uint8_t *ops; uint32_t nops;
    int state;                /* 0 new, 1 waiting for path, 2 waiting for columns, 3 done */
} hnode_t;

typedef struct { hnode_t *v; int n, cap; } hvec_t;
static int hv_push(hvec_t *h, hnode_t x)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->v = (hnode_t *)realloc(h->v, (size_t)h->cap * sizeof(hnode_t)); }
    h->v[h->n] = x;
    return h->n++;
}

static uint32_t emit_tree(const hvec_t *h, int node, uint8_t *out)
{
    const hnode_t *x = &h->v[node];
    if (x->left < 0) { memcpy(out, x->ops, x->nops); return x->nops; }
    uint32_t a = emit_tree(h, x->left, out);
    return a + emit_tree(h, x->right, out + a);
}

int lf_edlib_solve(int device, int n, const char *q, const uint64_t *qoff, const char *t, const uint64_t *toff,
                   const uint8_t *mode, int32_t *ed, int32_t *endloc, uint8_t *ops, uint32_t *ops_len,
                   float *kernel_ms, uint64_t *n_launch_rounds)
{
    int rc = LF_OK;
    float ms = 0, ms_acc = 0;
    uint8_t *task = (uint8_t *)malloc((size_t)n + 1);
    int n_big = 0;
    for (int i = 0; i < n; i++) {
        int64_t nn = (int64_t)(qoff[i + 1] - qoff[i]), mm = (int64_t)(toff[i + 1] - toff[i]);
        /* the kernels run edlib's Hirschberg recursion themselves for queries up to LF_SWEEP_MAX_N; only longer ones
         * are split from here (distance first, then column-score requests level by level) */
        task[i] = (nn <= LF_SWEEP_MAX_N || lf_is_leaf(nn, mm)) ? LF_TASK_PATH : LF_TASK_DIST;
        n_big += task[i] == LF_TASK_DIST;
    }
    rc = lfg_edlib(device, n, q, qoff, t, toff, mode, task, ed, endloc, ops, ops_len, &ms);
    ms_acc += ms;
    if (n_launch_rounds) (*n_launch_rounds)++;
    if (rc != LF_OK || n_big == 0) { free(task); if (kernel_ms) *kernel_ms = ms_acc; return rc; }

    /* roots of the large problems: NW path of q vs t[0..end] with the known distance */
    hvec_t H = { 0, 0, 0 };
    int *root = (int *)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; i++) {
        root[i] = -1;
        if (task[i] != LF_TASK_DIST) continue;
        hnode_t x; memset(&x, 0, sizeof x);
        x.prob = i; x.qo = qoff[i]; x.to = toff[i]; x.n = (uint32_t)(qoff[i + 1] - qoff[i]);
        x.m = (uint32_t)(endloc[i] + 1); x.best = ed[i]; x.left = x.right = -1;
        root[i] = hv_push(&H, x);
    }
    for (;;) {
        /* collect this round's requests */
        int np = 0, nc = 0;
        for (int k = 0; k < H.n; k++) if (H.v[k].state == 0) {
            hnode_t *x = &H.v[k];
            if (x->n == 0 || x->m == 0 || lf_is_leaf(x->n, x->m)) { x->state = 1; np++; }
            else { x->state = 2; nc++; }
        }
        if (np == 0 && nc == 0) break;
        if (n_launch_rounds) (*n_launch_rounds)++;
        if (np) {
            /* leaf paths: a sub-batch over copies of the substrings */
            uint64_t *qo = (uint64_t *)calloc((size_t)np + 1, 8), *to = (uint64_t *)calloc((size_t)np + 1, 8);
            int *who = (int *)malloc((size_t)np * sizeof(int));
            int j = 0;
            for (int k = 0; k < H.n; k++) if (H.v[k].state == 1) { who[j] = k; qo[j + 1] = qo[j] + H.v[k].n; to[j + 1] = to[j] + H.v[k].m; j++; }
            char *qb = (char *)malloc(qo[np] + 1), *tb = (char *)malloc(to[np] + 1);
            for (j = 0; j < np; j++) { const hnode_t *x = &H.v[who[j]]; memcpy(qb + qo[j], q + x->qo, x->n); memcpy(tb + to[j], t + x->to, x->m); }
            int32_t *e2 = (int32_t *)malloc((size_t)np * 4), *l2 = (int32_t *)malloc((size_t)np * 4);
            uint32_t *len2 = (uint32_t *)malloc((size_t)np * 4);
            uint8_t *ops2 = (uint8_t *)malloc(qo[np] + to[np] + 1);
            rc = lfg_edlib(device, np, qb, qo, tb, to, NULL, NULL, e2, l2, ops2, len2, &ms);
            ms_acc += ms;
            if (rc == LF_OK) for (j = 0; j < np; j++) {
                hnode_t *x = &H.v[who[j]];
                if (e2[j] != x->best) { lf_set_error("hirschberg: leaf distance %d != expected %d", e2[j], x->best); rc = LF_ERR_HIP; break; }
                x->nops = len2[j]; x->ops = (uint8_t *)malloc(x->nops + 1);
                memcpy(x->ops, ops2 + qo[j] + to[j] + (x->n + x->m - x->nops), x->nops);     /* kernel output is end-aligned */
                x->state = 3;
            }
            free(qo); free(to); free(who); free(qb); free(tb); free(e2); free(l2); free(len2); free(ops2);
            if (rc != LF_OK) break;
        }
        if (nc) {
            /* two column-score requests per node: forward over the left half, backward over the right half */
            const int nr = 2 * nc;
            uint64_t *qo = (uint64_t *)calloc((size_t)nr + 1, 8), *to = (uint64_t *)calloc((size_t)nr + 1, 8), *co = (uint64_t *)calloc((size_t)nr + 1, 8);
            uint8_t *rev = (uint8_t *)calloc((size_t)nr, 1);
            int *who = (int *)malloc((size_t)nc * sizeof(int));
            int j = 0;
            for (int k = 0; k < H.n; k++) if (H.v[k].state == 2) {
                const hnode_t *x = &H.v[k];
                const uint32_t lw = x->m / 2, rw = x->m - lw;
                who[j / 2] = k;
                qo[j + 1] = qo[j] + x->n; to[j + 1] = to[j] + lw; co[j + 1] = co[j] + x->n + 1; rev[j] = 0; j++;
                qo[j + 1] = qo[j] + x->n; to[j + 1] = to[j] + rw; co[j + 1] = co[j] + x->n + 1; rev[j] = 1; j++;
            }
            char *qb = (char *)malloc(qo[nr] + 1), *tb = (char *)malloc(to[nr] + 1);
            for (j = 0; j < nc; j++) {
                const hnode_t *x = &H.v[who[j]];
                const uint32_t lw = x->m / 2;
                memcpy(qb + qo[2 * j], q + x->qo, x->n); memcpy(tb + to[2 * j], t + x->to, lw);
                memcpy(qb + qo[2 * j + 1], q + x->qo, x->n); memcpy(tb + to[2 * j + 1], t + x->to + lw, x->m - lw);
            }
            int32_t *cols = (int32_t *)malloc((co[nr] + 1) * 4);
            rc = lfg_colscores(device, nr, qb, qo, tb, to, rev, cols, co, &ms);
            ms_acc += ms;
            if (rc == LF_OK) for (j = 0; j < nc; j++) {
                const int k = who[j];
                const uint32_t nn = H.v[k].n, mm = H.v[k].m, lw = mm / 2, rw = mm - lw;
                const int best = H.v[k].best;
                const int32_t *F = cols + co[2 * j];        /* F[r] = dist(q[0..r), t[0..lw))        */
                const int32_t *R = cols + co[2 * j + 1];    /* R[x] = dist(last x of q, t[lw..m))    */
                int split = -2, ls = 0, rs = 0;
                for (int qi = 0; qi + 2 <= (int)nn; qi++)
                    if (F[qi + 1] + R[nn - qi - 1] == best) { split = qi; ls = F[qi + 1]; rs = R[nn - qi - 1]; break; }
                if (split == -2 && (int)lw + R[nn] == best) { split = -1; ls = (int)lw; rs = R[nn]; }
                if (split == -2 && F[nn] + (int)rw == best) { split = (int)nn - 1; ls = F[nn]; rs = (int)rw; }
                if (split == -2) { lf_set_error("hirschberg: no split row found"); rc = LF_ERR_HIP; break; }
                const uint32_t ul = (uint32_t)(split + 1);
                hnode_t a, b; memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
                a.prob = b.prob = H.v[k].prob; a.left = a.right = b.left = b.right = -1;
                a.qo = H.v[k].qo; a.to = H.v[k].to; a.n = ul; a.m = lw; a.best = ls;
                b.qo = H.v[k].qo + ul; b.to = H.v[k].to + lw; b.n = nn - ul; b.m = rw; b.best = rs;
                const int ia = hv_push(&H, a), ib = hv_push(&H, b);
                H.v[k].left = ia; H.v[k].right = ib; H.v[k].state = 3;
            }
            free(qo); free(to); free(co); free(rev); free(who); free(qb); free(tb); free(cols);
            if (rc != LF_OK) break;
        }
    }
    if (rc == LF_OK) for (int i = 0; i < n; i++) if (root[i] >= 0) {
        uint8_t *o = ops + qoff[i] + toff[i];
        const uint64_t cap = (qoff[i + 1] - qoff[i]) + (toff[i + 1] - toff[i]);
        ops_len[i] = emit_tree(&H, root[i], o);
        memmove(o + cap - ops_len[i], o, ops_len[i]);                                       /* same convention: end-aligned */
    }
    for (int k = 0; k < H.n; k++) free(H.v[k].ops);
    free(H.v); free(root); free(task);
    if (kernel_ms) *kernel_ms = ms_acc;
    return rc;
}

int lf_edlib_batch(int n, const char *q, const uint64_t *qoff, const char *t, const uint64_t *toff,
                   const uint8_t *mode, int32_t *edit_distance, int32_t *end_location,
                   uint8_t *ops, uint32_t *ops_len, int device, float *kernel_ms)
{
    if (n < 0 || !qoff || !toff) { lf_set_error("lf_edlib_batch: bad argument"); return LF_ERR_ARG; }
    if (mode) for (int i = 0; i < n; i++) if (mode[i] > 1) { lf_set_error("lf_edlib_batch: only NW (0) and SHW (1) are implemented"); return LF_ERR_ARG; }
    int rc = lf_edlib_solve(device, n, q, qoff, t, toff, mode, edit_distance, end_location, ops, ops_len, kernel_ms, NULL);
    if (rc != LF_OK) return rc;
    /* the kernels leave each ops run END-aligned in its region; the public contract is start-aligned */
    for (int i = 0; i < n; i++) {
        uint8_t *o = ops + qoff[i] + toff[i];
        const uint64_t cap = (qoff[i + 1] - qoff[i]) + (toff[i + 1] - toff[i]);
        if (ops_len[i] && ops_len[i] < cap) memmove(o, o + cap - ops_len[i], ops_len[i]);
    }
    return LF_OK;
}
