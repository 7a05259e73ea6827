/*
 * lfo_clasp.c -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).  Not part of the product.
 *
 * Plain-C restatement of `chain_seeds_clasp` (src/Chain.cpp:39-209) and of the part of the clasp
 * library it drives with chainmode = SOP, lambda = 0.15, eps = 0, maxgap = -1:
 *   bl_slClusterSop      lib/clasp/slchain.c:568-655
 *   bl_slChainSop        lib/clasp/slchain.c:668-826
 *   bl_slChainSopRMQ     lib/clasp/slchain.c:841-912
 *   bl_slChainSopActivate lib/clasp/slchain.c:924-974
 *   bl_slExtractPoints / bl_slGetTrans / comparators   lib/clasp/slchain.c:49-90, 369-404, 412-559
 *   quickSort            lib/clasp/sort.c:164-225        (deterministic, unstable: replayed literally)
 *   range tree shape     lib/clasp/rangetree.c:204-330   (balanced over the first-dimension ranks)
 * The associated structure of a range-tree node (a van-Emde-Boas tree in the reference build,
 * lib/clasp/vebtree.c) is restated as what it is used as: an ordered map  y-rank -> chain  with
 * pred / succ / insert / delete.  This file keeps that map literally (a sorted array per node) and
 * replays inserts and deletes in time order; the GPU code uses an order-free formulation instead, so
 * the two are independent.
 *
 * Parity status: PINNED against the compiled reference (`ref_chain_clasp`, tests/test_oracle_vs_ref.py).
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "lf_oracle.h"

static const double LAMBDA = 0.15, EPS = 0.0;      /* src/Chain.cpp:52-53 */

/* slmatch_t fields that are used (lib/clasp/sltypes.h:36-56) */
typedef struct { int i, j; long p, q; double scr; uint32_t orig; } cfrag_t;
#define FS_S(f) ((f)->p)
#define FE_S(f) ((f)->p + (f)->q - 1)
#define FS_Q(f) ((f)->i)
#define FE_Q(f) ((f)->i + (f)->j - 1)

/* D(a,b) of lib/clasp/slchain.h:40 */
static long dl(long a, long b) { return (a > b) ? (a - b - 1) : (b > a) ? (b - a - 1) : 1; }
static int  di(int a, int b)   { return (a > b) ? (a - b - 1) : (b > a) ? (b - a - 1) : 1; }

/* GSOP(fprim, f) (lib/clasp/slchain.h:45-47): fprim's start against f's end */
static double gsop(const cfrag_t *cur, long fend_s, int fend_q)
{
    long dx = dl(FS_S(cur), fend_s);
    int  dy = di(FS_Q(cur), fend_q);
    return (dx >= dy) ? (LAMBDA * dx + (EPS - LAMBDA) * dy) : (LAMBDA * dy + (EPS - LAMBDA) * dx);
}

/* ---- quickSort (lib/clasp/sort.c:164-225): returns the sorted index array ---- */
typedef unsigned (*ccmp_fn)(uint32_t a, uint32_t b, const void *data);

static uint32_t *clasp_quicksort(const void *data, uint32_t size, ccmp_fn cmp)
{
    uint32_t *sorted = (uint32_t *)malloc(sizeof(uint32_t) * (size ? size : 1));
    for (uint32_t k = 0; k < size; k++) sorted[k] = k;
    int cap = 64, top = 0;
    int (*st)[2] = malloc(sizeof(int[2]) * cap);
    st[top][0] = 0; st[top][1] = (int)size - 1; top++;
    while (top > 0) {
        top--;
        int left = st[top][0], right = st[top][1];
        while (left < right) {
            uint32_t x = sorted[(left + right) / 2];
            int l2 = left, r2 = right;
            do {
                while (cmp(sorted[l2], x, data) == 2) l2++;
                while (cmp(sorted[r2], x, data) == 1) r2--;
                if (l2 <= r2) { uint32_t t = sorted[r2]; sorted[r2] = sorted[l2]; sorted[l2] = t; l2++; r2--; }
            } while (r2 >= l2);
            if (top == cap) { cap *= 2; st = realloc(st, sizeof(int[2]) * cap); }
            if ((l2 - left) > (right - l2)) { st[top][0] = left; st[top][1] = r2; top++; left = l2; }
            else { st[top][0] = l2; st[top][1] = right; top++; right = r2; }
        }
    }
    free(st);
    return sorted;
}

/* point_t (lib/clasp/slchain.h:65-72) */
typedef struct { int x, y; uint32_t trans[4]; uint32_t index; int start; } cpoint_t;

static unsigned cmp_end(uint32_t a, uint32_t b, const void *d)
{   /* cmp_slmatch_end_quick :428-441 */
    const cfrag_t *m = (const cfrag_t *)d;
    if (FE_S(m + a) > FE_S(m + b)) return 1;
    if (FE_S(m + a) < FE_S(m + b)) return 2;
    return 0;
}
static unsigned cmp_t1x(uint32_t a, uint32_t b, const void *d)
{   /* cmp_slmatch_trans_first_x :452-473: x - y, end points first */
    const cpoint_t *m = (const cpoint_t *)d;
    int ax = m[a].x - m[a].y, bx = m[b].x - m[b].y;
    if (ax > bx) return 1;
    if (ax < bx) return 2;
    if (m[a].start > m[b].start) return 1;
    if (m[a].start < m[b].start) return 2;
    return 0;
}
static unsigned cmp_t1y(uint32_t a, uint32_t b, const void *d)
{   /* cmp_slmatch_trans_first_y :484-505: y, start points first */
    const cpoint_t *m = (const cpoint_t *)d;
    if (m[a].y > m[b].y) return 1;
    if (m[a].y < m[b].y) return 2;
    if (m[a].start < m[b].start) return 1;
    if (m[a].start > m[b].start) return 2;
    return 0;
}
static unsigned cmp_t2x(uint32_t a, uint32_t b, const void *d)
{   /* cmp_slmatch_trans_second_x :516-537: x, start points first */
    const cpoint_t *m = (const cpoint_t *)d;
    if (m[a].x > m[b].x) return 1;
    if (m[a].x < m[b].x) return 2;
    if (m[a].start < m[b].start) return 1;
    if (m[a].start > m[b].start) return 2;
    return 0;
}
static unsigned cmp_t2y(uint32_t a, uint32_t b, const void *d)
{   /* cmp_slmatch_trans_second_y :548-559: y - x, end points first */
    const cpoint_t *m = (const cpoint_t *)d;
    int ay = m[a].y - m[a].x, by = m[b].y - m[b].x;
    if (ay > by) return 1;
    if (ay < by) return 2;
    if (m[a].start > m[b].start) return 1;
    if (m[a].start < m[b].start) return 2;
    return 0;
}

/* ---- the chains ---- */
/* slchain_t as kept in the range trees: one per fragment, made at its end point (:735-790).  Its end
 * coordinates are those of its last fragment; `first` = matches[0]; `prev` = the chain it extends. */
typedef struct { double scr, prioA, prioB; int first, prev; } cchain_t;
/* match->chain: the best chain that STARTS at this fragment: chain `base` (+ fragment `extra`) */
typedef struct { int set; double scr; int base, extra; } cbest_t;

/* ---- range tree with an ordered map per node ---- */
typedef struct { uint32_t y; int chain; } cent_t;
typedef struct { uint32_t value; int left, right; cent_t *ent; int ne; } cnode_t;
typedef struct { cnode_t *nodes; int nnodes; } ctree_t;

static int tree_build(ctree_t *t, uint32_t start, uint32_t n)
{   /* bl_rangetreeInitLev (lib/clasp/rangetree.c:204-330): dimension-1 coordinates are the ranks
     * 0..N-1 themselves, left part = ceil(n/2) */
    int id = t->nnodes++;
    cnode_t *nd = &t->nodes[id];
    nd->ent = (cent_t *)malloc(sizeof(cent_t) * n); nd->ne = 0;
    if (n == 1) { nd->value = start; nd->left = nd->right = -1; return id; }
    uint32_t mid = n / 2 + (n % 2 != 0);
    nd->value = start + mid - 1;
    int l = tree_build(t, start, mid);
    int r = tree_build(t, start + mid, n - mid);
    t->nodes[id].left = l; t->nodes[id].right = r;
    return id;
}
static void tree_init(ctree_t *t, uint32_t N)
{
    t->nodes = (cnode_t *)malloc(sizeof(cnode_t) * (2 * (size_t)N)); t->nnodes = 0;
    tree_build(t, 0, N);
}
static void tree_free(ctree_t *t)
{
    for (int k = 0; k < t->nnodes; k++) free(t->nodes[k].ent);
    free(t->nodes);
}
/* index of the largest key < y, or -1 (bl_vebtreePred, lib/clasp/vebtree.c:265) */
static int ent_pred(const cnode_t *nd, uint32_t y)
{
    int lo = 0, hi = nd->ne;          /* first index with key >= y */
    while (lo < hi) { int m = (lo + hi) / 2; if (nd->ent[m].y < y) lo = m + 1; else hi = m; }
    return lo - 1;
}

typedef struct {
    const cfrag_t *frag;      /* the cluster's fragments */
    cchain_t *chain; cbest_t *best;
} cctx_t;

/* bl_slChainSopRMQ (:841-912). which = 0: tree A (prioA), 1: tree B (prioB). */
static int rmq(cctx_t *cx, ctree_t *t, int which, uint32_t x, uint32_t y, int cur)
{
    const cfrag_t *c = cx->frag + cur;
    int res = -1; double resprio = -DBL_MAX;
    int node = 0;
    while (node != -1) {
        cnode_t *nd = &t->nodes[node];
        if (nd->value <= x) {
            const cnode_t *as = (nd->left != -1) ? &t->nodes[nd->left] : nd;
            int pi = ent_pred(as, y);
            if (pi >= 0) {
                int tmp = as->ent[pi].chain;
                const cfrag_t *tf = cx->frag + tmp;          /* the chain ends with fragment tmp */
                double g = gsop(c, FE_S(tf), FE_Q(tf));
                if (c->scr >= g) {                            /* :877 */
                    cbest_t *local = &cx->best[cx->chain[tmp].first];
                    if (local->scr < c->scr + cx->chain[tmp].scr - g) {      /* :884 */
                        local->scr = cx->chain[tmp].scr + (c->scr - g);      /* :890 */
                        local->base = tmp; local->extra = cur;
                    }
                }
                double pr = which ? cx->chain[tmp].prioB : cx->chain[tmp].prioA;
                if (pr > resprio) { res = tmp; resprio = pr; }                /* :898 */
            }
            node = nd->right;
        } else node = nd->left;
    }
    return res;
}

/* bl_slChainSopActivate (:924-974) */
static void activate(cctx_t *cx, ctree_t *t, int which, int cur, uint32_t x, uint32_t y)
{
    double prio = which ? cx->chain[cur].prioB : cx->chain[cur].prioA;
    int node = 0;
    while (node != -1) {
        cnode_t *nd = &t->nodes[node];
        int pi = ent_pred(nd, y + 1);
        int ok = 1;
        if (pi >= 0) {
            int pc = nd->ent[pi].chain;
            double pp = which ? cx->chain[pc].prioB : cx->chain[pc].prioA;
            ok = (prio >= pp);
        }
        if (ok) {
            int at = pi + 1;                                  /* keys are distinct ranks */
            int del = 0;                                      /* successors with a lower priority go */
            while (at + del < nd->ne) {
                int sc = nd->ent[at + del].chain;
                double sp = which ? cx->chain[sc].prioB : cx->chain[sc].prioA;
                if (prio > sp) del++; else break;
            }
            if (del != 1) memmove(nd->ent + at + 1, nd->ent + at + del, sizeof(cent_t) * (size_t)(nd->ne - at - del));
            nd->ent[at].y = y; nd->ent[at].chain = cur;
            nd->ne += 1 - del;
        }
        node = (nd->value < x) ? nd->right : nd->left;
    }
}

/* bl_slChainSop (:668-826) over one cluster of `size` fragments (sorted by start) */
static void chain_sop(const cfrag_t *frag, uint32_t size, cchain_t *chain, cbest_t *best)
{
    /* bl_slExtractPoints :49-90 */
    uint32_t *sorted = clasp_quicksort(frag, size, cmp_end);
    uint32_t N = 2 * size, np = 0;
    cpoint_t *pt = (cpoint_t *)malloc(sizeof(cpoint_t) * N);
    uint32_t i, j = 0;
    for (i = 0; i < size; i++) {
        const cfrag_t *a = frag + i, *b = frag + sorted[j];
        while (FS_S(a) > FE_S(b)) {
            pt[np].x = (int)FE_S(b); pt[np].y = FE_Q(b); pt[np].index = sorted[j]; pt[np].start = 0; np++;
            b = frag + sorted[++j];
        }
        pt[np].x = (int)FS_S(a); pt[np].y = FS_Q(a); pt[np].index = i; pt[np].start = 1; np++;
    }
    while (j < size) {
        const cfrag_t *b = frag + sorted[j];
        pt[np].x = (int)FE_S(b); pt[np].y = FE_Q(b); pt[np].index = sorted[j]; pt[np].start = 0; np++;
        j++;
    }
    free(sorted);
    /* bl_slGetTrans :369-404 */
    ccmp_fn cmps[4] = { cmp_t1x, cmp_t1y, cmp_t2x, cmp_t2y };
    uint32_t *tr[4];
    for (int k = 0; k < 4; k++) {
        tr[k] = clasp_quicksort(pt, N, cmps[k]);
        for (uint32_t r = 0; r < N; r++) pt[tr[k][r]].trans[k] = r;
    }
    int ta = pt[tr[2][N - 1]].x, tb = pt[tr[1][N - 1]].y;     /* t.a, t.b :703-706 */
    for (int k = 0; k < 4; k++) free(tr[k]);

    ctree_t A, B; tree_init(&A, N); tree_init(&B, N);
    int *prev = (int *)malloc(sizeof(int) * size);
    for (i = 0; i < size; i++) { prev[i] = -1; best[i].set = 0; }
    cctx_t cx = { frag, chain, best };

    for (i = 0; i < N; i++) {
        const cpoint_t *P = &pt[i];
        int cur = (int)P->index;
        const cfrag_t *c = frag + cur;
        if (P->start) {
            int ap = rmq(&cx, &A, 0, P->trans[0], P->trans[1], cur);
            int bp = rmq(&cx, &B, 1, P->trans[2], P->trans[3], cur);
            int pv;
            if (ap < 0) pv = bp;
            else if (bp < 0) pv = ap;
            else {
                double ga = gsop(c, FE_S(frag + ap), FE_Q(frag + ap)), gb = gsop(c, FE_S(frag + bp), FE_Q(frag + bp));
                pv = (chain[ap].scr - ga >= chain[bp].scr - gb) ? ap : bp;      /* :727-733 */
            }
            if (pv >= 0 && chain[pv].scr < gsop(c, FE_S(frag + pv), FE_Q(frag + pv))) pv = -1;   /* :739 */
            prev[cur] = pv;
        } else {
            if (prev[cur] >= 0) {
                int cand = prev[cur];
                chain[cur].scr = c->scr + chain[cand].scr - gsop(c, FE_S(frag + cand), FE_Q(frag + cand));  /* :759 */
                chain[cur].first = chain[cand].first; chain[cur].prev = cand;
                cbest_t *fb = &best[chain[cur].first];
                if (fb->set && fb->scr <= chain[cur].scr) { fb->scr = chain[cur].scr; fb->base = cur; fb->extra = -1; }  /* :772-779 */
            } else {
                chain[cur].scr = c->scr; chain[cur].first = cur; chain[cur].prev = -1;
                best[cur].set = 1; best[cur].scr = c->scr; best[cur].base = cur; best[cur].extra = -1;       /* :783-796 */
            }
            /* GCSOP1 / GCSOP2 (lib/clasp/slchain.h:51-52) with the chain's end = this fragment's end */
            double g1 = LAMBDA * dl(ta, FE_S(c)) + (EPS - LAMBDA) * di(tb, FE_Q(c));
            double g2 = LAMBDA * di(tb, FE_Q(c)) + (EPS - LAMBDA) * dl(ta, FE_S(c));
            chain[cur].prioA = chain[cur].scr - g1;
            chain[cur].prioB = chain[cur].scr - g2;
            activate(&cx, &A, 0, cur, P->trans[0], P->trans[1]);
            activate(&cx, &B, 1, cur, P->trans[2], P->trans[3]);
        }
    }
    free(prev); free(pt); tree_free(&A); tree_free(&B);
}

/* chain_seeds_clasp (src/Chain.cpp:39-209).  `seeds` is not modified.  For n == 0 the reference sets score = -1 and leaves
 * chainLen AND the chain's seeds as its previous call left them (:68, :92): the caller then works with a STALE chain
 * (lfo_map.c keeps one chain buffer per mapping context like the reference's _pf_topChains, and clears it per read). */
void lfo_chain_clasp(const lfo_seed_t *seeds, uint32_t n, lfo_seed_t *out, uint32_t *chainLen, float *score)
{
    *score = -1;
    if (n == 0) return;
    *chainLen = 0;
    cfrag_t *f = (cfrag_t *)malloc(sizeof(cfrag_t) * n), *tmpf = (cfrag_t *)malloc(sizeof(cfrag_t) * n);
    for (uint32_t k = 0; k < n; k++) {
        f[k].p = seeds[k].tPos; f[k].i = (int)seeds[k].qPos; f[k].q = f[k].j = (int)seeds[k].len;
        f[k].scr = seeds[k].len; f[k].orig = k;
    }
    /* qsort(cmp_slmatch_qsort) (:94, slchain.c:412): glibc's qsort is a stable merge sort; key = start on the
     * reference (the int-truncated difference of two longs; positions are < 2^31 here) */
    for (uint32_t w = 1; w < n; w *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * w) {
            uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmpf[o++] = ((int)(f[b].p - f[a].p) < 0) ? f[b++] : f[a++];
            while (a < mid) tmpf[o++] = f[a++];
            while (b < hi) tmpf[o++] = f[b++];
        }
        cfrag_t *sw = f; f = tmpf; tmpf = sw;
    }
    free(tmpf);

    cchain_t *chain = (cchain_t *)malloc(sizeof(cchain_t) * n);
    cbest_t *best = (cbest_t *)malloc(sizeof(cbest_t) * n);
    uint32_t *cbeg = (uint32_t *)malloc(sizeof(uint32_t) * n);    /* cluster start of every fragment */

    /* bl_slClusterSop (slchain.c:568-655); maxgap = -1 so MAXGAP2 is false; one subject */
    {
        uint32_t begin = 0, i, length;
        int max_yst, min_yend, min_yst, max_yend, cor = 0;
        double max_spp;
        const cfrag_t *a = f, *b;
        max_spp = a->scr / (double)a->j;
        min_yst = max_yst = FS_Q(a); min_yend = max_yend = FE_Q(a);
        for (i = 0; i + 1 < n; i++) {
            b = f + (i + 1);
            if (FS_Q(b) < min_yst) min_yst = FS_Q(b);
            if (FS_Q(b) > max_yst) max_yst = FS_Q(b);
            if (FE_Q(b) < min_yend) min_yend = FE_Q(b);
            if (FE_Q(b) > max_yend) max_yend = FE_Q(b);
            if (b->scr / (double)b->j > max_spp) max_spp = b->scr / (double)b->j;
            if (LAMBDA > EPS && max_yst > min_yend) cor = (int)((EPS - LAMBDA) * (double)(max_yst - min_yend));
            if ((int)FS_S(b) > (int)FE_S(a) &&
                (int)FS_S(b) - (int)FE_S(a) >= max_yst - min_yend &&
                (double)LAMBDA * (double)dl(FS_S(b), FE_S(a)) + cor > (double)(max_yend - min_yst + 1) * max_spp) {
                length = i - begin + 1;
                for (uint32_t k = begin; k <= i; k++) cbeg[k] = begin;
                if (length == 1) {
                    chain[i].scr = f[i].scr; chain[i].first = 0; chain[i].prev = -1;
                    best[i].set = 1; best[i].scr = f[i].scr; best[i].base = 0; best[i].extra = -1;
                } else chain_sop(f + begin, length, chain + begin, best + begin);
                begin = i + 1;
            }
            a = b;
        }
        length = i - begin + 1;
        for (uint32_t k = begin; k < n; k++) cbeg[k] = begin;
        if (length == 1) {
            chain[begin].scr = f[begin].scr; chain[begin].first = 0; chain[begin].prev = -1;
            best[begin].set = 1; best[begin].scr = f[begin].scr; best[begin].base = 0; best[begin].extra = -1;
        } else chain_sop(f + begin, length, chain + begin, best + begin);
    }

    /* src/Chain.cpp:128-147: first chain with the strictly greatest score, compared with a float */
    float bestScore = -1; int bj = -1;
    for (uint32_t jx = 0; jx < n; jx++)
        if (best[jx].set && best[jx].scr > bestScore) { bestScore = (float)best[jx].scr; bj = (int)jx; }
    if (bj >= 0) {
        uint32_t cb = cbeg[bj], len = 0;
        for (int e = best[bj].base; e >= 0; e = chain[cb + e].prev) len++;
        uint32_t total = len + (best[bj].extra >= 0 ? 1 : 0), k = len;
        for (int e = best[bj].base; e >= 0; e = chain[cb + e].prev) {
            const cfrag_t *g = f + cb + e; k--;
            out[k].tPos = (uint32_t)g->p; out[k].qPos = (uint32_t)g->i; out[k].len = (uint32_t)g->j;
        }
        if (best[bj].extra >= 0) {
            const cfrag_t *g = f + cb + best[bj].extra;
            out[len].tPos = (uint32_t)g->p; out[len].qPos = (uint32_t)g->i; out[len].len = (uint32_t)g->j;
        }
        *chainLen = total; *score = bestScore;
    }
    free(cbeg); free(best); free(chain); free(f);
}
