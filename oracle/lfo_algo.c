/*
 * lfo_algo.c -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * libstdc++ sort/heap order restatements, dp-n2 chaining, edlib's output function, ksw_extend2.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "lf_oracle.h"
#include "lfo_internal.h"

/* =====================================================================================
 * libstdc++ (GCC 11) std::sort / std::push_heap / pop_heap / sort_heap, restated over a generic
 * element size so that equal-key order is reproduced exactly (SURVEY 7 hard part 1).
 * Follows bits/stl_algo.h:1855-1960 and bits/stl_heap.h:128-430.
 * less(a,b,ctx) is the strict weak order "a before b".
 * ===================================================================================== */
#define EL(i) (base + (size_t)(i) * es)
#define MAXES 128

static inline void el_swap(char *a, char *b, size_t es)
{
    char t[MAXES];
    memcpy(t, a, es); memcpy(a, b, es); memcpy(b, t, es);
}

static void push_heap_(char *base, size_t es, long hole, long top, const char *val, lfo_less_fn less, void *ctx)
{
    long parent = (hole - 1) / 2;
    while (hole > top && less(EL(parent), val, ctx)) {
        memcpy(EL(hole), EL(parent), es);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    memcpy(EL(hole), val, es);
}

static void adjust_heap_(char *base, size_t es, long hole, long len, const char *val, lfo_less_fn less, void *ctx)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (less(EL(child), EL(child - 1), ctx)) child--;
        memcpy(EL(hole), EL(child), es);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        memcpy(EL(hole), EL(child - 1), es);
        hole = child - 1;
    }
    push_heap_(base, es, hole, top, val, less, ctx);
}

void lfo_push_heap(void *b, size_t n, size_t es, lfo_less_fn less, void *ctx)
{
    char *base = (char *)b, v[MAXES];
    memcpy(v, EL(n - 1), es);
    push_heap_(base, es, (long)n - 1, 0, v, less, ctx);
}

/* __pop_heap(first, last, result): moves the top to `result` slot (index res), re-heapifies [0,len) */
static void pop_heap_to_(char *base, size_t es, long len, long res, lfo_less_fn less, void *ctx)
{
    char v[MAXES];
    memcpy(v, EL(res), es);
    memcpy(EL(res), EL(0), es);
    adjust_heap_(base, es, 0, len, v, less, ctx);
}

void lfo_pop_heap(void *b, size_t n, size_t es, lfo_less_fn less, void *ctx)
{
    if (n > 1) pop_heap_to_((char *)b, es, (long)n - 1, (long)n - 1, less, ctx);
}

void lfo_sort_heap(void *b, size_t n, size_t es, lfo_less_fn less, void *ctx)
{
    long last = (long)n;
    while (last > 1) { --last; pop_heap_to_((char *)b, es, last, last, less, ctx); }
}

static void make_heap_(char *base, size_t es, long len, lfo_less_fn less, void *ctx)
{
    if (len < 2) return;
    long parent = (len - 2) / 2;
    for (;;) {
        char v[MAXES];
        memcpy(v, EL(parent), es);
        adjust_heap_(base, es, parent, len, v, less, ctx);
        if (parent == 0) return;
        parent--;
    }
}

static void unguarded_linear_insert_(char *base, size_t es, long last, lfo_less_fn less, void *ctx)
{
    char v[MAXES];
    memcpy(v, EL(last), es);
    long next = last - 1;
    while (less(v, EL(next), ctx)) { memcpy(EL(last), EL(next), es); last = next; --next; }
    memcpy(EL(last), v, es);
}

static void insertion_sort_(char *base, size_t es, long first, long last, lfo_less_fn less, void *ctx)
{
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (less(EL(i), EL(first), ctx)) {
            char v[MAXES];
            memcpy(v, EL(i), es);
            memmove(EL(first + 1), EL(first), (size_t)(i - first) * es);
            memcpy(EL(first), v, es);
        } else unguarded_linear_insert_(base, es, i, less, ctx);
    }
}

static void introsort_loop_(char *base, size_t es, long first, long last, long depth, lfo_less_fn less, void *ctx)
{
    while (last - first > 16) {
        if (depth == 0) {                      /* __partial_sort(first,last,last): heap sort */
            char *b2 = EL(first);
            make_heap_(b2, es, last - first, less, ctx);
            lfo_sort_heap(b2, (size_t)(last - first), es, less, ctx);
            return;
        }
        --depth;
        /* __move_median_to_first(first, first+1, mid, last-1) */
        long a = first + 1, b = first + (last - first) / 2, c = last - 1;
        if (less(EL(a), EL(b), ctx)) {
            if (less(EL(b), EL(c), ctx)) el_swap(EL(first), EL(b), es);
            else if (less(EL(a), EL(c), ctx)) el_swap(EL(first), EL(c), es);
            else el_swap(EL(first), EL(a), es);
        } else if (less(EL(a), EL(c), ctx)) el_swap(EL(first), EL(a), es);
        else if (less(EL(b), EL(c), ctx)) el_swap(EL(first), EL(c), es);
        else el_swap(EL(first), EL(b), es);
        /* __unguarded_partition(first+1, last, pivot=first) */
        long lo = first + 1, hi = last;
        for (;;) {
            while (less(EL(lo), EL(first), ctx)) ++lo;
            --hi;
            while (less(EL(first), EL(hi), ctx)) --hi;
            if (!(lo < hi)) break;
            el_swap(EL(lo), EL(hi), es);
            ++lo;
        }
        introsort_loop_(base, es, lo, last, depth, less, ctx);
        last = lo;
    }
}

void lfo_std_sort(void *b, size_t n, size_t es, lfo_less_fn less, void *ctx)
{
    char *base = (char *)b;
    if (n == 0) return;
    long lg = 0;
    for (size_t t = n; t > 1; t >>= 1) lg++;
    introsort_loop_(base, es, 0, (long)n, 2 * lg, less, ctx);
    if (n > 16) {
        insertion_sort_(base, es, 0, 16, less, ctx);
        for (long i = 16; i < (long)n; ++i) unguarded_linear_insert_(base, es, i, less, ctx);
    } else insertion_sort_(base, es, 0, (long)n, less, ctx);
}

static int seed_qpos_less(const void *a, const void *b, void *ctx)
{
    (void)ctx;
    return ((const lfo_seed_t *)a)->qPos < ((const lfo_seed_t *)b)->qPos;   /* compare_seed, src/Chain.cpp:227-230 */
}

void lfo_sort_seeds_by_qpos(lfo_seed_t *a, size_t n) { lfo_std_sort(a, n, sizeof(lfo_seed_t), seed_qpos_less, NULL); }

/* =====================================================================================
 * chain_seeds_n2 (src/Chain.cpp:211-310)
 * ===================================================================================== */
void lfo_chain_n2(const lfo_params_t *p, lfo_seed_t *s, uint32_t n, lfo_seed_t *chain, uint32_t *chainLen, float *score)
{
    double *dp = (double *)malloc(((size_t)n + 1) * sizeof(double));
    int *prev = (int *)malloc(((size_t)n + 1) * sizeof(int));
    double best = -1;
    int bestIdx = -1;
    const double reward = p->chain_reward * (double)p->min_anchor_len;   /* score_reward, :211-215 */

    lfo_sort_seeds_by_qpos(s, n);                                        /* std::sort, :244 */
    for (int i = 0; i < (int)n; i++) {
        dp[i] = s[i].len;
        prev[i] = -1;
        for (int j = i - 1; j >= 0; j--) {
            int distR = (int)s[i].qPos - ((int)s[j].qPos + (int)s[j].len - 1);
            if (distR <= 0) continue;
            int distT = (int)(s[i].tPos - (s[j].tPos + s[j].len - 1));   /* u32 arithmetic narrowed to int */
            if (distT <= 0) continue;
            int dist = distR < distT ? distT - distR : distR - distT;    /* score_penalty, :217-225 */
            double pen = dist <= 1 ? 0 : 0.1 * dist + p->chain_penalty * log(dist);
            double cand = dp[j] + reward - pen;
            if (cand > dp[i]) { dp[i] = cand; prev[i] = j; }
        }
        if (dp[i] > best) { best = dp[i]; bestIdx = i; }
    }
    uint32_t len = 0;
    for (int k = bestIdx; k != -1; k = prev[k]) len++;
    uint32_t w = len;
    for (int k = bestIdx; k != -1; k = prev[k]) chain[--w] = s[k];
    *chainLen = len;
    *score = (float)best;
    free(dp); free(prev);
}

/* =====================================================================================
 * edlibAlign(q, n, t, m, {k=-1, mode, EDLIB_TASK_PATH}) as the pure function of SURVEY App. F.
 * The reference computes it with Myers bit-vectors in an Ukkonen band with k-doubling
 * (lib/edlib/edlib.cpp:101-221,475-858); band and doubling never change the result, so this checker
 * deliberately uses a plain integer DP -- an independent formulation from the bit-parallel GPU kernel.
 * ===================================================================================== */

/* last row distances: out[r] = dist(q[0..r), t[0..m)) for r = 0..n ; rev = walk both strings backwards */
static void nw_last_col(const char *q, int n, const char *t, int m, int rev, int *out)
{
    for (int r = 0; r <= n; r++) out[r] = r;
    for (int c = 1; c <= m; c++) {
        char tc = rev ? t[m - c] : t[c - 1];
        int diag = out[0];
        out[0] = c;
        for (int r = 1; r <= n; r++) {
            char qc = rev ? q[n - r] : q[r - 1];
            int up = out[r - 1], left = out[r];
            int v = diag + (qc != tc);
            if (up + 1 < v) v = up + 1;
            if (left + 1 < v) v = left + 1;
            diag = left;
            out[r] = v;
        }
    }
}

/* Leaf: full matrix + traceback with edlib's move priority Up(1) -> Left(2) -> Diagonal
 * (lib/edlib/edlib.cpp:905-1066). Appends ops (forward order) at ops+*nops. */
static void path_leaf(const char *q, int n, const char *t, int m, uint8_t *ops, int *nops)
{
    size_t W = (size_t)m + 1;
    int *D = (int *)malloc(((size_t)n + 1) * W * sizeof(int));
    for (int c = 0; c <= m; c++) D[c] = c;
    for (int r = 1; r <= n; r++) {
        int *row = D + (size_t)r * W, *pr = row - W;
        row[0] = r;
        char qc = q[r - 1];
        for (int c = 1; c <= m; c++) {
            int v = pr[c - 1] + (qc != t[c - 1]);
            if (pr[c] + 1 < v) v = pr[c] + 1;
            if (row[c - 1] + 1 < v) v = row[c - 1] + 1;
            row[c] = v;
        }
    }
    uint8_t *rev = (uint8_t *)malloc((size_t)n + (size_t)m + 1);
    int k = 0, r = n, c = m;
    for (;;) {
        int cur = D[(size_t)r * W + c];
        if (D[(size_t)(r - 1) * W + c] + 1 == cur) {                 /* up: consume query */
            rev[k++] = 1; r--;
            if (r == 0) { for (int i = 0; i < c; i++) rev[k++] = 2; break; }
        } else if (D[(size_t)r * W + c - 1] + 1 == cur) {            /* left: consume target */
            rev[k++] = 2; c--;
            if (c == 0) { for (int i = 0; i < r; i++) rev[k++] = 1; break; }
        } else {
            rev[k++] = (D[(size_t)(r - 1) * W + c - 1] == cur) ? 0 : 3;
            r--; c--;
            if (c == 0) { for (int i = 0; i < r; i++) rev[k++] = 1; break; }
            if (r == 0) { for (int i = 0; i < c; i++) rev[k++] = 2; break; }
        }
    }
    for (int i = 0; i < k; i++) ops[*nops + i] = rev[k - 1 - i];
    *nops += k;
    free(rev); free(D);
}

/* obtainAlignment (lib/edlib/edlib.cpp:1090-1143) + obtainAlignmentHirschberg (:1161-1330) */
static void path_rec(const char *q, int n, const char *t, int m, int best, uint8_t *ops, int *nops)
{
    if (n == 0) { for (int i = 0; i < m; i++) ops[(*nops)++] = 2; return; }
    if (m == 0) { for (int i = 0; i < n; i++) ops[(*nops)++] = 1; return; }
    long long nb = (n + 63) / 64;
    if (20LL * nb * m + 8LL * m < 1024 * 1024) { path_leaf(q, n, t, m, ops, nops); return; }

    int lw = m / 2, rw = m - lw;
    int *Fc = (int *)malloc(((size_t)n + 1) * sizeof(int));
    int *Rc = (int *)malloc(((size_t)n + 1) * sizeof(int));
    nw_last_col(q, n, t, lw, 0, Fc);            /* Fc[r] = dist(q[0..r), t[0..lw))      */
    nw_last_col(q, n, t + lw, rw, 1, Rc);       /* Rc[x] = dist(last x of q, t[lw..m))  */
    /* edlib's row index `queryIdx` (0-based, inclusive end of the upper part) -> our r = queryIdx+1 rows.
     * Order of candidates (:1263-1289): queryIdx = 0..n-2 ascending, then -1, then n-1. */
    int split = -2, ls = 0, rs = 0;
    for (int qi = 0; qi <= n - 2; qi++) {
        if (Fc[qi + 1] + Rc[n - qi - 1] == best) { split = qi; ls = Fc[qi + 1]; rs = Rc[n - qi - 1]; break; }
    }
    if (split == -2 && lw + Rc[n] == best) { split = -1; ls = lw; rs = Rc[n]; }
    if (split == -2 && Fc[n] + rw == best) { split = n - 1; ls = Fc[n]; rs = rw; }
    free(Fc); free(Rc);
    if (split == -2) { fprintf(stderr, "[lfo] hirschberg: no split found\n"); abort(); }
    int ul = split + 1;
    path_rec(q, ul, t, lw, ls, ops, nops);
    path_rec(q + ul, n - ul, t + lw, rw, rs, ops, nops);
}

int lfo_edlib(const char *q, int n, const char *t, int m, int mode, int *endLoc, uint8_t *ops, int *nops)
{
    int *col = (int *)malloc(((size_t)(n > m ? n : m) + 2) * sizeof(int));
    int ed, tl = m;
    *nops = 0;
    if (mode == 0) {
        nw_last_col(q, n, t, m, 0, col);
        ed = col[n];
        *endLoc = m - 1;
    } else {
        /* SHW: min over target prefixes c in [0,m] of D[n][c], smallest c on ties, c = 0 allowed
         * (lib/edlib/edlib.cpp:583-618). Row-wise DP keeping D[n][*]. */
        int *row = (int *)malloc(((size_t)m + 1) * sizeof(int));
        for (int c = 0; c <= m; c++) row[c] = c;
        for (int r = 1; r <= n; r++) {
            int diag = row[0];
            row[0] = r;
            char qc = q[r - 1];
            for (int c = 1; c <= m; c++) {
                int v = diag + (qc != t[c - 1]);
                if (row[c] + 1 < v) v = row[c] + 1;
                if (row[c - 1] + 1 < v) v = row[c - 1] + 1;
                diag = row[c];
                row[c] = v;
            }
        }
        /* the empty prefix (end location -1) is only reachable through edlib's wildcard padding of the
         * last 64-row block (lib/edlib/edlib.cpp:595: position = c - W), i.e. when n % 64 != 0 */
        int bc = (n % 64 != 0 || m == 0) ? 0 : 1;
        for (int c = bc + 1; c <= m; c++) if (row[c] < row[bc]) bc = c;
        ed = row[bc];
        tl = bc;
        *endLoc = bc - 1;
        free(row);
    }
    free(col);
    path_rec(q, n, t, tl, ed, ops, nops);
    return ed;
}

/* =====================================================================================
 * ksw_extend2 (lib/bwa/ksw.c:380-478) specialised to lordFAST's 5x5 clip matrix
 * (match +2, mismatch -16, any N 0; src/LordFAST.cpp:82-85,178-187), end_bonus 0.
 * H/E kept in two arrays; same cell order, band, z-drop and row-trimming rules.
 * ===================================================================================== */
static inline int clip_score(int a, int b) { return (a > 3 || b > 3) ? 0 : (a == b ? 2 : -16); }

int lfo_ksw_extend2(int qlen, const uint8_t *q, int tlen, const uint8_t *t, int o_del, int e_del,
                    int o_ins, int e_ins, int w, int zdrop, int h0, int *qle, int *tle)
{
    int32_t *H = (int32_t *)calloc((size_t)qlen + 2, sizeof(int32_t));
    int32_t *E = (int32_t *)calloc((size_t)qlen + 2, sizeof(int32_t));
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    /* first row (:394-397) */
    H[0] = h0;
    H[1] = h0 > oe_ins ? h0 - oe_ins : 0;
    for (int j = 2; j <= qlen && H[j - 1] > e_ins; ++j) H[j] = H[j - 1] - e_ins;
    /* band clamp (:398-407): max matrix entry is 2, end_bonus 0 */
    int max_ins = (int)((double)(qlen * 2 - o_ins) / e_ins + 1.);
    if (max_ins < 1) max_ins = 1;
    if (w > max_ins) w = max_ins;
    int max_del = (int)((double)(qlen * 2 - o_del) / e_del + 1.);
    if (max_del < 1) max_del = 1;
    if (w > max_del) w = max_del;

    int max = h0, max_i = -1, max_j = -1, beg = 0, end = qlen;
    for (int i = 0; i < tlen; ++i) {
        int f = 0, h1, m = 0, mj = -1, j;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        for (j = beg; j < end; ++j) {
            int M = H[j], e = E[j], h, tt;
            H[j] = h1;
            M = M ? M + clip_score(t[i], q[j]) : 0;
            h = M > e ? M : e;
            h = h > f ? h : f;
            h1 = h;
            mj = m > h ? mj : j;
            m = m > h ? m : h;
            tt = M - oe_del; if (tt < 0) tt = 0;
            e -= e_del; if (e < tt) e = tt;
            E[j] = e;
            tt = M - oe_ins; if (tt < 0) tt = 0;
            f -= e_ins; if (f < tt) f = tt;
        }
        H[end] = h1; E[end] = 0;
        if (m == 0) break;
        if (m > max) { max = m; max_i = i; max_j = mj; }
        else if (zdrop > 0) {
            if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        for (j = beg; j < end && H[j] == 0 && E[j] == 0; ++j) {}
        beg = j;
        for (j = end; j >= beg && H[j] == 0 && E[j] == 0; --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    free(H); free(E);
    if (qle) *qle = max_j + 1;
    if (tle) *tle = max_i + 1;
    return max;
}
