/* lf_clasp_kernel.h -- clasp (sum-of-pair gap cost) chaining on the device, shared by lf_chain.hip (host-supplied
 * windows) and lf_vote.hip (windows built on the device).
 *
 * Replaces chain_seeds_clasp (src/Chain.cpp:39-209) and the clasp calls it makes with chainmode SOP, lambda 0.15,
 * eps 0, maxgap -1: bl_slClusterSop / bl_slChainSop / bl_slChainSopRMQ / bl_slChainSopActivate
 * (lib/clasp/slchain.c:568-974), bl_slExtractPoints / bl_slGetTrans (:49-90, :369-404), quickSort
 * (lib/clasp/sort.c:164-225) and the shape of the 2-d range tree (lib/clasp/rangetree.c:204-330).
 *
 * One wavefront per window; fragments arrive stably sorted by target start (what the reference's qsort gives).
 * The formulation is NOT the reference's pointer structure:
 *   - the reference keeps, in every range-tree node, an ordered map  y-rank -> chain  as a priority staircase
 *     (insert if prio >= predecessor's, then delete lower successors).  At any time that map is the Pareto
 *     front of the chains activated so far in the node, so what a query reads from a node -- pred(y) -- is
 *         argmax over activated entries of the node with y-rank < y of (prio, y-rank)   (lexicographic).
 *     That is order free: entries live in flat arrays indexed by the first-dimension rank, a node is a
 *     contiguous rank range, and a query is one strided scan + wavefront arg-max per canonical node.
 *   - the canonical nodes of a query are enumerated arithmetically (left part = ceil(n/2)), root to leaf, which is
 *     also the order the reference applies its side effect in (a visited predecessor chain may improve the best
 *     chain remembered at its first fragment, slchain.c:877-895) -- lane 0 replays those in that order.
 *   - chains are (score, first fragment, previous fragment) triples; the best chain remembered at a fragment is
 *     (score, base chain, optional extra fragment).
 * What has to be replayed literally is clasp's own quickSort: it is unstable, and the tie order of equal
 * diagonals / equal positions decides the ranks.  Lane 0 sorts the fragment ends, lanes 0-3 the four point orders.
 * All arithmetic is integer or FP64 in the reference's evaluation order (-ffp-contract=off).
 *
 * Working set: 136 B per fragment in LDS (16-bit ranks, sort keys aliased with the rank / entry arrays they are dead
 * before), 200 B in the HBM workspace (32-bit ranks) for windows above LF_CLASP_LDS_MAX fragments.  The kernel is a
 * template on that choice so that the LDS instantiation addresses LDS directly (ds_read / ds_write, no FLAT).
 * A tree walk runs in three phases (lf_clasp_rmq): scans + DPP arg-max per node, one lane per node winner for the gap
 * costs, lane 0 for the ordered side effects.
 */
#ifndef LF_CLASP_KERNEL_H
#define LF_CLASP_KERNEL_H
#include "lf_gpu_common.h"
#include "lf_chain_kernel.h"
#include <float.h>
#include <type_traits>

/* one wavefront per workgroup: a hand-over between its lanes needs the outstanding LDS / memory operations to be complete (the
 * fence's wait counters) and no motion of accesses across it -- not an s_barrier */
#define LF_CLASP_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#define LF_CLASP_BYTES_PER_FRAG 200u       /* HBM working set per fragment (32-bit ranks), see lf_clasp_carve */
#define LF_CLASP_LDS_BYTES_PER_FRAG 136u   /* LDS working set per fragment (16-bit ranks, keys aliased) */
#define LF_CLASP_LDS_MAX 1024u             /* windows up to this many fragments work in LDS (136 KiB) */
#define LF_CLASP_NONE 0xFFFFFFFFu

/* R = type of a rank / point index: uint16_t in the LDS instantiation (at most 2 * LF_CLASP_LDS_MAX points), uint32_t in
 * the HBM-workspace one */
template <class R>
struct lf_clasp_mem {
    double *chain_scr, *best_scr, *prioA, *prioB;
    int *fp, *fq, *fl, *chain_first, *chain_prev, *best_base, *best_extra, *prev;
    int *px, *py, *pidx;
    R *tr;                        /* tr[k * N + point] = rank of the point in order k */
    R *entyA, *entfA, *entyB, *entfB;
    int *keys;                    /* keys[k * N + point]; 16-bit ranks: aliases tr + ent* (dead once the ranks exist) */
    R *sorted;                    /* sorted[k * N + rank] = point; aliases prioA/prioB (dead before the sweep) */
    uint32_t *ring; int ring_cap; /* pending ranges of the breadth-first sort; aliases chain_scr / best_scr (dead while a cluster is sorted) */
};

template <class R>
__device__ __forceinline__ void lf_clasp_carve(lf_clasp_mem<R> &m, unsigned char *base, size_t cap)
{
    const size_t N = 2 * cap;
    m.chain_scr = reinterpret_cast<double *>(base);
    m.best_scr = m.chain_scr + cap;
    m.prioA = m.best_scr + cap;
    m.prioB = m.prioA + N;
    m.fp = reinterpret_cast<int *>(m.prioB + N);
    m.fq = m.fp + cap; m.fl = m.fq + cap; m.chain_first = m.fl + cap; m.chain_prev = m.chain_first + cap;
    m.best_base = m.chain_prev + cap; m.best_extra = m.best_base + cap; m.prev = m.best_extra + cap;
    m.px = m.prev + cap; m.py = m.px + N; m.pidx = m.py + N;
    m.keys = m.pidx + N;                                              /* 4N ints */
    if (sizeof(R) == 2) {                                             /* 136 B per fragment */
        m.tr = reinterpret_cast<R *>(m.keys);                         /* 4N + 4N 16-bit entries == the 4N ints of keys */
        m.entyA = m.tr + 4 * N;
    } else {                                                          /* 200 B per fragment */
        m.tr = reinterpret_cast<R *>(m.keys + 4 * N);
        m.entyA = m.tr + 4 * N;
    }
    m.entfA = m.entyA + N; m.entyB = m.entfA + N; m.entfB = m.entyB + N;
    m.sorted = reinterpret_cast<R *>(m.prioA);
    m.ring = reinterpret_cast<uint32_t *>(m.chain_scr); m.ring_cap = (int)(4 * cap);      /* 2 * cap doubles = 4 * cap entries >= 4 arrays x N / 2 ranges */
}

/* D(a,b) of lib/clasp/slchain.h:40 */
__device__ __forceinline__ int lf_clasp_d(int a, int b) { return (a > b) ? (a - b - 1) : (b > a) ? (b - a - 1) : 1; }
/* GSOP(fprim, f) (lib/clasp/slchain.h:45-47), lambda = 0.15, eps = 0 */
__device__ __forceinline__ double lf_clasp_gsop(int cs, int cq, int fend_s, int fend_q)
{
    const int dx = lf_clasp_d(cs, fend_s), dy = lf_clasp_d(cq, fend_q);
    return (dx >= dy) ? (0.15 * (double)dx + (0.0 - 0.15) * (double)dy) : (0.15 * (double)dy + (0.0 - 0.15) * (double)dx);
}

/* clasp's quickSort (lib/clasp/sort.c:164-225) on an index array; comparator = the int key (equal keys compare 0) */
template <class R>
__device__ __forceinline__ void lf_clasp_qsort(R *sorted, const int *keys, int size)
{
    int stk[96]; int top = 0;
    stk[0] = 0; stk[1] = size - 1; top = 1;
    while (top > 0) {
        top--;
        int left = stk[2 * top], right = stk[2 * top + 1];
        while (left < right) {
            const int xk = keys[sorted[(left + right) >> 1]];
            int l2 = left, r2 = right;
            do {
                while (keys[sorted[l2]] < xk) l2++;
                while (keys[sorted[r2]] > xk) r2--;
                if (l2 <= r2) { const R t = sorted[r2]; sorted[r2] = sorted[l2]; sorted[l2] = t; l2++; r2--; }
            } while (r2 >= l2);
            if (top < 47) {
                if ((l2 - left) > (right - l2)) { stk[2 * top] = left; stk[2 * top + 1] = r2; top++; left = l2; }
                else { stk[2 * top] = l2; stk[2 * top + 1] = right; top++; right = r2; }
            } else {   /* cannot happen below 2^40 elements (the continued part at most halves) */
                __builtin_trap();
            }
        }
    }
}

/* The same sort, replayed BREADTH-FIRST by the whole wavefront (16-bit instantiation: at most 2^15 elements per array).
 * quickSort's result does not depend on the order in which its pending sub-ranges are processed -- they are disjoint, and a
 * partition step only looks at its own range -- so instead of one lane walking the recursion (N log N dependent LDS round
 * trips) every pending range gets a lane of its own: a FIFO ring of (array, left, right) entries, 64 ranges partitioned per
 * round with the reference's own Hoare loop, children appended behind.  The longest range of a round bounds its time:
 * ~2 N steps for the whole sort.  `n_arr` arrays of `size` elements, array a at sorted + a * stride / keys + a * stride, are
 * sorted together.  ring: 2 * n_arr * size / 2 entries at most are ever pending (ranges of >= 2 elements are disjoint). */
template <class R>
__device__ __forceinline__ void lf_clasp_qsort_bf(R *sorted, const int *keys, int size, int n_arr, int stride, uint32_t *ring, int ring_cap, int lane)
{
    if (size < 2) return;
    if (lane < n_arr) ring[lane] = ((uint32_t)lane << 30) | (uint32_t)(size - 1);       /* (array, left = 0, right = size - 1) */
    int head = 0, cnt = n_arr;                                     /* wave-uniform */
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    while (cnt > 0) {
        LF_CLASP_SYNC();
        const int take = cnt < 64 ? cnt : 64;
        int c0 = 0, c1 = 0; uint32_t e0 = 0, e1 = 0;
        if (lane < take) {
            int idx = head + lane; if (idx >= ring_cap) idx -= ring_cap;
            const uint32_t e = ring[idx];
            const int a = (int)(e >> 30), left = (int)((e >> 15) & 0x7fffu), right = (int)(e & 0x7fffu);
            R *sd = sorted + a * stride; const int *ky = keys + a * stride;
            const int xk = ky[sd[(left + right) >> 1]];
            int l2 = left, r2 = right;
            do {
                while (ky[sd[l2]] < xk) l2++;
                while (ky[sd[r2]] > xk) r2--;
                if (l2 <= r2) { const R t = sd[r2]; sd[r2] = sd[l2]; sd[l2] = t; l2++; r2--; }
            } while (r2 >= l2);
            if (left < r2) { c0 = 1; e0 = ((uint32_t)a << 30) | ((uint32_t)left << 15) | (uint32_t)r2; }
            if (l2 < right) { c1 = 1; e1 = ((uint32_t)a << 30) | ((uint32_t)l2 << 15) | (uint32_t)right; }
        }
        const uint64_t b0 = lf_ballot(c0 != 0), b1 = lf_ballot(c1 != 0);
        const int n0 = __popcll(b0), n1 = __popcll(b1);
        head += take; if (head >= ring_cap) head -= ring_cap;
        cnt -= take;
        int tail = head + cnt; if (tail >= ring_cap) tail -= ring_cap;
        if (c0) { int w = tail + __popcll(b0 & below); if (w >= ring_cap) w -= ring_cap; ring[w] = e0; }
        if (c1) { int w = tail + n0 + __popcll(b1 & below); if (w >= ring_cap) w -= ring_cap; if (w >= ring_cap) w -= ring_cap; ring[w] = e1; }
        cnt += n0 + n1;
    }
    LF_CLASP_SYNC();
}

/* one candidate of a range-tree node: the entry with the greatest (prio, y-rank) below the query's y-rank */
struct lf_clasp_cand { double pr; uint32_t ey, ix; };
struct lf_clasp_better {
    __device__ __forceinline__ lf_clasp_cand operator()(const lf_clasp_cand &a, const lf_clasp_cand &b) const
    {
        const bool take_b = b.ey != LF_CLASP_NONE && (a.ey == LF_CLASP_NONE || b.pr > a.pr || (b.pr == a.pr && b.ey > a.ey));
        return take_b ? b : a;
    }
};
/* wavefront arg-max of (prio, y-rank), lexicographic, over the lanes that hold a candidate; the winner's prio and entry index
 * come back in every lane.  Three 32-bit maxima (high word of the order-preserving image of the double, low word among the
 * lanes that tie on it, y-rank among those) instead of a generic reduction of a 16-byte struct: y-ranks are distinct, so one
 * lane is left. */
__device__ __forceinline__ bool lf_clasp_wave_best(bool cand, double pr, uint32_t ey, uint32_t ix, double &wpr, uint32_t &wix)
{
    if (lf_ballot(cand) == 0) return false;
    uint64_t k = (uint64_t)__double_as_longlong(pr + 0.0);
    k = (k >> 63) ? ~k : (k | 0x8000000000000000ull);               /* a < b  <=>  image(a) < image(b) */
    /* (every maximum is taken by ALL lanes, outside the && : a DPP move that only some lanes execute reads stale registers of the others) */
    const uint32_t hi = cand ? (uint32_t)(k >> 32) : 0u;
    const uint32_t hmax = lf_wave_max_u32(hi);
    const bool c2 = cand && hi == hmax;
    const uint32_t lo = c2 ? (uint32_t)k : 0u;
    const uint32_t lmax = lf_wave_max_u32(lo);
    const bool c3 = c2 && lo == lmax;
    const uint32_t e = c3 ? ey + 1u : 0u;
    const uint32_t emax = lf_wave_max_u32(e);
    const bool c4 = c3 && e == emax;
    const int src = __ffsll((long long)lf_ballot(c4)) - 1;
    const uint64_t pb = (uint64_t)__double_as_longlong(pr);
    const uint32_t ph = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pb >> 32), src), pl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pb, src);
    wpr = __longlong_as_double((long long)(((uint64_t)ph << 32) | pl));
    wix = (uint32_t)__builtin_amdgcn_readlane((int)ix, src);
    return true;
}

/* per-query scratch of one tree walk: what lane 0 needs of the canonical nodes' winners, root to leaf (at most ~log2 N + 1) */
#define LF_CLASP_MAX_NODES 40
struct lf_clasp_nodes {
    double pr[LF_CLASP_MAX_NODES], cmp[LF_CLASP_MAX_NODES], nw[LF_CLASP_MAX_NODES];
    int tf[LF_CLASP_MAX_NODES], first[LF_CLASP_MAX_NODES];
};

/* bl_slChainSopRMQ (slchain.c:841-912) for the start point of fragment `cur`; returns the chosen chain or -1 (lane 0).
 * Three phases: (A) every canonical node, root to leaf: strided scan + wavefront arg-max, node K's winner parked in lane K's
 * registers; (B) lane k evaluates node k's winner -- gap cost, the test of :877 and both score expressions -- all nodes at
 * once; (C) lane 0 applies the side effects and picks the result in node order, which is all that has to be sequential.
 * One wavefront works on a window and LDS operations of a wavefront complete in order: a wave-level fence orders the
 * hand-over (B) -> (C), no workgroup barrier. */
template <class R>
__device__ __forceinline__ int lf_clasp_rmq(const lf_clasp_mem<R> &m, const R *enty, const R *entf, const double *prio,
                                   int N, uint32_t x, uint32_t y, int cur, int lane, lf_clasp_nodes *nd)
{
    const int cs = m.fp[cur], cq = m.fq[cur]; const double cscr = (double)m.fl[cur];
    int K = 0;
    int s = 0, cnt = N;
    double my_pr = 0; uint32_t my_ix = 0;
    while (cnt > 0) {
        int lo, len;
        if (cnt == 1) {                         /* leaf: its own map, then right == NULL / left == NULL */
            if ((uint32_t)s > x) break;
            lo = s; len = 1; cnt = 0;
        } else {
            const int mid = (cnt >> 1) + (cnt & 1);
            if ((uint32_t)(s + mid - 1) > x) { cnt = mid; continue; }      /* go left */
            lo = s; len = mid; s += mid; cnt -= mid;                          /* left child's map, go right */
        }
        double pr = 0; uint32_t ey = LF_CLASP_NONE, ix = 0;
        for (int k = lo + lane; k < lo + len; k += 64) {
            const uint32_t e = enty[k];            /* an unset entry is (R)~0: never below a rank */
            const double p2 = prio[k];             /* (both loads are issued together) */
            if (e < y && (ey == LF_CLASP_NONE || p2 > pr || (p2 == pr && e > ey))) { pr = p2; ey = e; ix = (uint32_t)k; }
        }
        double wpr; uint32_t wix;
        if (!lf_clasp_wave_best(ey != LF_CLASP_NONE, pr, ey, ix, wpr, wix)) continue;
        if (lane == K) { my_pr = wpr; my_ix = wix; }
        K++;
    }
    if (K == 0) return -1;
    if (lane < K) {
        const int tf = (int)entf[my_ix];
        const double g = lf_clasp_gsop(cs, cq, m.fp[tf] + m.fl[tf] - 1, m.fq[tf] + m.fl[tf] - 1);
        const double cscr_tf = m.chain_scr[tf];
        nd->pr[lane] = my_pr;
        nd->tf[lane] = tf;
        nd->first[lane] = (cscr >= g) ? m.chain_first[tf] : -1;                        /* :877 */
        nd->cmp[lane] = cscr + cscr_tf - g;                                            /* :884 */
        nd->nw[lane] = cscr_tf + (cscr - g);                                           /* :890 */
    }
    LF_CLASP_SYNC();
    int res = -1;
    if (lane == 0) {
        double resprio = -DBL_MAX;
        for (int k = 0; k < K; k++) {
            const int tf = nd->tf[k], first = nd->first[k];
            if (first >= 0 && m.best_scr[first] < nd->cmp[k]) { m.best_scr[first] = nd->nw[k]; m.best_base[first] = tf; m.best_extra[first] = cur; }
            if (nd->pr[k] > resprio) { res = tf; resprio = nd->pr[k]; }                /* :898 */
        }
    }
    LF_CLASP_SYNC();
    return res;
}

/* bl_slChainSop (slchain.c:668-826) over the cluster of fragments [cb, cb + cm) */
template <class R>
__device__ __forceinline__ void lf_clasp_chain_sop(const lf_clasp_mem<R> &m, int cb, int cm, int lane, lf_clasp_nodes *wr_tmp)
{
    const int N = 2 * cm;
    const int xmin = m.fp[cb];
    /* ---- bl_slExtractPoints: fragment ends sorted with clasp's quickSort, merged with the starts ---- */
    for (int i = lane; i < cm; i += 64) { m.sorted[i] = (R)i; m.keys[i] = m.fp[cb + i] + m.fl[cb + i] - 1 - xmin; m.best_base[cb + i] = -1; m.prev[cb + i] = -1; }
    LF_CLASP_SYNC();
    if (sizeof(R) == 2) {
        lf_clasp_qsort_bf(m.sorted, m.keys, cm, 1, 0, m.ring, m.ring_cap, lane);
        /* the merge of slchain.c:49-90 emits, in front of start a, every end (in sorted order) that lies before it: the place of
         * a point is its rank in its own list + the number of points of the other list in front of it -- two binary searches
         * per fragment instead of one lane walking both lists (starts are in target order: the fragments arrive sorted) */
        for (int i = lane; i < cm; i += 64) {
            const int a = cb + i, sa = m.fp[a] - xmin;
            int lo = 0, hi = cm;                                   /* ends with end < start_a */
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (m.keys[m.sorted[mid]] < sa) lo = mid + 1; else hi = mid; }
            const int np = i + lo;
            m.px[np] = m.fp[a]; m.py[np] = m.fq[a]; m.pidx[np] = (a << 1) | 1;
            const int b = cb + (int)m.sorted[i], eb = m.keys[m.sorted[i]] + xmin;
            lo = 0; hi = cm;                                       /* starts with start <= end_b */
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (m.fp[cb + mid] <= eb) lo = mid + 1; else hi = mid; }
            const int nq = i + lo;
            m.px[nq] = eb; m.py[nq] = m.fq[b] + m.fl[b] - 1; m.pidx[nq] = b << 1;
        }
    } else if (lane == 0) {
        lf_clasp_qsort(m.sorted, m.keys, cm);
        int np = 0, j = 0;
        for (int i = 0; i < cm; i++) {
            const int a = cb + i; int b = cb + (int)m.sorted[j];
            while (m.fp[a] > m.fp[b] + m.fl[b] - 1) {
                m.px[np] = m.fp[b] + m.fl[b] - 1; m.py[np] = m.fq[b] + m.fl[b] - 1; m.pidx[np] = b << 1; np++;
                j++; b = cb + (int)m.sorted[j];
            }
            m.px[np] = m.fp[a]; m.py[np] = m.fq[a]; m.pidx[np] = (a << 1) | 1; np++;
        }
        while (j < cm) {
            const int b = cb + (int)m.sorted[j];
            m.px[np] = m.fp[b] + m.fl[b] - 1; m.py[np] = m.fq[b] + m.fl[b] - 1; m.pidx[np] = b << 1; np++; j++;
        }
    }
    LF_CLASP_SYNC();
    /* ---- bl_slGetTrans: four orders of the points (comparators slchain.c:452-559 folded into int keys) ---- */
    for (int i = lane; i < N; i += 64) {
        const int x = m.px[i] - xmin, y = m.py[i], st = m.pidx[i] & 1;
        m.keys[i] = (x - y) * 2 + st;               /* T1.x = x - y, end points first */
        m.keys[N + i] = y * 2 + (1 - st);           /* T1.y = y, start points first */
        m.keys[2 * N + i] = x * 2 + (1 - st);       /* T2.x = x, start points first */
        m.keys[3 * N + i] = (y - x) * 2 + st;       /* T2.y = y - x, end points first */
        m.sorted[i] = m.sorted[N + i] = m.sorted[2 * N + i] = m.sorted[3 * N + i] = (R)i;
    }
    LF_CLASP_SYNC();
    if (sizeof(R) == 2) lf_clasp_qsort_bf(m.sorted, m.keys, N, 4, N, m.ring, m.ring_cap, lane);
    else if (lane < 4) lf_clasp_qsort(m.sorted + lane * N, m.keys + lane * N, N);
    LF_CLASP_SYNC();
    for (int r = lane; r < N; r += 64) {
        uint32_t rk[4];
#pragma unroll
        for (int k = 0; k < 4; k++) rk[k] = m.sorted[k * N + r];
#pragma unroll
        for (int k = 0; k < 4; k++) m.tr[k * N + rk[k]] = (R)r;
    }
    const int ta = m.px[m.sorted[2 * N + N - 1]], tb = m.py[m.sorted[N + N - 1]];     /* t.a, t.b :703-706 */
    LF_CLASP_SYNC();
    for (int i = lane; i < N; i += 64) { m.entyA[i] = (R)~(R)0; m.entyB[i] = (R)~(R)0; }
    LF_CLASP_SYNC();
    /* ---- the sweep over the points ---- */
    for (int t = 0; t < N; t++) {
        const int pi = m.pidx[t], cur = pi >> 1;
        const uint32_t t0 = m.tr[t], t1 = m.tr[N + t], t2 = m.tr[2 * N + t], t3 = m.tr[3 * N + t];
        if (pi & 1) {
            const int ap = lf_clasp_rmq(m, m.entyA, m.entfA, m.prioA, N, t0, t1, cur, lane, wr_tmp);
            const int bp = lf_clasp_rmq(m, m.entyB, m.entfB, m.prioB, N, t2, t3, cur, lane, wr_tmp);
            if (lane == 0) {
                const int cs = m.fp[cur], cq = m.fq[cur];
                int pv;
                if (ap < 0) pv = bp;
                else if (bp < 0) pv = ap;
                else {
                    const double ga = lf_clasp_gsop(cs, cq, m.fp[ap] + m.fl[ap] - 1, m.fq[ap] + m.fl[ap] - 1);
                    const double gb = lf_clasp_gsop(cs, cq, m.fp[bp] + m.fl[bp] - 1, m.fq[bp] + m.fl[bp] - 1);
                    pv = (m.chain_scr[ap] - ga >= m.chain_scr[bp] - gb) ? ap : bp;                  /* :727-733 */
                }
                if (pv >= 0 && m.chain_scr[pv] < lf_clasp_gsop(cs, cq, m.fp[pv] + m.fl[pv] - 1, m.fq[pv] + m.fl[pv] - 1)) pv = -1;   /* :739 */
                m.prev[cur] = pv;
            }
        } else if (lane == 0) {
            const int cs = m.fp[cur], cq = m.fq[cur]; const double cscr = (double)m.fl[cur];
            const int cand = m.prev[cur];
            double scr;
            if (cand >= 0) {
                scr = cscr + m.chain_scr[cand] - lf_clasp_gsop(cs, cq, m.fp[cand] + m.fl[cand] - 1, m.fq[cand] + m.fl[cand] - 1);   /* :759 */
                const int first = m.chain_first[cand];
                m.chain_first[cur] = first; m.chain_prev[cur] = cand;
                if (m.best_base[first] >= 0 && m.best_scr[first] <= scr) { m.best_scr[first] = scr; m.best_base[first] = cur; m.best_extra[first] = -1; }   /* :772-779 */
            } else {
                scr = cscr; m.chain_first[cur] = cur; m.chain_prev[cur] = -1;
                m.best_scr[cur] = scr; m.best_base[cur] = cur; m.best_extra[cur] = -1;                  /* :783-796 */
            }
            m.chain_scr[cur] = scr;
            const int fe_s = cs + m.fl[cur] - 1, fe_q = cq + m.fl[cur] - 1;
            const int ds = lf_clasp_d(ta, fe_s), dq = lf_clasp_d(tb, fe_q);
            const double g1 = 0.15 * (double)ds + (0.0 - 0.15) * (double)dq;                           /* GCSOP1 */
            const double g2 = 0.15 * (double)dq + (0.0 - 0.15) * (double)ds;                           /* GCSOP2 */
            m.prioA[t0] = scr - g1; m.entfA[t0] = (R)cur; m.entyA[t0] = (R)t1;
            m.prioB[t2] = scr - g2; m.entfB[t2] = (R)cur; m.entyB[t2] = (R)t3;
        }
        LF_CLASP_SYNC();
    }
}

/* LDS = true: the window's working set is carved from dynamic LDS (windows up to `cap` fragments); every pointer is
 * then derived from the LDS base alone, so the compiler emits ds_read / ds_write instead of FLAT accesses.
 * LDS = false: windows above LF_CLASP_LDS_MAX work in their slice of the HBM workspace. */
template <bool LDS>
static __global__ void __launch_bounds__(64)
lf_clasp_kernel(const lf_chain_win *__restrict__ wins, int n_wins, const uint32_t *__restrict__ seeds /* (tPos, qPos:20|len:12), by target start */,
                const uint32_t *__restrict__ shift /* per window id: value subtracted from tPos (0 or 2000000000), may be null */,
                uint32_t cap, unsigned char *__restrict__ ws,
                uint32_t *__restrict__ chain_idx, uint32_t *__restrict__ chain_len, float *__restrict__ score,
                uint32_t n_min, uint32_t n_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ lf_clasp_nodes wr_tmp;
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= n_wins) return;
    const lf_chain_win w = wins[blockIdx.x];
    if (w.n < n_min || w.n > n_max) return;
    const int n = (int)w.n;
    if (n == 0) { if (lane == 0) { chain_len[w.id] = 0; score[w.id] = -1.0f; } return; }
    typedef typename std::conditional<LDS, uint16_t, uint32_t>::type R;
    lf_clasp_mem<R> m;
    if (LDS) lf_clasp_carve(m, smem, cap);
    else lf_clasp_carve(m, ws + w.ws_off * (uint64_t)LF_CLASP_BYTES_PER_FRAG, w.n);
    const uint32_t sh = shift ? shift[w.id] : 0u;
    const uint32_t *sd = seeds + 2 * w.off;
    for (int i = lane; i < n; i += 64) { const uint32_t qpl = sd[2 * i + 1]; m.fp[i] = (int)(sd[2 * i] - sh); m.fq[i] = (int)(qpl & 0xFFFFF); m.fl[i] = (int)(qpl >> 20); }
    LF_CLASP_SYNC();

    /* bl_slClusterSop (slchain.c:568-655): lane 0 scans, every cluster is chained as soon as it closes.  The scan
     * state is NOT reset at a cluster boundary (neither does the reference). */
    int sc_i = 0, min_yst = 0, max_yst = 0, min_yend = 0, max_yend = 0, cor = 0;
    if (lane == 0) { min_yst = max_yst = m.fq[0]; min_yend = max_yend = m.fq[0] + m.fl[0] - 1; }
    float bestScore = -1.0f; int bestFrag = -1;
    int begin = 0;
    while (begin < n) {
        int end = n - 1;
        if (lane == 0) {
            while (sc_i < n - 1) {
                const int a = sc_i, b = sc_i + 1;
                const int bs = m.fq[b], be = m.fq[b] + m.fl[b] - 1;
                if (bs < min_yst) min_yst = bs;
                if (bs > max_yst) max_yst = bs;
                if (be < min_yend) min_yend = be;
                if (be > max_yend) max_yend = be;
                /* max_score_per_pos stays 1.0: scr == len for every fragment (src/Chain.cpp:76-77) */
                if (max_yst > min_yend) cor = (int)((0.0 - 0.15) * (double)(max_yst - min_yend));
                const int fsb = m.fp[b], fea = m.fp[a] + m.fl[a] - 1;
                const bool cut = fsb > fea && fsb - fea >= max_yst - min_yend &&
                                 0.15 * (double)lf_clasp_d(fsb, fea) + cor > (double)(max_yend - min_yst + 1) * 1.0;
                sc_i++;
                if (cut) { end = a; break; }
            }
        }
        end = __shfl(end, 0);
        const int cm = end - begin + 1;
        if (cm == 1) {
            if (lane == 0) {
                m.chain_scr[begin] = (double)m.fl[begin]; m.chain_first[begin] = begin; m.chain_prev[begin] = -1;
                m.best_scr[begin] = (double)m.fl[begin]; m.best_base[begin] = begin; m.best_extra[begin] = -1;
            }
        } else lf_clasp_chain_sop(m, begin, cm, lane, &wr_tmp);
        if (lane == 0) {      /* src/Chain.cpp:128-147: first strictly greater, compared with the float kept so far */
            for (int j = begin; j <= end; j++)
                if (m.best_base[j] >= 0 && m.best_scr[j] > (double)bestScore) { bestScore = (float)m.best_scr[j]; bestFrag = j; }
        }
        LF_CLASP_SYNC();
        begin = end + 1;
    }
    if (lane == 0) {
        uint32_t len = 0;
        if (bestFrag >= 0) {
            for (int e = m.best_base[bestFrag]; e >= 0; e = m.chain_prev[e]) len++;
            uint32_t *out = chain_idx + w.off; uint32_t k = len;
            for (int e = m.best_base[bestFrag]; e >= 0; e = m.chain_prev[e]) out[--k] = (uint32_t)e;
            if (m.best_extra[bestFrag] >= 0) out[len++] = (uint32_t)m.best_extra[bestFrag];
        }
        chain_len[w.id] = len; score[w.id] = bestScore;
    }
}

#endif
