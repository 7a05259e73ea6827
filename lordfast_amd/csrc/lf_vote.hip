/*
 * lf_vote.hip -- candidate selection and chaining without leaving HBM (gfx950).
 *
 * The seed stage leaves every hit of the chunk in device memory (tPos, qPos|len, strand; hits of one read are
 * contiguous).  This stage turns them into chains; only the chains (a few dozen seeds per read) go to the host.
 *
 *   1-3 vote    one block per read, in LDS: every hit votes, with weight 1 + (len - k), for the windows floor(tPos/L) and
 *               floor(tPos/L) - 1 of its strand (src/LordFAST.cpp:588-620) into a hash table; local-maximum test (:630-632),
 *               best / second-best score, coarse / fine decision (:531-553), fine-mode candidate list (:875-877)
 *               (lf_vote_hash_kernel; no sort of the votes, see the comment there)
 *   4 requests  window -> [lo, hi] reference range clipped to the contig of the window's midpoint (:995-1003)
 *   5 gather    the read's hits of that strand inside the range, original order kept (:1004-1012)
 *   6 sort      by qPos (src/Chain.cpp:244 std::sort(compare_seed)).  A radix sort gives the unique order whenever all
 *               qPos of a request differ; requests with equal qPos get libstdc++'s introsort replayed by one lane on the
 *               original order (lf_stdsort.h), because the reference's tie order reaches the chain DP
 *   7 chain     lf_chain_n2_kernel (lf_chain.hip) on the device-resident requests; chains gathered and copied back
 *
 * Integer work; the hits are read once (9 B per hit) by the vote and once per request by the gather.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <mutex>
#include "lf_internal.h"
#include "lf_gpu_common.h"
#include "lf_scan.h"
#include "lf_chain_kernel.h"
#include "lf_reqsort.h"
#include "lf_clasp_kernel.h"

#define LF_STDSORT_FN static __device__
#include "lf_stdsort.h"

#define VK_WIN_BITS 26
#define VK_WMASK ((1u << VK_WIN_BITS) - 1u)

/* ---- 1-3: votes, local maxima, coarse / fine decision: ONE BLOCK PER READ, all in LDS ----
 * The reference adds every seed's weight to a dense per-thread array indexed by window (src/LordFAST.cpp:588-620), scans it
 * for local maxima (:630-632), keeps the best max_map of them in a min-heap (:634-654), sorts the heap (:528) and decides:
 * coarse if the best window scores >= 4 x the second (:531-549), else every local maximum above best / 4 is a fine-mode
 * candidate, forward list first, ascending (:553, :875-877).
 *
 * What reaches the output of that procedure is order-free: the largest and second-largest scores (a heap of >= 2
 * elements never evicts either), the window of the largest when it is unique (a tie can never be >= 4 x the second),
 * the number of candidates (0 / 1 / more), and the SET of local maxima above best / 4 -- whose order is the scan order,
 * i.e. ascending (strand, window).  So no sort of the votes is needed at all: a read's ~4 k votes go into an open-addressing
 * hash table in LDS (atomicAdd of the weight), the local-maximum test looks its two neighbours up in the same table, two
 * atomicMax passes find best and second, and only the handful of fine-mode candidates is ordered (rank by counting).
 * Round 1 radix-sorted and reduced all votes through HBM (hipCUB, 55 GB of traffic per 100 k reads for 7 GB of votes).
 * Reads with more votes than the largest LDS table keep theirs in a global scratch area (same code, GTAB). */
__device__ __forceinline__ uint32_t vk_hash(uint32_t key, uint32_t mask) { return (key * 2654435761u) >> 7 & mask; }
#define LF_VOTE_FILTER_WORDS 1024u
__device__ __forceinline__ uint32_t vk_hash2(uint32_t key) { return ((key ^ (key >> 15)) * 0x9E3779B1u) >> 17; }      /* 15 bits */

#define LF_VOTE_THREADS 1024
template <bool GTAB>
__global__ void __launch_bounds__(LF_VOTE_THREADS)
lf_vote_hash_kernel(int n_reads, const uint64_t *__restrict__ off, const uint64_t *__restrict__ read_off,
                    const uint32_t *__restrict__ tpos, const uint32_t *__restrict__ qpl, const uint8_t *__restrict__ strand,
                    uint32_t min_anchor_len, uint32_t l_pac, uint32_t min_read_len,
                    uint64_t votes_lo, uint64_t votes_hi, uint32_t lds_cap,
                    const uint64_t *__restrict__ gtab_off, uint32_t *__restrict__ gtab,
                    uint8_t *__restrict__ mode, uint32_t *__restrict__ nreq, int64_t *__restrict__ seg0,
                    uint32_t *__restrict__ stage, uint32_t *__restrict__ tmp_list, float *__restrict__ vscore, unsigned long long *__restrict__ dbg)
{
    extern __shared__ uint32_t s_tab[];
    __shared__ unsigned long long s_best;
    __shared__ uint32_t s_second, s_ncand, s_nlist;
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t a = read_off[r], b = read_off[r + 1], votes = 2 * (b - a);
    if (votes < votes_lo || votes > votes_hi) return;                       /* another launch's size class */
    unsigned long long t_prev = dbg ? __builtin_readcyclecounter() : 0;
#define VDBG(k) do { if (dbg && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&dbg[k], t_ - t_prev); t_prev = t_; } } while (0)
    uint32_t cap = lds_cap;
    if (GTAB) { cap = 4096; while ((uint64_t)cap < 2 * votes) cap <<= 1; }
    uint32_t *tab = GTAB ? gtab + gtab_off[r] : s_tab, *bits = tab + 2 * cap;       /* slot h: key + 1 at tab[2h], count at tab[2h + 1] (one 8-byte read) */
    const uint32_t mask = cap - 1;
    /* bits: a 32 Kbit presence filter over a second hash of the key.  Most windows are isolated, so most neighbour lookups
     * ask for a window that was never touched; the filter answers those with one LDS read instead of a probe sequence whose
     * length -- the longest among the 64 lanes -- the whole wavefront would walk */
    for (uint32_t i = tid; i < 2 * cap + LF_VOTE_FILTER_WORDS; i += LF_VOTE_THREADS) tab[i] = 0;
    if (tid == 0) { s_best = 0; s_second = 0; s_ncand = 0; s_nlist = 0; seg0[r] = (int64_t)(2 * a); }
    __syncthreads();
    VDBG(0);
    const uint32_t L = (uint32_t)(off[r + 1] - off[r]);
    auto insert = [&](uint32_t key, uint32_t w) {
        uint32_t h = vk_hash(key, mask);
        for (;;) {
            const uint32_t old = atomicCAS(&tab[2 * h], 0u, key + 1u);
            if (old == 0u) { const uint32_t fb = vk_hash2(key); atomicOr(&bits[fb >> 5], 1u << (fb & 31)); }
            if (old == 0u || old == key + 1u) { atomicAdd(&tab[2 * h + 1], w); return; }
            h = (h + 1) & mask;
        }
    };
    for (uint64_t j = a + tid; j < b; j += LF_VOTE_THREADS) {
        const uint32_t id = tpos[j] / L;
        const uint32_t w = (uint32_t)(1 + ((int32_t)(qpl[j] >> 20) - (int32_t)min_anchor_len));
        const uint32_t sk = (uint32_t)strand[j] << VK_WIN_BITS;
        insert(sk | id, w);
        if (id >= 1) insert(sk | (id - 1), w);                              /* windows are 2L wide, stride L (:612-619) */
    }
    __syncthreads();
    VDBG(1);
    auto lookup = [&](uint32_t key) -> int64_t {                           /* count of a window, -1 if it was never touched */
        const uint32_t fb = vk_hash2(key);
        if (!((bits[fb >> 5] >> (fb & 31)) & 1u)) return -1;
        uint32_t h = vk_hash(key, mask);
        for (;;) {
            const uint2 kc = *reinterpret_cast<const uint2 *>(&tab[2 * h]);
            if (kc.x == 0u) return -1;
            if (kc.x == key + 1u) return (int64_t)(kc.y & 0x7fffffffu);
            h = (h + 1) & mask;
        }
    };
    const uint32_t refWinNum = l_pac / min_read_len;                        /* src/LordFAST.cpp:130 */
    uint32_t lim = l_pac / L + 2; if (lim > refWinNum) lim = refWinNum;     /* :622-624 */
    /* pass A: local maxima below the window limit (:630-632); flagged in the top bit of their count.  Per-thread best /
     * count first, one atomic per wavefront (thousands of same-address LDS atomics serialise) */
    unsigned long long lbest = 0; uint32_t lcnt = 0;
    for (uint32_t i = tid; i < cap; i += LF_VOTE_THREADS) {
        const uint2 kc = *reinterpret_cast<const uint2 *>(&tab[2 * i]);
        const uint32_t k = kc.x;
        if (k == 0u) continue;
        const uint32_t key = k - 1u, id = key & VK_WMASK, c = kc.y & 0x7fffffffu;
        if (id >= lim) continue;
        bool ok = true;
        if (id != 0) { const int64_t cl = lookup(key - 1u); ok = cl < 0 || (int64_t)c >= cl; }
        if (ok && id != refWinNum - 1) { const int64_t cr = lookup(key + 1u); ok = cr < 0 || (int64_t)c > cr; }
        if (ok) {
            atomicOr(&tab[2 * i + 1], 0x80000000u);
            const unsigned long long v = ((unsigned long long)c << 32) | key;
            lbest = v > lbest ? v : lbest; lcnt++;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long x = __shfl_xor(lbest, o); lbest = x > lbest ? x : lbest;
        lcnt += __shfl_xor(lcnt, o);
    }
    if ((tid & 63) == 0 && lcnt) { atomicMax(&s_best, lbest); atomicAdd(&s_ncand, lcnt); }
    __syncthreads();
    VDBG(2);
    const uint32_t n_cand = s_ncand;
    if (n_cand == 0) { if (tid == 0) { mode[r] = 1; nreq[r] = 0; vscore[r] = 0; } return; }
    const uint32_t best_c = (uint32_t)(s_best >> 32), best_key = (uint32_t)s_best;
    /* pass B: the second-largest score among the local maxima (the largest one itself excluded once) */
    if (n_cand > 1) {
        uint32_t lsec = 0;
        for (uint32_t i = tid; i < cap; i += LF_VOTE_THREADS) {
            const uint2 kc = *reinterpret_cast<const uint2 *>(&tab[2 * i]); const uint32_t k = kc.x, cc = kc.y;
            if (k != 0u && (cc & 0x80000000u) && k - 1u != best_key) { const uint32_t v = cc & 0x7fffffffu; lsec = v > lsec ? v : lsec; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t x = __shfl_xor(lsec, o); lsec = x > lsec ? x : lsec; }
        if ((tid & 63) == 0 && lsec) atomicMax(&s_second, lsec);
        __syncthreads();
    }
    VDBG(3);
    const float top = (float)best_c, second = (float)s_second, scoreRatio = 4;
    const uint64_t sg = 2 * a;
    if (n_cand == 1 || top >= scoreRatio * second) {                         /* coarse (:531): one request, the best window */
        if (tid == 0) { mode[r] = 2; nreq[r] = 1; stage[sg] = (best_key & VK_WMASK) | ((best_key >> VK_WIN_BITS) << 31); vscore[r] = top; }
        return;
    }
    /* fine: every local maximum above best / 4 (:553, :875-877), in scan order = ascending (strand, window) */
    const float minScore = top / scoreRatio;
    for (uint32_t i = tid; i < cap; i += LF_VOTE_THREADS) {
        const uint2 kc = *reinterpret_cast<const uint2 *>(&tab[2 * i]); const uint32_t k = kc.x, cc = kc.y;
        if (k != 0u && (cc & 0x80000000u) && (float)(cc & 0x7fffffffu) > minScore) tmp_list[sg + atomicAdd(&s_nlist, 1u)] = k - 1u;
    }
    __syncthreads();
    const uint32_t nl = s_nlist;
    if (nl <= 64) {                                                          /* the usual handful: rank by counting */
        for (uint32_t e = tid; e < nl; e += LF_VOTE_THREADS) {
            const uint32_t key = tmp_list[sg + e];
            uint32_t rank = 0;
            for (uint32_t f = 0; f < nl; f++) rank += tmp_list[sg + f] < key;
            stage[sg + rank] = (key & VK_WMASK) | ((key >> VK_WIN_BITS) << 31);
        }
    } else {
        /* a read without a dominant window (best score of a few votes) makes every local maximum a candidate: thousands.
         * The table is not needed any more: its memory holds the list for a bitonic sort (nl <= 2/3 cap, so the next
         * power of two fits into the 2 cap words) */
        uint32_t P = 128; while (P < nl) P <<= 1;
        for (uint32_t i = tid; i < P; i += LF_VOTE_THREADS) tab[i] = i < nl ? tmp_list[sg + i] : 0xffffffffu;
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < P; i += LF_VOTE_THREADS) {
                    const uint32_t x = i ^ j;
                    if (x > i) {
                        const uint32_t va = tab[i], vb = tab[x];
                        if ((va > vb) == ((i & k) == 0)) { tab[i] = vb; tab[x] = va; }
                    }
                }
                __syncthreads();
            }
        for (uint32_t e = tid; e < nl; e += LF_VOTE_THREADS) { const uint32_t key = tab[e]; stage[sg + e] = (key & VK_WMASK) | ((key >> VK_WIN_BITS) << 31); }
    }
    VDBG(4);
    if (dbg && tid == 0) { atomicAdd(&dbg[5], 1ull); atomicAdd(&dbg[6], (unsigned long long)nl); if (nl > 64) atomicAdd(&dbg[7], 1ull); }
    if (tid == 0) { mode[r] = 3; nreq[r] = nl; vscore[r] = top; }
}

/* ---- 1-3 again, for the reads whose table fits into LDS: CELLS instead of windows, every cell looked at by its creator ----
 * lf_vote_hash_kernel inserts two votes per hit and finds the local maxima by scanning the whole table three times; it issued
 * more scalar than vector instructions (3.3 G against 2.6 G per 100 k reads: divergent probe loops and `continue`s) and ran at
 * the CU's scalar issue rate.  Two observations remove half of the work:
 *   - a hit in cell i = floor(tPos / L) votes for the windows i and i - 1 (src/LordFAST.cpp:612-619), so
 *     score(w) = cell(w) + cell(w + 1): ONE insert per hit into a table of cells (half the atomics, half the table -- twice
 *     the reads per CU), and the local-maximum test of window w (:630-632), score(w) >= score(w - 1) && score(w) > score(w + 1),
 *     is   cell(w + 1) >= cell(w - 1)  &&  cell(w) > cell(w + 2);
 *   - a distinct cell has exactly one creator, the thread whose compare-and-swap found the slot empty.  It keeps the slot in
 *     a register (H hits per thread, H a template parameter: all loops unrolled) and tests the cell's windows after the
 *     barrier: window c always; window c - 1 only when cell c - 1 does not exist (otherwise that cell's creator does it) --
 *     and that window can only be a maximum when it is the last one of the reference.  The table is never scanned.
 * Same candidate set, same scores, same outputs as the scan (which the reads above the LDS classes still take, GTAB). */
template <int H>
__global__ void __launch_bounds__(LF_VOTE_THREADS)
lf_vote_cell_kernel(int n_reads, const uint64_t *__restrict__ off, const uint64_t *__restrict__ read_off,
                    const uint32_t *__restrict__ tpos, const uint32_t *__restrict__ qpl, const uint8_t *__restrict__ strand,
                    uint32_t min_anchor_len, uint32_t l_pac, uint32_t min_read_len,
                    uint64_t hits_lo, uint64_t hits_hi, uint32_t cap,
                    uint8_t *__restrict__ mode, uint32_t *__restrict__ nreq, int64_t *__restrict__ seg0,
                    uint32_t *__restrict__ stage, uint32_t *__restrict__ tmp_list, float *__restrict__ vscore)
{
    extern __shared__ uint32_t s_tab[];
    __shared__ unsigned long long s_best;
    __shared__ uint32_t s_second, s_ncand, s_nlist;
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t a = read_off[r], b = read_off[r + 1], hits = b - a;
    if (hits < hits_lo || hits > hits_hi) return;                           /* another launch's size class */
    uint32_t *tab = s_tab, *bits = tab + 2 * cap;                            /* slot h: cell key + 1 at tab[2h], weight sum at tab[2h + 1] */
    const uint32_t mask = cap - 1;
    {   /* 2 cap + LF_VOTE_FILTER_WORDS words, 16 bytes per store */
        uint4 *t4 = reinterpret_cast<uint4 *>(tab);
        for (uint32_t i = tid; i < (2 * cap + LF_VOTE_FILTER_WORDS) / 4; i += LF_VOTE_THREADS) t4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid == 0) { s_best = 0; s_second = 0; s_ncand = 0; s_nlist = 0; seg0[r] = (int64_t)(2 * a); }
    /* the thread's hits: all loads in flight together */
    const uint32_t L = (uint32_t)(off[r + 1] - off[r]);
    uint32_t h_t[H], h_q[H], h_s[H];
#pragma unroll
    for (int u = 0; u < H; u++) {
        const uint64_t j = a + (uint64_t)tid + (uint64_t)u * LF_VOTE_THREADS;
        const bool in = j < b;
        h_t[u] = in ? tpos[j] : 0u; h_q[u] = in ? qpl[j] : 0u; h_s[u] = in ? (uint32_t)strand[j] : 0xffu;      /* 0xff: no hit */
    }
    __syncthreads();
    /* Probe loops are WAVE-SYNCHRONOUS: the loop condition is a ballot (one scalar branch per trip), everything inside is
     * predicated -- a lane that is done keeps walking without effect.  The compiler's own structurisation of per-lane `for (;;)`
     * probe loops with early exits cost ~25 scalar instructions per trip and lookup. */
    uint32_t own[H];                                                         /* slot of a cell this thread created, ~0 otherwise */
    {
        uint32_t key[H], wgt[H], hh[H], pend = 0;
#pragma unroll
        for (int u = 0; u < H; u++) {
            own[u] = ~0u;
            key[u] = (h_s[u] << VK_WIN_BITS) | (h_t[u] / L);
            wgt[u] = (uint32_t)(1 + ((int32_t)(h_q[u] >> 20) - (int32_t)min_anchor_len));
            hh[u] = vk_hash(key[u], mask);
            pend |= (h_s[u] != 0xffu ? 1u : 0u) << u;
        }
        while (lf_any(pend != 0u)) {
#pragma unroll
            for (int u = 0; u < H; u++) {
                if ((pend >> u) & 1u) {
                    const uint32_t old = atomicCAS(&tab[2 * hh[u]], 0u, key[u] + 1u);
                    if (old == 0u) { const uint32_t fb = vk_hash2(key[u]); atomicOr(&bits[fb >> 5], 1u << (fb & 31)); own[u] = hh[u]; }
                    if (old == 0u || old == key[u] + 1u) { atomicAdd(&tab[2 * hh[u] + 1], wgt[u]); pend &= ~(1u << u); }
                    hh[u] = (hh[u] + 1) & mask;
                }
            }
        }
    }
    __syncthreads();
    /* weight sums of three cells at once, 0 for a cell that does not exist (existing cells have >= 1); go = look at all */
    auto cells3 = [&](bool go, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t &c0, uint32_t &c1, uint32_t &c2) {
        const uint32_t f0 = vk_hash2(k0), f1 = vk_hash2(k1), f2 = vk_hash2(k2);
        bool p0 = go && ((bits[f0 >> 5] >> (f0 & 31)) & 1u), p1 = go && ((bits[f1 >> 5] >> (f1 & 31)) & 1u), p2 = go && ((bits[f2 >> 5] >> (f2 & 31)) & 1u);
        uint32_t h0 = vk_hash(k0, mask), h1 = vk_hash(k1, mask), h2 = vk_hash(k2, mask);
        c0 = 0; c1 = 0; c2 = 0;
        while (lf_any(p0 || p1 || p2)) {
            const uint2 a0 = *reinterpret_cast<const uint2 *>(&tab[2 * h0]), a1 = *reinterpret_cast<const uint2 *>(&tab[2 * h1]), a2 = *reinterpret_cast<const uint2 *>(&tab[2 * h2]);
            c0 = p0 && a0.x == k0 + 1u ? a0.y : c0; p0 = p0 && a0.x != 0u && a0.x != k0 + 1u; h0 = (h0 + 1) & mask;
            c1 = p1 && a1.x == k1 + 1u ? a1.y : c1; p1 = p1 && a1.x != 0u && a1.x != k1 + 1u; h1 = (h1 + 1) & mask;
            c2 = p2 && a2.x == k2 + 1u ? a2.y : c2; p2 = p2 && a2.x != 0u && a2.x != k2 + 1u; h2 = (h2 + 1) & mask;
        }
    };
    const uint32_t refWinNum = l_pac / min_read_len;                        /* src/LordFAST.cpp:130 */
    uint32_t lim = l_pac / L + 2; if (lim > refWinNum) lim = refWinNum;     /* :622-624 */
    /* windows of the thread's cells: local maxima below the window limit (:630-632).  Entry e: window c of cell u = e;
     * entry H + e: window c - 1 of the same cell (only without a cell c - 1) */
    uint32_t w_key[H], w_sc[H], w_cc[H];
    uint32_t is_max = 0;
    unsigned long long lbest = 0; uint32_t lcnt = 0;
#pragma unroll
    for (int e = 0; e < H; e++) {
        {
            const bool mine = own[e] != ~0u;
            const uint2 kc = *reinterpret_cast<const uint2 *>(&tab[2 * (mine ? own[e] : 0u)]);
            const uint32_t key = kc.x - 1u, c = key & VK_WMASK, cc = kc.y;
            uint32_t l1, r1, r2;
            cells3(mine, key - 1u, key + 1u, key + 2u, l1, r1, r2);
            l1 = c >= 1 ? l1 : 0u;                                         /* (key - 1 of cell 0 is another strand's or no cell at all) */
            const uint32_t sc = cc + r1;
            w_key[e] = mine ? key : 0u; w_sc[e] = mine ? sc : 0u; w_cc[e] = mine ? cc : 0u;
            if (mine && c < lim && (c == 0 || r1 >= l1) && (c == refWinNum - 1 || cc > r2)) {
                is_max |= 1u << e;
                const unsigned long long v = ((unsigned long long)sc << 32) | key;
                lbest = v > lbest ? v : lbest; lcnt++;
            }
            /* window c - 1 without a cell c - 1: score cc, never above its right neighbour's cc + r1 -- a maximum only as the
             * reference's last window, where the right neighbour is not looked at (a hit past the last window: rare) */
            const bool sec = mine && c >= 1 && l1 == 0u && c - 1 < lim && c - 1 == refWinNum - 1;
            if (lf_any(sec)) {
                uint32_t l2, d1, d2;
                cells3(sec && c >= 2, key - 2u, key - 2u, key - 2u, l2, d1, d2);
                if (sec && (c - 1 == 0 || cc >= l2)) {
                    is_max |= 1u << (H + e);
                    const unsigned long long v = ((unsigned long long)cc << 32) | (key - 1u);
                    lbest = v > lbest ? v : lbest; lcnt++;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long x = __shfl_xor(lbest, o); lbest = x > lbest ? x : lbest;
        lcnt += __shfl_xor(lcnt, o);
    }
    if ((tid & 63) == 0 && lcnt) { atomicMax(&s_best, lbest); atomicAdd(&s_ncand, lcnt); }
    __syncthreads();
    const uint32_t n_cand = s_ncand;
    if (n_cand == 0) { if (tid == 0) { mode[r] = 1; nreq[r] = 0; vscore[r] = 0; } return; }
    const uint32_t best_c = (uint32_t)(s_best >> 32), best_key = (uint32_t)s_best;
    /* the second-largest score among the local maxima (the largest one itself excluded once) */
    if (n_cand > 1) {
        uint32_t lsec = 0;
#pragma unroll
        for (int e = 0; e < H; e++) {
            const uint32_t v1 = ((is_max >> e) & 1u) && w_key[e] != best_key ? w_sc[e] : 0u; lsec = v1 > lsec ? v1 : lsec;
            const uint32_t v2 = ((is_max >> (H + e)) & 1u) && w_key[e] - 1u != best_key ? w_cc[e] : 0u; lsec = v2 > lsec ? v2 : lsec;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t x = __shfl_xor(lsec, o); lsec = x > lsec ? x : lsec; }
        if ((tid & 63) == 0 && lsec) atomicMax(&s_second, lsec);
        __syncthreads();
    }
    const float top = (float)best_c, second = (float)s_second, scoreRatio = 4;
    const uint64_t sg = 2 * a;
    if (n_cand == 1 || top >= scoreRatio * second) {                         /* coarse (:531): one request, the best window */
        if (tid == 0) { mode[r] = 2; nreq[r] = 1; stage[sg] = (best_key & VK_WMASK) | ((best_key >> VK_WIN_BITS) << 31); vscore[r] = top; }
        return;
    }
    /* fine: every local maximum above best / 4 (:553, :875-877), in scan order = ascending (strand, window) */
    const float minScore = top / scoreRatio;
#pragma unroll
    for (int e = 0; e < H; e++) {
        if (((is_max >> e) & 1u) && (float)w_sc[e] > minScore) tmp_list[sg + atomicAdd(&s_nlist, 1u)] = w_key[e];
        if (((is_max >> (H + e)) & 1u) && (float)w_cc[e] > minScore) tmp_list[sg + atomicAdd(&s_nlist, 1u)] = w_key[e] - 1u;
    }
    __syncthreads();
    const uint32_t nl = s_nlist;
    if (nl <= 64) {                                                          /* the usual handful: rank by counting */
        for (uint32_t e = tid; e < nl; e += LF_VOTE_THREADS) {
            const uint32_t key = tmp_list[sg + e];
            uint32_t rank = 0;
            for (uint32_t f = 0; f < nl; f++) rank += tmp_list[sg + f] < key;
            stage[sg + rank] = (key & VK_WMASK) | ((key >> VK_WIN_BITS) << 31);
        }
    } else {
        /* a read without a dominant window makes every local maximum a candidate: thousands.  The table is not needed any
         * more: its memory holds the list for a bitonic sort (at most one window per cell and cells <= 2/3 cap, so the next
         * power of two fits into the 2 cap words) */
        uint32_t P = 128; while (P < nl) P <<= 1;
        for (uint32_t i = tid; i < P; i += LF_VOTE_THREADS) tab[i] = i < nl ? tmp_list[sg + i] : 0xffffffffu;
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < P; i += LF_VOTE_THREADS) {
                    const uint32_t x = i ^ j;
                    if (x > i) {
                        const uint32_t va = tab[i], vb = tab[x];
                        if ((va > vb) == ((i & k) == 0)) { tab[i] = vb; tab[x] = va; }
                    }
                }
                __syncthreads();
            }
        for (uint32_t e = tid; e < nl; e += LF_VOTE_THREADS) { const uint32_t key = tab[e]; stage[sg + e] = (key & VK_WMASK) | ((key >> VK_WIN_BITS) << 31); }
    }
    if (tid == 0) { mode[r] = 3; nreq[r] = nl; vscore[r] = top; }
}

/* ---- 4: requests ---- */
__global__ void lf_req_build_kernel(int n_reads, const uint64_t *__restrict__ off, const uint32_t *__restrict__ nreq, const uint64_t *__restrict__ req0,
                                    const int64_t *__restrict__ seg0, const uint32_t *__restrict__ stage,
                                    const int64_t *__restrict__ ctg_off, const int64_t *__restrict__ ctg_len, int n_ctg, int64_t l_pac,
                                    uint32_t *__restrict__ req_read, uint32_t *__restrict__ req_win, int64_t *__restrict__ req_lo, int64_t *__restrict__ req_hi)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t L = (uint32_t)(off[r + 1] - off[r]), margin = L >> 1;
    const uint32_t n = nreq[r];
    const uint64_t q0 = req0[r];
    for (uint32_t c = 0; c < n; c++) {
        const uint32_t w = stage[seg0[r] + c], id = w & 0x7fffffffu;
        const uint32_t tStart = id * L, tEnd = (id + 2) * L - 1;                 /* uint32 arithmetic as in top_push */
        const int64_t mid = (int64_t)(((uint64_t)tStart + (uint64_t)tEnd) >> 1);  /* bwt_get_chr_boundaries: contig of the midpoint */
        int rid;
        if (mid >= l_pac) rid = n_ctg - 1;
        else { int lo = 0, hi = n_ctg - 1; while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (ctg_off[m] <= mid) lo = m; else hi = m - 1; } rid = lo; }
        const int64_t cb = (int64_t)(uint32_t)ctg_off[rid], ce = (int64_t)(uint32_t)(ctg_off[rid] + ctg_len[rid] - 1);
        const int64_t lo = ((int64_t)tStart - (int64_t)margin > cb) ? (int64_t)tStart - (int64_t)margin : cb;
        const int64_t hi = ((int64_t)tEnd + (int64_t)margin < ce) ? (int64_t)tEnd + (int64_t)margin : ce;
        req_read[q0 + c] = (uint32_t)r; req_win[q0 + c] = w; req_lo[q0 + c] = lo; req_hi[q0 + c] = hi;
    }
}

/* ---- 5: count / gather the hits of a request (one wavefront per request, order kept) ---- */
template <bool WRITE>
__global__ void __launch_bounds__(64)
lf_req_gather_kernel(int n_req, const uint32_t *__restrict__ req_read, const uint32_t *__restrict__ req_win, const int64_t *__restrict__ req_lo,
                     const int64_t *__restrict__ req_hi, const uint64_t *__restrict__ read_off, const uint32_t *__restrict__ tpos,
                     const uint32_t *__restrict__ qpl, const uint8_t *__restrict__ strand, uint32_t *__restrict__ req_n,
                     const uint64_t *__restrict__ req_off, uint2 *__restrict__ gathered, uint64_t *__restrict__ skeys, int key_by_tpos)
{
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= n_req) return;
    const uint32_t r = req_read[q], s = req_win[q] >> 31;
    const int64_t lo = req_lo[q], hi = req_hi[q];
    const uint64_t a = read_off[r], b = read_off[r + 1];
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint64_t out = WRITE ? req_off[q] : 0; uint32_t cnt = 0;
    for (uint64_t base = a; base < b; base += 64) {
        const uint64_t j = base + lane;
        bool in = false; uint32_t t = 0, ql = 0;
        if (j < b && strand[j] == s) { t = tpos[j]; in = (int64_t)t >= lo && (int64_t)t <= hi; if (WRITE && in) ql = qpl[j]; }
        const uint64_t m = lf_ballot(in);
        if (WRITE && in) {
            const uint64_t p = out + cnt + (uint32_t)__popcll(m & below);
            gathered[p] = make_uint2(t, ql);
            /* dp-n2 sorts a request by qPos (std::sort), clasp by target start (qsort = stable merge sort) */
            skeys[p] = key_by_tpos ? (((uint64_t)(uint32_t)q << 32) | t) : (((uint64_t)(uint32_t)q << 20) | (ql & 0xFFFFFu));
        }
        cnt += (uint32_t)__popcll(m);
    }
    /* (count pass: skeys = 256 words for the largest request, request q uses word q mod 256 -- 100 k wavefronts on ONE word took longer than the gather itself) */
    if (!WRITE && lane == 0) { req_n[q] = cnt; if (skeys) atomicMax(reinterpret_cast<unsigned int *>(skeys) + (q & 255), cnt); }
}

/* ---- 5b: a request's seeds in key order: the segmented LDS / HBM sorts of lf_reqsort.h ---- */
/* ---- 6: equal qPos inside a request -> replay std::sort on the original order ---- */
__global__ void lf_tie_flag_kernel(uint64_t n, const uint64_t *__restrict__ skeys_sorted, uint8_t *__restrict__ flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 || i >= n) return;
    if (skeys_sorted[i] == skeys_sorted[i - 1]) flag[skeys_sorted[i] >> 20] = 1;
}
__global__ void lf_count_flags_kernel(const uint8_t *__restrict__ flag, int n, unsigned long long *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t b = lf_ballot(i < n && flag[i] != 0);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned long long)__popcll(b));
}
struct lf_dseed { uint32_t tPos, qpl; };
#define DSEED_QLESS(a, b) (((a)->qpl & 0xFFFFFu) < ((b)->qpl & 0xFFFFFu))     /* compare_seed (src/Chain.cpp:227-230) */
LF_DEFINE_STDSORT(dseedq, lf_dseed, DSEED_QLESS)
__global__ void __launch_bounds__(64)
lf_tie_sort_kernel(int n_req, const uint8_t *__restrict__ flag, const uint64_t *__restrict__ req_off, const uint32_t *__restrict__ req_n,
                   const uint2 *__restrict__ gathered, uint2 *__restrict__ sorted)
{
    const int q = blockIdx.x;
    if (q >= n_req || !flag[q]) return;
    const uint64_t o = req_off[q]; const uint32_t n = req_n[q];
    for (uint32_t i = threadIdx.x; i < n; i += 64) sorted[o + i] = gathered[o + i];
    __syncthreads();
    if (threadIdx.x == 0) dseedq_sort(reinterpret_cast<lf_dseed *>(sorted + o), (long)n);
}

/* ---- 7: chain descriptors, chain gather ---- */
__global__ void lf_req_wins_kernel(int n_req, const uint64_t *__restrict__ req_off, const uint32_t *__restrict__ req_n, const uint64_t *__restrict__ ws_off,
                                   lf_chain_win *__restrict__ wins)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_req) return;
    lf_chain_win w; w.off = req_off[q]; w.n = req_n[q]; w.id = (uint32_t)q; w.ws_off = ws_off[q];
    wins[q] = w;
}
__global__ void lf_chain_gather_kernel(int n_req, const uint64_t *__restrict__ req_off, const uint32_t *__restrict__ chain_idx, const uint32_t *__restrict__ chain_len,
                                       const uint64_t *__restrict__ chain_off, const uint2 *__restrict__ sorted, uint2 *__restrict__ chain_seeds)
{
    const int q = blockIdx.x;
    if (q >= n_req) return;
    const uint32_t n = chain_len[q];
    const uint64_t o = req_off[q], co = chain_off[q];
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) chain_seeds[co + k] = sorted[o + chain_idx[o + k]];
}
struct lf_scan_big { const uint32_t *p; uint32_t lim; __device__ __forceinline__ uint64_t operator()(uint32_t i) const { const uint32_t n = p[i]; return n > lim ? (uint64_t)n : 0ull; } };   /* requests above the LDS classes: workspace in HBM */
/* clasp keeps positions in `int`: windows above 2e9 are shifted down (src/LordFAST.cpp:684-692, 1030-1046) */
__global__ void lf_req_shift_kernel(int n_req, const int64_t *__restrict__ req_lo, uint32_t *__restrict__ shift)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_req) shift[q] = req_lo[q] > 2000000000ll ? 2000000000u : 0u;
}
struct lf_twice { __host__ __device__ int operator()(uint64_t v) const { return (int)(2 * v); } };
struct lf_w8 { __host__ __device__ uint64_t operator()(uint8_t v) const { return v; } };

/* penalty table, evaluated exactly as score_penalty does (src/Chain.cpp:224) with the host libm; cached per device */
static int pen_table(int device, const lf_params_t *p, uint64_t want, hipStream_t s, const double **d_pen, uint32_t *pen_n)
{
    static std::mutex mu; static double *tab[16]; static uint64_t tab_n[16]; static double tab_cp[16];
    std::lock_guard<std::mutex> g(mu);
    if (want > (64u << 20)) want = 64u << 20;
    if (!tab[device] || tab_n[device] < want || tab_cp[device] != p->chain_penalty) {
        uint64_t n = 2 * want; if (n < (1u << 20)) n = 1u << 20;
        std::vector<double> pen((size_t)n);
        for (uint64_t d = 0; d < n; d++) pen[d] = d <= 1 ? 0.0 : 0.1 * (double)(int)d + p->chain_penalty * log((double)(int)d);
        /* an outgrown table is not freed: another lane may still have a launch in flight that reads it (rare, small) */
        HIPCHK(hipMalloc((void **)&tab[device], n * 8));
        HIPCHK(hipMemcpy(tab[device], pen.data(), n * 8, hipMemcpyHostToDevice));
        tab_n[device] = n; tab_cp[device] = p->chain_penalty;
    }
    (void)s;
    *d_pen = tab[device]; *pen_n = (uint32_t)tab_n[device];
    return LF_OK;
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int lfg_vote_chain(const struct lf_index *ix, const lf_params_t *p, int n_reads, uint64_t n_hits, uint32_t max_read_len, lfg_vc_t *out)
{
    memset(out, 0, sizeof *out);
    if (n_reads == 0) return LF_OK;
    const int dv = ix->device;
    HIPCHK(hipSetDevice(dv));
    hipStream_t s = (hipStream_t)lfg_lane_stream(dv, 0);
    if (!s) return LF_ERR_HIP;
    if (n_reads >= (1 << 20)) { lf_set_error("lfg_vote_chain: too many reads in one chunk"); return LF_ERR_ARG; }
    /* left in HBM by lfg_seed */
    const uint64_t *d_off = (const uint64_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 1, 0), *d_read_off = (const uint64_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 6, 0);
    const uint32_t *d_tpos = (const uint32_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 9, 0), *d_qpl = (const uint32_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 10, 0);
    const uint8_t *d_strand = (const uint8_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 11, 0);
    if (!d_off || !d_read_off || !d_tpos || !d_qpl || !d_strand) { lf_set_error("lfg_vote_chain: no resident seed batch"); return LF_ERR_ARG; }
#define VSLOT(k, bytes) lfg_dev_slot(dv, LF_DS_VOTE0 + (k), (bytes))
    hipEvent_t e0 = (hipEvent_t)lfg_lane_event(dv, 16), e1 = (hipEvent_t)lfg_lane_event(dv, 17), e2 = (hipEvent_t)lfg_lane_event(dv, 18);
    if (!e0 || !e1 || !e2) return LF_ERR_HIP;
    HIPCHK(hipEventRecord(e0, s));

    /* per-read arrays, one allocation */
    const size_t R = (size_t)n_reads;
    char *pr = (char *)VSLOT(0, al256(R) + al256(R * 4) + al256((R + 1) * 8) + al256(R * 8) + al256(R * 4) + 1024);
    if (!pr) return LF_ERR_NOMEM;
    uint8_t *d_mode = (uint8_t *)pr; pr += al256(R);
    uint32_t *d_nreq = (uint32_t *)pr; pr += al256(R * 4);
    uint64_t *d_req0 = (uint64_t *)pr; pr += al256((R + 1) * 8);
    int64_t *d_seg0 = (int64_t *)pr; pr += al256(R * 8);
    float *d_vscore = (float *)pr; pr += al256(R * 4);
    int *d_nruns = (int *)pr;
    uint64_t *h_small = (uint64_t *)lfg_pin_slot(LF_PS_VOTE0 + 0, 256 + 1024);
    if (!h_small) return LF_ERR_NOMEM;

    uint32_t n_req = 0;
    uint32_t *d_stage = nullptr;
    if (n_hits) {
        const uint64_t E = 2 * n_hits;
        if (E >= (1ull << 31)) { lf_set_error("lfg_vote_chain: %llu seed hits in one chunk exceed the 2^30 the vote sort takes; use smaller chunks (LF_CHUNK_READS)", (unsigned long long)n_hits); return LF_ERR_ARG; }      /* 2 votes per hit are indexed with 32 bits; lf_map_batch cuts a chunk in two before this can happen */
        d_stage = (uint32_t *)VSLOT(1, E * 4 + 16);
        uint32_t *d_tmp_list = (uint32_t *)VSLOT(2, E * 4 + 16);
        if (!d_stage || !d_tmp_list) return LF_ERR_NOMEM;
        /* LDS table classes (load factor <= 2/3): tables of cells (lf_vote_cell_kernel: hits <= 2/3 cap) or, with LF_VOTE_SCAN=1 /
         * LF_VOTE_DEBUG (A / B runs; the scan kernel's per-phase cycle counters), of windows (lf_vote_hash_kernel: votes <= 2/3
         * cap).  Reads above the largest class keep a table of windows in a global scratch area. */
        const bool vote_scan = lf_env_long("LF_VOTE_SCAN", 0) != 0;
        static const uint32_t caps[3] = { 4096, 8192, 16384 };
        static const uint32_t cell_caps[4] = { 2048, 4096, 8192, 16384 };
        uint64_t v_max_lds = vote_scan ? (uint64_t)caps[2] * 2 / 3 : 2 * ((uint64_t)cell_caps[3] * 2 / 3);      /* in votes = 2 x hits */
        if (lf_env_set("LF_VOTE_LDS_MAX_VOTES")) { const uint64_t x = strtoull(lf_env("LF_VOTE_LDS_MAX_VOTES"), nullptr, 10); if (x < v_max_lds) v_max_lds = x; }   /* test hook: push reads to the global-table path */
        const uint64_t *h_read_off = (const uint64_t *)lfg_pin_slot(LF_PS_HITS_OFF, 0);      /* lfg_seed left it there */
        uint64_t gtab_words = 0; std::vector<uint64_t> gtab_off;
        uint64_t v_max = 0;
        if (h_read_off) {
            for (int r = 0; r < n_reads; r++) {
                const uint64_t v = 2 * (h_read_off[r + 1] - h_read_off[r]);
                if (v > v_max) v_max = v;
                if (v > v_max_lds) {
                    if (gtab_off.empty()) gtab_off.assign((size_t)n_reads, 0);
                    uint64_t cap = 4096; while (cap < 2 * v) cap <<= 1;
                    gtab_off[(size_t)r] = gtab_words; gtab_words += 2 * cap + LF_VOTE_FILTER_WORDS;
                }
            }
        } else { lf_set_error("lfg_vote_chain: no resident seed batch (host offsets)"); return LF_ERR_ARG; }
        static bool attr_set[16] = { false };
        if (!attr_set[dv]) {
            const int big = 16384 * 8 + LF_VOTE_FILTER_WORDS * 4;
            HIPCHK(hipFuncSetAttribute((const void *)lf_vote_hash_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
            HIPCHK(hipFuncSetAttribute((const void *)lf_vote_cell_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
            HIPCHK(hipFuncSetAttribute((const void *)lf_vote_cell_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
            HIPCHK(hipFuncSetAttribute((const void *)lf_vote_cell_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
            HIPCHK(hipFuncSetAttribute((const void *)lf_vote_cell_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
            attr_set[dv] = true;
        }
        unsigned long long *d_dbg = nullptr;
        uint64_t lo = 0;                                         /* in votes */
        if (!vote_scan) {
            /* hits per thread of a class: hits <= 2/3 cap <= H x LF_VOTE_THREADS */
            uint64_t hlo = 0; const uint64_t h_max_lds = v_max_lds / 2;
            for (int k = 0; k < 4 && hlo <= h_max_lds; k++) {
                uint64_t hhi = (uint64_t)cell_caps[k] * 2 / 3; if (hhi > h_max_lds) hhi = h_max_lds;
                const size_t lds = (size_t)cell_caps[k] * 8 + LF_VOTE_FILTER_WORDS * 4;
#define LF_VOTE_CELL(HH) hipLaunchKernelGGL(lf_vote_cell_kernel<HH>, dim3((unsigned)n_reads), dim3(LF_VOTE_THREADS), lds, s, n_reads, d_off, d_read_off, d_tpos, d_qpl, d_strand, \
                                   (uint32_t)p->min_anchor_len, (uint32_t)ix->l_pac, (uint32_t)p->min_read_len, hlo, hhi, cell_caps[k], d_mode, d_nreq, d_seg0, d_stage, d_tmp_list, d_vscore)
                if (2 * hlo <= v_max) { if (k == 0) LF_VOTE_CELL(2); else if (k == 1) LF_VOTE_CELL(3); else if (k == 2) LF_VOTE_CELL(6); else LF_VOTE_CELL(11); }
#undef LF_VOTE_CELL
                hlo = hhi + 1;
            }
            lo = 2 * h_max_lds + 2;
        } else
        for (int k = 0; k < 3 && lo <= v_max_lds; k++) {
            uint64_t hi = (uint64_t)caps[k] * 2 / 3; if (hi > v_max_lds) hi = v_max_lds;
            if (lo <= v_max)
                hipLaunchKernelGGL(lf_vote_hash_kernel<false>, dim3((unsigned)n_reads), dim3(LF_VOTE_THREADS), (size_t)caps[k] * 8 + LF_VOTE_FILTER_WORDS * 4, s, n_reads, d_off, d_read_off, d_tpos, d_qpl, d_strand,
                                   (uint32_t)p->min_anchor_len, (uint32_t)ix->l_pac, (uint32_t)p->min_read_len, lo, hi, caps[k],
                                   (const uint64_t *)nullptr, (uint32_t *)nullptr, d_mode, d_nreq, d_seg0, d_stage, d_tmp_list, d_vscore, d_dbg);
            lo = hi + 1;
        }
        if (gtab_words) {
            uint64_t *d_gtab_off = (uint64_t *)VSLOT(3, (size_t)n_reads * 8);
            uint32_t *d_gtab = (uint32_t *)VSLOT(4, gtab_words * 4 + 16);
            if (!d_gtab_off || !d_gtab) return LF_ERR_NOMEM;
            HIPCHK(hipMemcpyAsync(d_gtab_off, gtab_off.data(), (size_t)n_reads * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(lf_vote_hash_kernel<true>, dim3((unsigned)n_reads), dim3(LF_VOTE_THREADS), 0, s, n_reads, d_off, d_read_off, d_tpos, d_qpl, d_strand,
                               (uint32_t)p->min_anchor_len, (uint32_t)ix->l_pac, (uint32_t)p->min_read_len, lo, ~0ull, 0u,
                               (const uint64_t *)d_gtab_off, d_gtab, d_mode, d_nreq, d_seg0, d_stage, d_tmp_list, d_vscore, d_dbg);
            HIPCHK(hipStreamSynchronize(s));                    /* gtab_off (host vector) is read by the copy above */
        }
        if (d_dbg) { unsigned long long h[8]; HIPCHK(hipMemcpyAsync(h, d_dbg, 64, hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s));
            fprintf(stderr, "[lf] vote kernel cycles (sum over blocks, thread 0): zero %llu insert %llu passA %llu passB %llu fine %llu | fine reads %llu, candidates %llu, lists > 64: %llu (of %d reads)\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], n_reads); }
        /* request ids: exclusive scan over n_reads + 1 counts (the last one is a zero pad) */
        { lf_scan_u32 f; f.p = d_nreq; const int src = lf_scan_excl(dv, 1, s, f, d_req0, (size_t)n_reads); if (src != LF_OK) return src; }
        HIPCHK(hipMemcpyAsync(h_small, d_req0 + (R - 1), 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h_small + 1, d_nreq + (R - 1), 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const uint64_t nr = h_small[0] + (uint32_t)h_small[1];
        if (nr >= (1ull << 31)) { lf_set_error("lfg_vote_chain: too many candidate windows"); return LF_ERR_ARG; }
        n_req = (uint32_t)nr;
    } else {
        HIPCHK(hipMemsetAsync(d_mode, 1, R, s));
        HIPCHK(hipMemsetAsync(d_nreq, 0, R * 4, s));
        HIPCHK(hipMemsetAsync(d_req0, 0, (R + 1) * 8, s));
    }
    HIPCHK(hipEventRecord(e1, s));

    /* host-side result arrays (pinned, valid until this lane's next call) */
    out->n_reads = n_reads; out->n_req = (int)n_req;
    out->mode = (uint8_t *)lfg_pin_slot(LF_PS_VOTE0 + 1, R + 16);
    out->req0 = (uint64_t *)lfg_pin_slot(LF_PS_VOTE0 + 2, (R + 1) * 8);
    out->nreq = (uint32_t *)lfg_pin_slot(LF_PS_VOTE0 + 3, R * 4 + 16);
    out->vscore = (float *)lfg_pin_slot(LF_PS_VOTE0 + 4, R * 4 + 16);
    if (!out->mode || !out->req0 || !out->nreq || !out->vscore) return LF_ERR_NOMEM;
    HIPCHK(hipMemcpyAsync(out->mode, d_mode, R, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out->req0, d_req0, R * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out->nreq, d_nreq, R * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out->vscore, d_vscore, R * 4, hipMemcpyDeviceToHost, s));
    if (n_req == 0) {
        HIPCHK(hipStreamSynchronize(s));
        out->req0[R] = 0;
        HIPCHK(hipEventElapsedTime(&out->ms_vote, e0, e1));
        return LF_OK;
    }

    /* ---- requests ---- */
    const size_t Q = (size_t)n_req;
    char *pq = (char *)VSLOT(6, al256(Q * 4) * 4 + al256(Q * 8) * 2 + al256((Q + 1) * 8) * 3 + al256(Q) + al256(Q * sizeof(lf_chain_win)) + 256 + 1024);
    int64_t *d_ctg = (int64_t *)VSLOT(7, (size_t)ix->n_seqs * 16 + 64);
    if (!pq || !d_ctg) return LF_ERR_NOMEM;
    uint32_t *d_req_read = (uint32_t *)pq; pq += al256(Q * 4);
    uint32_t *d_req_win = (uint32_t *)pq; pq += al256(Q * 4);
    uint32_t *d_req_n = (uint32_t *)pq; pq += al256(Q * 4);
    uint32_t *d_clen = (uint32_t *)pq; pq += al256(Q * 4);
    int64_t *d_req_lo = (int64_t *)pq; pq += al256(Q * 8);
    int64_t *d_req_hi = (int64_t *)pq; pq += al256(Q * 8);
    uint64_t *d_req_off = (uint64_t *)pq; pq += al256((Q + 1) * 8);
    uint64_t *d_ws_off = (uint64_t *)pq; pq += al256((Q + 1) * 8);
    uint64_t *d_coff = (uint64_t *)pq; pq += al256((Q + 1) * 8);
    uint8_t *d_flag = (uint8_t *)pq; pq += al256(Q);
    lf_chain_win *d_wins = (lf_chain_win *)pq; pq += al256(Q * sizeof(lf_chain_win));
    uint64_t *d_maxn = (uint64_t *)pq;                      /* the largest request (sizes the segmented sort's LDS) */
    float *d_cscore = (float *)VSLOT(8, Q * 4 + 64);
    if (!d_cscore) return LF_ERR_NOMEM;
    {
        int64_t *h_ctg = (int64_t *)lfg_pin_slot(LF_PS_VOTE0 + 5, (size_t)ix->n_seqs * 16 + 64);
        if (!h_ctg) return LF_ERR_NOMEM;
        for (int i = 0; i < ix->n_seqs; i++) { h_ctg[i] = ix->contigs[i].offset; h_ctg[ix->n_seqs + i] = ix->contigs[i].len; }
        HIPCHK(hipMemcpyAsync(d_ctg, h_ctg, (size_t)ix->n_seqs * 16, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(lf_req_build_kernel, dim3((unsigned)((n_reads + 127) / 128)), dim3(128), 0, s, n_reads, d_off, d_nreq, d_req0, d_seg0, d_stage,
                       d_ctg, d_ctg + ix->n_seqs, ix->n_seqs, (int64_t)ix->l_pac, d_req_read, d_req_win, d_req_lo, d_req_hi);
    HIPCHK(hipMemsetAsync(d_maxn, 0, 1024, s));
    hipLaunchKernelGGL(lf_req_gather_kernel<false>, dim3((unsigned)n_req), dim3(64), 0, s, (int)n_req, d_req_read, d_req_win, d_req_lo, d_req_hi, d_read_off,
                       d_tpos, d_qpl, d_strand, d_req_n, (const uint64_t *)nullptr, (uint2 *)nullptr, d_maxn, 0);
    const bool clasp = p->chain_alg == 1;
    const uint32_t big_lim = clasp ? LF_CLASP_LDS_MAX : LF_CHAIN_LDS_MAX;
    { lf_scan_u32 f; f.p = d_req_n; const int src = lf_scan_excl(dv, 1, s, f, d_req_off, (size_t)n_req); if (src != LF_OK) return src; }
    { lf_scan_big f; f.p = d_req_n; f.lim = big_lim; const int src = lf_scan_excl(dv, 1, s, f, d_ws_off, (size_t)n_req); if (src != LF_OK) return src; }
    HIPCHK(hipMemcpyAsync(h_small + 2, d_req_off + (Q - 1), 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_small + 3, d_ws_off + (Q - 1), 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_small + 4, d_req_n + (Q - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_small + 32, d_maxn, 1024, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint32_t max_n = 0;
    for (int u = 0; u < 256; u++) { const uint32_t x = reinterpret_cast<const uint32_t *>(h_small + 32)[u]; if (x > max_n) max_n = x; }
    const uint32_t last_n = (uint32_t)h_small[4];
    const uint64_t S = h_small[2] + last_n, WS = h_small[3] + (last_n > big_lim ? last_n : 0);
    if (S >= (1ull << 31)) { lf_set_error("lfg_vote_chain: too many seeds in candidate windows (%llu)", (unsigned long long)S); return LF_ERR_ARG; }
    out->n_req_seeds = S;

    uint2 *d_gath = (uint2 *)VSLOT(10, S * 8 + 64), *d_sorted = (uint2 *)VSLOT(11, S * 8 + 64);
    uint64_t *d_sk = (uint64_t *)VSLOT(12, S * 8 + 64), *d_sk2 = (uint64_t *)VSLOT(13, S * 8 + 64);
    uint32_t *d_cidx = (uint32_t *)VSLOT(14, S * 4 + 64);
    double *d_dp = (double *)VSLOT(15, WS * (clasp ? (uint64_t)LF_CLASP_BYTES_PER_FRAG : 8) + 64); int *d_prev = (int *)VSLOT(16, clasp ? Q * 4 + 64 : WS * 4 + 64);
    if (!d_gath || !d_sorted || !d_sk || !d_sk2 || !d_cidx || !d_dp || !d_prev) return LF_ERR_NOMEM;
    bool have_ties = false;
    if (S) {
        hipLaunchKernelGGL(lf_req_gather_kernel<true>, dim3((unsigned)n_req), dim3(64), 0, s, (int)n_req, d_req_read, d_req_win, d_req_lo, d_req_hi, d_read_off,
                           d_tpos, d_qpl, d_strand, d_req_n, (const uint64_t *)d_req_off, d_gath, d_sk, clasp ? 1 : 0);
        {
            /* (LF_REQ_SORT_BIG_FROM lowers the bound from which the HBM-scratch form is used: test hook) */
            const int src_ = lf_req_sort_launch(dv, s, (int)n_req, (const uint64_t *)d_req_off, (const uint32_t *)d_req_n, (const uint2 *)d_gath, d_sorted, d_sk2, clasp ? 1 : 0, max_n,
                                                (uint32_t)lf_env_long("LF_REQ_SORT_BIG_FROM", 8193), (uint64_t *)VSLOT(17, max_n > 8192u || lf_env_set("LF_REQ_SORT_BIG_FROM") ? 2 * S * 8 + 256 : 16));
            if (src_ != LF_OK) return src_;
        }
        HIPCHK(hipMemsetAsync(d_flag, 0, Q, s));
        if (!clasp) {
        hipLaunchKernelGGL(lf_tie_flag_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, s, S, d_sk2, d_flag);
        hipLaunchKernelGGL(lf_tie_sort_kernel, dim3((unsigned)n_req), dim3(64), 0, s, (int)n_req, d_flag, d_req_off, d_req_n, d_gath, d_sorted);
        {   /* how many requests needed the introsort replay (statistics) */
            uint64_t *d_nt = (uint64_t *)d_nruns + 1;
            HIPCHK(hipMemsetAsync(d_nt, 0, 8, s));
            hipLaunchKernelGGL(lf_count_flags_kernel, dim3((unsigned)((n_req + 255) / 256)), dim3(256), 0, s, (const uint8_t *)d_flag, (int)n_req, (unsigned long long *)d_nt);
            HIPCHK(hipMemcpyAsync(h_small + 8, d_nt, 8, hipMemcpyDeviceToHost, s)); have_ties = true;
        }
        }
    }
    /* ---- chains ---- */
    uint64_t max_d = 4ull * max_read_len + 8192;
    const double *d_pen; uint32_t pen_n;
    { int rc = pen_table(dv, p, max_d, s, &d_pen, &pen_n); if (rc != LF_OK) return rc; }
    const double reward = p->chain_reward * (double)p->min_anchor_len;     /* score_reward, src/Chain.cpp:211-215 */
    hipLaunchKernelGGL(lf_req_wins_kernel, dim3((unsigned)((n_req + 255) / 256)), dim3(256), 0, s, (int)n_req, d_req_off, d_req_n, d_ws_off, d_wins);
    if (clasp) {
        uint32_t *d_shift = (uint32_t *)d_prev;
        hipLaunchKernelGGL(lf_req_shift_kernel, dim3((unsigned)((n_req + 255) / 256)), dim3(256), 0, s, (int)n_req, d_req_lo, d_shift);
        /* LDS size classes (136 B per fragment: 18 / 12 / 9 / 7 / 6 / 4 / 3 / 2 / 1 / 1 windows per CU), the HBM-workspace class last.
         * A window is one wavefront with a long dependent chain, so the classes run CONCURRENTLY on their own streams
         * (the ones the alignment stage of this lane uses later), largest first: back to back on one stream the 512 class
         * -- one window per CU -- would hold the whole stage up. */
        enum { NCC = 11 };
        static const uint32_t CCAPS[NCC] = { 64, 96, 128, 160, 192, 256, 384, 512, 768, LF_CLASP_LDS_MAX, 0 };
        hipEvent_t ev[NCC + 1];
        for (int k = 0; k <= NCC; k++) { ev[k] = (hipEvent_t)lfg_lane_event(dv, 20 + k); if (!ev[k]) return LF_ERR_HIP; }
        HIPCHK(hipEventRecord(ev[NCC], s));
        for (int c = NCC - 1; c >= 0; c--) {
            const uint32_t lo = c == 0 ? 0u : CCAPS[c - 1] + 1u, hi = CCAPS[c] ? CCAPS[c] : 0xFFFFFFFFu;
            if (c == NCC - 1 && WS == 0) continue;
            hipStream_t cs = (hipStream_t)lfg_lane_stream(dv, 2 + c);
            if (!cs) return LF_ERR_HIP;
            HIPCHK(hipStreamWaitEvent(cs, ev[NCC], 0));
            const size_t smem = CCAPS[c] ? (size_t)CCAPS[c] * LF_CLASP_LDS_BYTES_PER_FRAG : 16;
            if (CCAPS[c]) hipLaunchKernelGGL(lf_clasp_kernel<true>, dim3((unsigned)n_req), dim3(64), smem, cs, (const lf_chain_win *)d_wins, (int)n_req,
                               (const uint32_t *)d_sorted, (const uint32_t *)d_shift, CCAPS[c], (unsigned char *)d_dp, d_cidx, d_clen, d_cscore, lo, hi);
            else hipLaunchKernelGGL(lf_clasp_kernel<false>, dim3((unsigned)n_req), dim3(64), smem, cs, (const lf_chain_win *)d_wins, (int)n_req,
                               (const uint32_t *)d_sorted, (const uint32_t *)d_shift, CCAPS[c], (unsigned char *)d_dp, d_cidx, d_clen, d_cscore, lo, hi);
            HIPCHK(hipEventRecord(ev[c], cs));
            HIPCHK(hipStreamWaitEvent(s, ev[c], 0));
        }
    } else {
        const int lrc = lf_chain_n2_launch_classes(dv, s, (const lf_chain_win *)d_wins, (int)n_req, (const uint32_t *)d_sorted, d_pen, pen_n, reward, p->chain_penalty,
                                                   d_dp, d_prev, WS != 0, d_cidx, d_clen, d_cscore, max_n);
        if (lrc != LF_OK) return lrc;
    }
    { lf_scan_u32 f; f.p = d_clen; const int src = lf_scan_excl(dv, 1, s, f, d_coff, (size_t)n_req); if (src != LF_OK) return src; }
    /* a chain is a subset of its request's seeds: S bounds the chains' total, so the gather needs no readback of it; and the chains stay in HBM --
     * the device walk (lf_walk.hip) reads them there, the host fetches the few it replays itself (lf_pipeline.c) or all of them on request
     * (lfg_vote_fetch_chains: the cross-check paths) */
    uint2 *d_cseeds = (uint2 *)VSLOT(19, S * 8 + 64);
    if (!d_cseeds) return LF_ERR_NOMEM;
    hipLaunchKernelGGL(lf_chain_gather_kernel, dim3((unsigned)n_req), dim3(64), 0, s, (int)n_req, d_req_off, d_cidx, d_clen, d_coff, d_sorted, d_cseeds);
    HIPCHK(hipEventRecord(e2, s));
    out->req_win = (uint32_t *)lfg_pin_slot(LF_PS_VOTE0 + 6, Q * 4 + 16);
    out->chain_len = (uint32_t *)lfg_pin_slot(LF_PS_VOTE0 + 7, Q * 4 + 16);
    out->chain_score = (float *)lfg_pin_slot(LF_PS_VOTE0 + 8, Q * 4 + 16);
    out->chain_off = (uint64_t *)lfg_pin_slot(LF_PS_VOTE0 + 9, (Q + 1) * 8);
    if (!out->req_win || !out->chain_len || !out->chain_score || !out->chain_off) return LF_ERR_NOMEM;
    HIPCHK(hipMemcpyAsync(out->req_win, d_req_win, Q * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out->chain_len, d_clen, Q * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out->chain_score, d_cscore, Q * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out->chain_off, d_coff, Q * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    const uint64_t C = out->chain_off[Q - 1] + out->chain_len[Q - 1];
    if (C > S) { lf_set_error("lfg_vote_chain: %llu chain seeds out of %llu request seeds", (unsigned long long)C, (unsigned long long)S); return LF_ERR_HIP; }
    out->chain_seeds = nullptr;
    out->chain_off[Q] = C; out->req0[R] = n_req;
    out->n_tie_req = have_ties ? h_small[8] : 0;
    out->n_chain_seeds = C;
    out->d_chain_seeds = d_cseeds; out->d_chain_off = d_coff; out->d_chain_len = d_clen; out->d_ctg = d_ctg;
    HIPCHK(hipEventElapsedTime(&out->ms_vote, e0, e1));
    HIPCHK(hipEventElapsedTime(&out->ms_chain, e1, e2));
#undef VSLOT
    return LF_OK;
}

/* all chains of the lane's last lfg_vote_chain, copied to a pinned host array (valid until the lane's next call): for the callers that walk every
 * chain on the host (lf_debug_crosscheck's host walk / host CIGAR modes) */
extern "C" int lfg_vote_fetch_chains(const struct lf_index *ix, lfg_vc_t *vc)
{
    if (vc->chain_seeds || !vc->n_chain_seeds || !vc->d_chain_seeds) return LF_OK;
    HIPCHK(hipSetDevice(ix->device));
    hipStream_t s = (hipStream_t)lfg_lane_stream(ix->device, 0);
    if (!s) return LF_ERR_HIP;
    Seed_t *h = (Seed_t *)lfg_pin_slot(LF_PS_VOTE0 + 10, vc->n_chain_seeds * 8 + 16);
    if (!h) return LF_ERR_NOMEM;
    HIPCHK(hipMemcpyAsync(h, vc->d_chain_seeds, vc->n_chain_seeds * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    vc->chain_seeds = h;
    return LF_OK;
}
