/* lf_chain_kernel.h -- the dp-n2 chain kernel, shared by lf_chain.hip (host-supplied windows) and lf_vote.hip
 * (windows built on the device).  See lf_chain.hip for the algorithm notes. */
#ifndef LF_CHAIN_KERNEL_H
#define LF_CHAIN_KERNEL_H
#include "lf_gpu_common.h"
#include <math.h>

#define LF_CHAIN_LDS_MAX 4096

struct lf_chain_win { uint64_t off; uint32_t n; uint32_t id; uint64_t ws_off; };

__device__ __forceinline__ void lf_wave_argmax(double &v, int &j)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(v, o);
        const int j2 = __shfl_xor(j, o);
        if (v2 > v || (v2 == v && j2 > j)) { v = v2; j = j2; }
    }
}

static __global__ void __launch_bounds__(64)
lf_chain_n2_kernel(const lf_chain_win *__restrict__ wins, int n_wins, const uint32_t *__restrict__ seeds /* tPos, qpl pairs */,
                   const double *__restrict__ pen, uint32_t pen_n, double reward, double chain_penalty, uint32_t cap,
                   double *__restrict__ ws_dp, int *__restrict__ ws_prev,
                   uint32_t *__restrict__ chain_idx, uint32_t *__restrict__ chain_len, float *__restrict__ score,
                   uint32_t n_min, uint32_t n_max /* size class served by this launch (inclusive) */)
{
    /* dynamic LDS, carved for `cap` seeds (launch-time, per size class): dp | tPos | qPos | prev | len */
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *s_dp = reinterpret_cast<double *>(smem);
    uint32_t *s_t = reinterpret_cast<uint32_t *>(smem + (size_t)cap * 8);
    uint32_t *s_q = s_t + cap;
    int *s_prev = reinterpret_cast<int *>(s_q + cap);
    uint16_t *s_l = reinterpret_cast<uint16_t *>(s_prev + cap);
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= n_wins) return;
    const lf_chain_win w = wins[blockIdx.x];
    if (w.n < n_min || w.n > n_max) return;
    const int n = (int)w.n;
    const uint32_t *sd = seeds + 2 * w.off;
    const bool in_lds = n <= (int)cap;
    double *dp = in_lds ? s_dp : ws_dp + w.ws_off;
    int *prev = in_lds ? s_prev : ws_prev + w.ws_off;
    if (in_lds) {
        for (int i = lane; i < n; i += 64) { const uint32_t qpl = sd[2 * i + 1]; s_t[i] = sd[2 * i]; s_q[i] = qpl & 0xFFFFF; s_l[i] = (uint16_t)(qpl >> 20); }
        __syncthreads();
    }
    double best = -1; int bestIdx = -1;
    for (int i = 0; i < n; i++) {
        uint32_t ti, qi, li;
        if (in_lds) { ti = s_t[i]; qi = s_q[i]; li = s_l[i]; }
        else { const uint32_t qpl = sd[2 * i + 1]; ti = sd[2 * i]; qi = qpl & 0xFFFFF; li = qpl >> 20; }
        double bv = -1.0e300; int bj = -1;
        for (int j = i - 1 - lane; j >= 0; j -= 64) {
            uint32_t tj, qj, lj;
            if (in_lds) { tj = s_t[j]; qj = s_q[j]; lj = s_l[j]; }
            else { const uint32_t qpl = sd[2 * j + 1]; tj = sd[2 * j]; qj = qpl & 0xFFFFF; lj = qpl >> 20; }
            const int distR = (int)qi - ((int)qj + (int)lj - 1);
            if (distR <= 0) continue;
            const int distT = (int)(ti - (tj + lj - 1));
            if (distT <= 0) continue;
            const uint32_t d = (uint32_t)(distR < distT ? distT - distR : distR - distT);
            /* score_penalty (src/Chain.cpp:217-225); the table covers every distance a window can produce */
            const double pn = d <= 1 ? 0.0 : (d < pen_n ? pen[d] : 0.1 * (double)(int)d + chain_penalty * log((double)(int)d));
            const double cand = (dp[j] + reward) - pn;
            if (cand > bv) { bv = cand; bj = j; }      /* j descends within a lane: first hit = largest j */
        }
        lf_wave_argmax(bv, bj);
        double di = (double)li; int pi = -1;
        if (bj >= 0 && bv > di) { di = bv; pi = bj; }
        if (lane == 0) { dp[i] = di; prev[i] = pi; }
        if (di > best) { best = di; bestIdx = i; }
        __syncthreads();
    }
    if (lane == 0) {
        uint32_t len = 0;
        for (int k = bestIdx; k != -1; k = prev[k]) len++;
        uint32_t wpos = len;
        uint32_t *out = chain_idx + w.off;
        for (int k = bestIdx; k != -1; k = prev[k]) out[--wpos] = (uint32_t)k;
        chain_len[w.id] = len;
        score[w.id] = (float)best;
    }
}

/* ---- LARGE windows (round 6): a lane OWNS a seed i, a workgroup of W wavefronts walks the window 64 seeds at a time.
 * The kernel above gives a window to one wavefront and spreads the inner loop (j) of every i over its lanes: a wave arg-max (twelve shuffles) and a barrier
 * per seed, ceil(i / 64) trips in between -- 232 ms for one satellite-array window of config C5 (thousands of seeds; the reference's loop is
 * src/Chain.cpp:246-283).  Here lane l of every wavefront owns seed i = b0 + l of the current block of 64:
 *   phase 1   the predecessors j < b0 are cut into W slices, one per wavefront; a lane walks its slice downwards with its own running best -- no exchange
 *             between lanes at all (t, q, len of seed j are wave-uniform loads, dp[j] is an LDS broadcast); the W partial results of a seed meet in LDS,
 *             largest j first, a smaller j only when strictly better: the reference's `>` while j descends;
 *   phase 2   the 64 seeds of the block among themselves, in wavefront 0: step s finalises seed b0 + s (lane s) and hands its (dp, t, q, len) to the lanes
 *             above it by readlane; they take it when it is at least as good as what they hold (`>=` while j ASCENDS keeps the largest j among equals).
 * Two barriers per 64 seeds instead of one per seed, and the O(n^2) part runs on W wavefronts.  dp lives in LDS (8 B per seed, up to LF_CHAIN_BIG_MAX seeds),
 * prev in LDS (PREV_LDS: windows up to LF_CHAIN_LDS_MAX) or in the request's HBM workspace.  Same double-precision expression and evaluation order as
 * the reference (-ffp-contract=off). */
#define LF_CHAIN_BIG_W 8
#define LF_CHAIN_BIG_FROM 513
#define LF_CHAIN_BIG_MAX 16384
template <int W, bool PREV_LDS>
static __global__ void __launch_bounds__(64 * W)
lf_chain_n2_big_kernel(const lf_chain_win *__restrict__ wins, int n_wins, const uint32_t *__restrict__ seeds /* tPos, qpl pairs */,
                       const double *__restrict__ pen, uint32_t pen_n, double reward, double chain_penalty, uint32_t cap,
                       int *__restrict__ ws_prev, uint32_t *__restrict__ chain_idx, uint32_t *__restrict__ chain_len, float *__restrict__ score,
                       uint32_t n_min, uint32_t n_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *s_dp = reinterpret_cast<double *>(smem);
    double *red_v = s_dp + cap;
    int *red_j = reinterpret_cast<int *>(red_v + W * 64);
    int *s_prev = red_j + W * 64;                                  /* PREV_LDS only */
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    if ((int)blockIdx.x >= n_wins) return;
    const lf_chain_win w = wins[blockIdx.x];
    if (w.n < n_min || w.n > n_max) return;
    const int n = (int)w.n;
    const uint32_t *sd = seeds + 2 * w.off;
    int *prev = PREV_LDS ? s_prev : ws_prev + w.ws_off;
    double best = -1; int bestIdx = -1;                            /* (wavefront 0) */
    auto cost = [&](int distR, int distT) -> double {               /* score_penalty (src/Chain.cpp:217-225) */
        const uint32_t d = (uint32_t)(distR < distT ? distT - distR : distR - distT);
        return d <= 1 ? 0.0 : (d < pen_n ? pen[d] : 0.1 * (double)(int)d + chain_penalty * log((double)(int)d));
    };
    for (int b0 = 0; b0 < n; b0 += 64) {
        const int i = b0 + lane;
        const bool have = i < n;
        const int ic = have ? i : n - 1;
        const uint32_t qpl = sd[2 * ic + 1];
        const int ti = (int)sd[2 * ic], qi = (int)(qpl & 0xFFFFF), li = (int)(qpl >> 20);
        /* phase 1: this wavefront's slice of [0, b0), downwards */
        const int per = (b0 + W - 1) / W, lo = wave * per, hi = lo + per < b0 ? lo + per : b0;
        double bv = -1.0e300; int bj = -1;
        for (int j = hi - 1; j >= lo; j--) {
            const uint32_t qj_ = sd[2 * j + 1];
            const int tj = (int)sd[2 * j], qj = (int)(qj_ & 0xFFFFF), lj = (int)(qj_ >> 20);
            const double dpj = s_dp[j];
            const int distR = qi - (qj + lj - 1), distT = (int)((uint32_t)ti - ((uint32_t)tj + (uint32_t)lj - 1u));
            if (distR > 0 && distT > 0) {
                const double cand = (dpj + reward) - cost(distR, distT);
                if (cand > bv) { bv = cand; bj = j; }
            }
        }
        red_v[wave * 64 + lane] = bv; red_j[wave * 64 + lane] = bj;
        __syncthreads();
        if (wave == 0) {
            bv = red_v[(W - 1) * 64 + lane]; bj = red_j[(W - 1) * 64 + lane];
#pragma unroll
            for (int u = W - 2; u >= 0; u--) { const double v2 = red_v[u * 64 + lane]; const int j2 = red_j[u * 64 + lane]; if (v2 > bv) { bv = v2; bj = j2; } }
            /* phase 2: the block's own seeds, one after the other */
            const int ns = n - b0 < 64 ? n - b0 : 64;
#pragma unroll 4
            for (int s = 0; s < ns; s++) {
                double di = (double)li; int pi = -1;
                if (bj >= 0 && bv > di) { di = bv; pi = bj; }
                /* seed b0 + s is final in lane s: its values for everybody */
                const int lo32 = __builtin_amdgcn_readlane((int)(__double_as_longlong(di) & 0xffffffffll), s), hi32 = __builtin_amdgcn_readlane((int)(__double_as_longlong(di) >> 32), s);
                const double ds = __longlong_as_double(((long long)hi32 << 32) | (unsigned int)lo32);
                const int ts = __builtin_amdgcn_readlane(ti, s), qs = __builtin_amdgcn_readlane(qi, s), ls = __builtin_amdgcn_readlane(li, s);
                if (lane == s) { s_dp[i] = di; prev[i] = pi; }
                if (ds > best) { best = ds; bestIdx = b0 + s; }
                if (lane > s && have) {
                    const int distR = qi - (qs + ls - 1), distT = (int)((uint32_t)ti - ((uint32_t)ts + (uint32_t)ls - 1u));
                    if (distR > 0 && distT > 0) {
                        const double cand = (ds + reward) - cost(distR, distT);
                        if (cand >= bv) { bv = cand; bj = b0 + s; }
                    }
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t len = 0;
        for (int k = bestIdx; k != -1; k = prev[k]) len++;
        uint32_t wpos = len;
        uint32_t *out = chain_idx + w.off;
        for (int k = bestIdx; k != -1; k = prev[k]) out[--wpos] = (uint32_t)k;
        chain_len[w.id] = len;
        score[w.id] = (float)best;
    }
}
/* dynamic LDS of the kernel above for windows of up to `cap` seeds */
static inline size_t lf_chain_big_smem(uint32_t cap, bool prev_lds) { return (size_t)cap * 8 + (size_t)LF_CHAIN_BIG_W * 64 * 12 + (prev_lds ? (size_t)cap * 4 : 0) + 16; }

/* one launch per size class over ALL windows (a block outside its class exits): the one-wavefront kernel up to 512 seeds, the workgroup kernel up to
 * LF_CHAIN_BIG_MAX, the one-wavefront kernel with its state in the HBM workspace above that.  n_largest: the largest window if the caller knows it. */
static inline int lf_chain_n2_launch_classes(int device, hipStream_t s, const lf_chain_win *d_wins, int n_wins, const uint32_t *d_seeds, const double *d_pen, uint32_t pen_n, double reward, double chain_penalty,
                                             double *d_dp, int *d_prev, bool have_ws, uint32_t *d_cidx, uint32_t *d_clen, float *d_cscore, uint32_t n_largest)
{
    if (n_wins <= 0) return LF_OK;
    static const uint32_t SMALL[2] = { 128, 512 };
    uint32_t lo = 0;
    for (int c = 0; c < 2 && lo <= n_largest; c++) {
        hipLaunchKernelGGL(lf_chain_n2_kernel, dim3((unsigned)n_wins), dim3(64), (size_t)SMALL[c] * 22 + 16, s, d_wins, n_wins, d_seeds, d_pen, pen_n, reward, chain_penalty, SMALL[c],
                           d_dp, d_prev, d_cidx, d_clen, d_cscore, lo, SMALL[c]);
        lo = SMALL[c] + 1;
    }
    static bool attr_done[64] = { false };          /* per device (and per translation unit: the kernel is a static function of each) */
    if (!attr_done[device & 63]) {
        HIPCHK(hipFuncSetAttribute((const void *)lf_chain_n2_big_kernel<LF_CHAIN_BIG_W, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf_chain_big_smem(LF_CHAIN_BIG_MAX, false)));
        attr_done[device & 63] = true;
    }
    static const uint32_t BIG[3] = { 2048, LF_CHAIN_LDS_MAX, LF_CHAIN_BIG_MAX };
    for (int c = 0; c < 3 && lo <= n_largest; c++) {
        if (BIG[c] <= LF_CHAIN_LDS_MAX)
            hipLaunchKernelGGL((lf_chain_n2_big_kernel<LF_CHAIN_BIG_W, true>), dim3((unsigned)n_wins), dim3(64 * LF_CHAIN_BIG_W), lf_chain_big_smem(BIG[c], true), s, d_wins, n_wins, d_seeds, d_pen, pen_n,
                               reward, chain_penalty, BIG[c], d_prev, d_cidx, d_clen, d_cscore, lo, BIG[c]);
        else if (have_ws)
            hipLaunchKernelGGL((lf_chain_n2_big_kernel<LF_CHAIN_BIG_W, false>), dim3((unsigned)n_wins), dim3(64 * LF_CHAIN_BIG_W), lf_chain_big_smem(BIG[c], false), s, d_wins, n_wins, d_seeds, d_pen, pen_n,
                               reward, chain_penalty, BIG[c], d_prev, d_cidx, d_clen, d_cscore, lo, BIG[c]);
        lo = BIG[c] + 1;
    }
    if (have_ws && lo <= n_largest)
        hipLaunchKernelGGL(lf_chain_n2_kernel, dim3((unsigned)n_wins), dim3(64), (size_t)22 + 16, s, d_wins, n_wins, d_seeds, d_pen, pen_n, reward, chain_penalty, 1u,
                           d_dp, d_prev, d_cidx, d_clen, d_cscore, lo, 0xFFFFFFFFu);
    return LF_OK;
}

#endif
