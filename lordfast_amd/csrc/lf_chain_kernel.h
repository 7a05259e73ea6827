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

#endif
