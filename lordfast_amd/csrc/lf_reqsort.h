/* lf_reqsort.h -- a request's (candidate window's) seeds in key order, one request per workgroup: shared by lf_vote.hip (the pipeline's requests,
 * built on the device) and lf_chain.hip (the stage API's / drop-in's windows of chain_seeds_clasp).  gfx950 only. */
#ifndef LF_REQSORT_H
#define LF_REQSORT_H
#include "lf_gpu_common.h"

/* a request's seeds in key order -- qPos for dp-n2 (what std::sort(compare_seed) orders by, src/Chain.cpp:244), target
 * start for clasp (qsort(cmp_slmatch_qsort), src/Chain.cpp:94) -- STABLE, i.e. equal keys keep the gathered order (clasp's
 * qsort is glibc's stable merge sort; for dp-n2 the unstable std::sort is replayed afterwards on the requests that have ties).
 * A SEGMENTED sort: requests are independent and a few hundred seeds long, so every request is sorted by its own workgroup in
 * LDS -- a bitonic network over 64-bit words (key << 32 | place in the gathered order): the words are distinct, so the network's
 * result is THE stable order -- instead of one radix sort over (request, key) pairs of the whole chunk through HBM (round 3:
 * hipCUB, 14 launches and four passes over 1.3 M pairs per chunk). ---- */
template <int THREADS>
static __global__ void __launch_bounds__(THREADS)
lf_req_sort_kernel(int n_req, const uint64_t *__restrict__ req_off, const uint32_t *__restrict__ req_n, const uint2 *__restrict__ gathered,
                   uint2 *__restrict__ sorted, uint64_t *__restrict__ skeys_sorted, int key_by_tpos, uint32_t n_lo, uint32_t n_hi)
{
    extern __shared__ uint64_t s_w[];
    const int q = blockIdx.x, t = threadIdx.x;
    if (q >= n_req) return;
    const uint32_t n = req_n[q];
    if (n < n_lo || n > n_hi) return;                       /* another launch's size class */
    const uint64_t off = req_off[q];
    uint32_t N = 1; while (N < n) N <<= 1;
    for (uint32_t i = t; i < N; i += THREADS) {
        uint64_t w = ~0ull;
        if (i < n) { const uint2 sd = gathered[off + i]; w = ((uint64_t)(key_by_tpos ? sd.x : (sd.y & 0xFFFFFu)) << 32) | i; }
        s_w[i] = w;
    }
    auto sync = [&]() { if (THREADS > 64) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } };
    sync();
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < N; i += THREADS) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t a = s_w[i], b = s_w[l];
                    if ((a > b) == ((i & k) == 0)) { s_w[i] = b; s_w[l] = a; }
                }
            }
            sync();
        }
    for (uint32_t i = t; i < n; i += THREADS) {
        const uint2 sd = gathered[off + (uint32_t)s_w[i]];
        sorted[off + i] = sd;
        skeys_sorted[off + i] = key_by_tpos ? (((uint64_t)(uint32_t)q << 32) | sd.x) : (((uint64_t)(uint32_t)q << 20) | (sd.y & 0xFFFFFu));
    }
}

/* ... and the requests above 8192 seeds (a window over a satellite array: every sample of the read hits it): the same network over a
 * scratch array in HBM (2 n words at 2 x the request's offset: the next power of two is below 2 n), one 1024-thread workgroup per
 * request.  Rare and L2-sized; replaces the chunk-wide hipCUB radix sort such a request used to switch the whole chunk to. */
static __global__ void __launch_bounds__(1024)
lf_req_sort_big_kernel(int n_req, const uint64_t *__restrict__ req_off, const uint32_t *__restrict__ req_n, const uint2 *__restrict__ gathered,
                       uint2 *__restrict__ sorted, uint64_t *__restrict__ skeys_sorted, uint64_t *__restrict__ scratch, int key_by_tpos, uint32_t n_lo)
{
    const int q = blockIdx.x, t = threadIdx.x;
    if (q >= n_req) return;
    const uint32_t n = req_n[q];
    if (n < n_lo) return;
    const uint64_t off = req_off[q];
    uint64_t *w = scratch + 2 * off;
    uint32_t N = 1; while (N < n) N <<= 1;
    for (uint32_t i = t; i < N; i += 1024) {
        uint64_t x = ~0ull;
        if (i < n) { const uint2 sd = gathered[off + i]; x = ((uint64_t)(key_by_tpos ? sd.x : (sd.y & 0xFFFFFu)) << 32) | i; }
        w[i] = x;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < N; i += 1024) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t a = w[i], b = w[l];
                    if ((a > b) == ((i & k) == 0)) { w[i] = b; w[l] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = t; i < n; i += 1024) {
        const uint2 sd = gathered[off + (uint32_t)w[i]];
        sorted[off + i] = sd;
        skeys_sorted[off + i] = key_by_tpos ? (((uint64_t)(uint32_t)q << 32) | sd.x) : (((uint64_t)(uint32_t)q << 20) | (sd.y & 0xFFFFFu));
    }
}


/* the launches for requests of up to max_n seeds: one wavefront per request up to 512 seeds (4 KiB of LDS), a 256-thread workgroup up to 8192 (64 KiB), above
 * that (from big_from on) a 1024-thread workgroup over `scratch` (2 x the seeds' total + 32 words, only needed then) */
static inline int lf_req_sort_launch(int device, hipStream_t s, int n_req, const uint64_t *d_req_off, const uint32_t *d_req_n, const uint2 *d_gath, uint2 *d_sorted, uint64_t *d_skeys_sorted,
                                     int key_by_tpos, uint32_t max_n, uint32_t big_from, uint64_t *scratch)
{
    (void)device;
    if (n_req <= 0) return LF_OK;
    const uint32_t lds_hi = big_from - 1u < 8192u ? big_from - 1u : 8192u;
    const uint32_t small_hi = max_n < 512u ? max_n : 512u, hi1 = lds_hi < 512u ? lds_hi : 512u;
    uint32_t cap1 = 64; while (cap1 < small_hi) cap1 <<= 1;
    hipLaunchKernelGGL(lf_req_sort_kernel<64>, dim3((unsigned)n_req), dim3(64), (size_t)cap1 * 8, s, n_req, d_req_off, d_req_n, d_gath, d_sorted, d_skeys_sorted, key_by_tpos, 0u, hi1);
    if (max_n > 512u && lds_hi > 512u) {
        const uint32_t top = max_n < lds_hi ? max_n : lds_hi;
        uint32_t cap2 = 1024; while (cap2 < top) cap2 <<= 1;
        hipLaunchKernelGGL(lf_req_sort_kernel<256>, dim3((unsigned)n_req), dim3(256), (size_t)cap2 * 8, s, n_req, d_req_off, d_req_n, d_gath, d_sorted, d_skeys_sorted, key_by_tpos, 513u, lds_hi);
    }
    if (max_n > lds_hi) {
        if (!scratch) { lf_set_error("lf_req_sort_launch: no scratch for requests above %u seeds", lds_hi); return LF_ERR_NOMEM; }
        hipLaunchKernelGGL(lf_req_sort_big_kernel, dim3((unsigned)n_req), dim3(1024), 0, s, n_req, d_req_off, d_req_n, d_gath, d_sorted, d_skeys_sorted, scratch, key_by_tpos, lds_hi + 1u);
    }
    return LF_OK;
}
#endif
