/*
 * lf_chain.hip -- dp-n2 anchor chaining (gfx950).  Reference: chain_seeds_n2 (src/Chain.cpp:232-310).
 *
 * One wavefront per window.  Seeds arrive already in the reference's std::sort order (the introsort
 * permutation is reproduced on the host, SURVEY 7 hard part 1).  The DP row i is sequential; its inner
 * scan over j < i is spread over the 64 lanes, each keeping its best (value, largest j), followed by a
 * wavefront max-reduction with the same tie rule -- the reference visits j downwards and updates on strict
 * '>', so the winner is the LARGEST j attaining the maximum, and only if it beats dp[i] = len_i.
 * FP64 throughout, evaluation order (dp[j] + reward) - pen, no contraction (-ffp-contract=off).
 * pen[d] = 0.1*d + chainPenalty*log(d) comes from a host-built table (glibc log, the same libm the
 * reference uses), so no device transcendental is involved.
 * Seeds + dp + prev live in LDS (<= LF_CHAIN_LDS_MAX seeds) or in an HBM workspace for huge windows.
 */
#include "lf_gpu_common.h"
#include <vector>
#include <algorithm>
#include <math.h>

#include "lf_chain_kernel.h"

extern "C" int lfg_chain_n2(int device, const lf_params_t *p, int n_windows, const Seed_t *sorted_seeds,
                            const uint64_t *off, uint32_t *chain_idx, uint32_t *chain_len, float *score, float *ms)
{
    if (ms) *ms = 0;
    if (n_windows == 0) return LF_OK;
    if (lfg_device_count() <= device) { lf_set_error("no gfx950 device %d visible (no CPU path)", device); return LF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(device));
    const uint64_t total = off[n_windows];
    std::vector<lf_chain_win> W((size_t)n_windows);
    size_t ws = 0;
    uint64_t max_d = 64;
    const uint32_t *raw = (const uint32_t *)sorted_seeds;
    for (int i = 0; i < n_windows; i++) {
        W[i].off = off[i]; W[i].n = (uint32_t)(off[i + 1] - off[i]); W[i].id = (uint32_t)i; W[i].ws_off = ws;
        if (W[i].n > LF_CHAIN_LDS_MAX) ws += W[i].n;
        if (W[i].n) {
            uint32_t tmin = 0xFFFFFFFFu, tmax = 0, qmax = 0;
            for (uint64_t k = off[i]; k < off[i + 1]; k++) {
                const uint32_t t = raw[2 * k], qv = raw[2 * k + 1] & 0xFFFFF;
                if (t < tmin) tmin = t;
                if (t > tmax) tmax = t;
                if (qv > qmax) qmax = qv;
            }
            const uint64_t d = (uint64_t)(tmax - tmin) + qmax + 8192;
            if (d > max_d) max_d = d;
        }
    }
    std::sort(W.begin(), W.end(), [](const lf_chain_win &a, const lf_chain_win &b) { return a.n < b.n || (a.n == b.n && a.id < b.id); });
    if (max_d > (64u << 20)) max_d = 64u << 20;      /* beyond the table the kernel evaluates the formula itself */
    /* penalty table, evaluated exactly as score_penalty does (src/Chain.cpp:224) with the host libm */
    std::vector<double> pen((size_t)max_d);
    for (uint64_t d = 0; d < max_d; d++) pen[d] = d <= 1 ? 0.0 : 0.1 * (double)(int)d + p->chain_penalty * log((double)(int)d);
    const double reward = p->chain_reward * (double)p->min_anchor_len;     /* score_reward, src/Chain.cpp:211-215 */

#define CSLOT(k, bytes) lfg_dev_slot(device, LF_DS_CHAIN0 + (k), (bytes))
    void *d_w = CSLOT(0, W.size() * sizeof(lf_chain_win)), *d_seeds = CSLOT(1, total * 8 + 16), *d_pen = CSLOT(2, pen.size() * 8);
    void *d_dp = CSLOT(3, ws * 8 + 16), *d_prev = CSLOT(4, ws * 4 + 16), *d_idx = CSLOT(5, total * 4 + 16);
    void *d_len = CSLOT(6, (size_t)n_windows * 4), *d_sc = CSLOT(7, (size_t)n_windows * 4);
#undef CSLOT
    if (!d_w || !d_seeds || !d_pen || !d_dp || !d_prev || !d_idx || !d_len || !d_sc) return LF_ERR_NOMEM;
    hipStream_t s = (hipStream_t)lfg_lane_stream(device, 15);
    if (!s) return LF_ERR_HIP;
    hipEvent_t e0 = (hipEvent_t)lfg_lane_event(device, 30), e1 = (hipEvent_t)lfg_lane_event(device, 31);
    if (!e0 || !e1) return LF_ERR_HIP;
    HIPCHK(hipMemcpyAsync(d_w, W.data(), W.size() * sizeof(lf_chain_win), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_seeds, sorted_seeds, total * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_pen, pen.data(), pen.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(e0, s));
    /* one launch per size class (lf_chain_kernel.h) */
    {
        uint32_t n_largest = 0; bool ws = false;
        for (const lf_chain_win &x : W) { if (x.n > n_largest) n_largest = x.n; if (x.n > LF_CHAIN_LDS_MAX) ws = true; }
        const int lrc = lf_chain_n2_launch_classes(device, s, (const lf_chain_win *)d_w, (int)W.size(), (const uint32_t *)d_seeds, (const double *)d_pen, (uint32_t)pen.size(), reward, p->chain_penalty,
                                                   (double *)d_dp, (int *)d_prev, ws, (uint32_t *)d_idx, (uint32_t *)d_len, (float *)d_sc, n_largest);
        if (lrc != LF_OK) return lrc;
    }
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipMemcpyAsync(chain_idx, d_idx, total * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(chain_len, d_len, (size_t)n_windows * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(score, d_sc, (size_t)n_windows * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (ms) HIPCHK(hipEventElapsedTime(ms, e0, e1));
    return LF_OK;
}

/* ---------------------------------------------------------------------------------------------------------------
 * clasp chaining for host-supplied windows (stage API / drop-in chain_seeds_clasp).  Seeds arrive in the caller's
 * order; the stable sort by target start that the reference's qsort performs (src/Chain.cpp:94) is one device radix
 * sort on (window, tPos).  Kernel: lf_clasp_kernel.h.  chain_out[off[w] .. off[w] + chain_len[w]) receives the chain.
 * ------------------------------------------------------------------------------------------------------------- */
#include "lf_reqsort.h"
#include "lf_clasp_kernel.h"

static __global__ void lf_clasp_gather_kernel(int n_windows, const uint64_t *__restrict__ off, const uint32_t *__restrict__ chain_idx,
                                              const uint32_t *__restrict__ chain_len, const uint2 *__restrict__ sorted, uint2 *__restrict__ chain_out)
{
    const int w = blockIdx.x;
    if (w >= n_windows) return;
    const uint64_t o = off[w];
    for (uint32_t k = threadIdx.x; k < chain_len[w]; k += blockDim.x) chain_out[o + k] = sorted[o + chain_idx[o + k]];
}

extern "C" int lfg_chain_clasp(int device, int n_windows, const Seed_t *seeds, const uint64_t *off,
                               Seed_t *chain_out, uint32_t *chain_len, float *score, float *ms)
{
    if (ms) *ms = 0;
    if (n_windows == 0) return LF_OK;
    if (lfg_device_count() <= device) { lf_set_error("no gfx950 device %d visible (no CPU path)", device); return LF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(device));
    const uint64_t total = off[n_windows];
    if (total >= (1ull << 31)) { lf_set_error("lfg_chain_clasp: too many seeds"); return LF_ERR_ARG; }
    std::vector<lf_chain_win> W((size_t)n_windows);
    uint64_t ws = 0;
    const uint32_t *raw = (const uint32_t *)seeds;
    for (int i = 0; i < n_windows; i++) {
        W[i].off = off[i]; W[i].n = (uint32_t)(off[i + 1] - off[i]); W[i].id = (uint32_t)i; W[i].ws_off = ws;
        if (W[i].n > LF_CLASP_LDS_MAX) ws += W[i].n;
        uint32_t tmin = 0xFFFFFFFFu, tmax = 0;
        for (uint64_t k = off[i]; k < off[i + 1]; k++) { const uint32_t t = raw[2 * k]; if (t < tmin) tmin = t; if (t > tmax) tmax = t; }
        /* clasp itself keeps positions in `int` (lib/clasp/slchain.c:65-72); the sort keys here need span + qPos < 2^29 */
        if (W[i].n && (tmax >= 0x7FFFF000u || tmax - tmin >= (1u << 28))) {
            lf_set_error("lfg_chain_clasp: window %d spans [%u, %u]: positions must be < 2^31 and a window narrower than 2^28", i, tmin, tmax);
            return LF_ERR_ARG;
        }
    }
#define CSLOT(k, bytes) lfg_dev_slot(device, LF_DS_CHAIN0 + (k), (bytes))
    /* the windows' fragments in target order, stable (clasp's qsort is glibc's merge sort): the segmented sort of the pipeline (lf_reqsort.h) */
    uint32_t max_n = 0;
    std::vector<uint32_t> wn((size_t)n_windows + 1);
    for (int i = 0; i < n_windows; i++) { wn[(size_t)i] = W[(size_t)i].n; if (W[(size_t)i].n > max_n) max_n = W[(size_t)i].n; }
    const size_t tb = max_n > 8192u ? 2 * (size_t)total * 8 + 256 : 0;
    void *d_w = CSLOT(0, W.size() * sizeof(lf_chain_win)), *d_seeds = CSLOT(1, total * 8 + 16), *d_sorted = CSLOT(2, total * 8 + 16);
    void *d_ws = CSLOT(3, ws * LF_CLASP_BYTES_PER_FRAG + 16), *d_keys = CSLOT(4, total * 16 + 16), *d_idx = CSLOT(5, total * 4 + 16);
    /* the chain stage owns 8 slots: the three per-window arrays share one, sort scratch and output another */
    const size_t wpad = (((size_t)n_windows + 1) * 4 + 255) & ~(size_t)255, tpad = (tb + 511) & ~(size_t)255;
    char *d_small = (char *)CSLOT(6, 3 * wpad + ((size_t)n_windows + 1) * 8), *d_big = (char *)CSLOT(7, tpad + total * 8 + 16);
#undef CSLOT
    if (!d_w || !d_seeds || !d_sorted || !d_ws || !d_keys || !d_idx || !d_small || !d_big) return LF_ERR_NOMEM;
    void *d_len = d_small, *d_sc = d_small + wpad, *d_wn = d_small + 2 * wpad, *d_off = d_small + 3 * wpad, *d_tmp = d_big, *d_out = d_big + tpad;
    hipStream_t s = (hipStream_t)lfg_lane_stream(device, 15);
    if (!s) return LF_ERR_HIP;
    hipEvent_t e0 = (hipEvent_t)lfg_lane_event(device, 30), e1 = (hipEvent_t)lfg_lane_event(device, 31);
    if (!e0 || !e1) return LF_ERR_HIP;
    HIPCHK(hipMemcpyAsync(d_w, W.data(), W.size() * sizeof(lf_chain_win), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_seeds, seeds, total * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_off, off, ((size_t)n_windows + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(e0, s));
    if (total) {
        HIPCHK(hipMemcpyAsync(d_wn, wn.data(), (size_t)n_windows * 4, hipMemcpyHostToDevice, s));
        const int src = lf_req_sort_launch(device, s, n_windows, (const uint64_t *)d_off, (const uint32_t *)d_wn, (const uint2 *)d_seeds, (uint2 *)d_sorted, (uint64_t *)d_keys, 1, max_n, 8193u,
                                           tb ? (uint64_t *)d_tmp : nullptr);
        if (src != LF_OK) return src;
        HIPCHK(hipStreamSynchronize(s));          /* (wn lives on this function's stack frame) */
    }
    static const uint32_t CCAPS[5] = { 128, 256, 512, LF_CLASP_LDS_MAX, 0 };
    uint32_t lo = 0;
    for (int c = 0; c < 5; c++) {
        const uint32_t hi = CCAPS[c] ? CCAPS[c] : 0xFFFFFFFFu;
        if (c == 4 && ws == 0) break;
        const size_t smem = CCAPS[c] ? (size_t)CCAPS[c] * LF_CLASP_LDS_BYTES_PER_FRAG : 16;
        if (CCAPS[c]) hipLaunchKernelGGL(lf_clasp_kernel<true>, dim3((unsigned)n_windows), dim3(64), smem, s, (const lf_chain_win *)d_w, n_windows,
                           (const uint32_t *)d_sorted, (const uint32_t *)nullptr, CCAPS[c], (unsigned char *)d_ws,
                           (uint32_t *)d_idx, (uint32_t *)d_len, (float *)d_sc, lo, hi);
        else hipLaunchKernelGGL(lf_clasp_kernel<false>, dim3((unsigned)n_windows), dim3(64), smem, s, (const lf_chain_win *)d_w, n_windows,
                           (const uint32_t *)d_sorted, (const uint32_t *)nullptr, CCAPS[c], (unsigned char *)d_ws,
                           (uint32_t *)d_idx, (uint32_t *)d_len, (float *)d_sc, lo, hi);
        lo = hi + 1;
    }
    hipLaunchKernelGGL(lf_clasp_gather_kernel, dim3((unsigned)n_windows), dim3(64), 0, s, n_windows, (const uint64_t *)d_off, (const uint32_t *)d_idx,
                       (const uint32_t *)d_len, (const uint2 *)d_sorted, (uint2 *)d_out);
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipMemcpyAsync(chain_out, d_out, total * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(chain_len, d_len, (size_t)n_windows * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(score, d_sc, (size_t)n_windows * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (ms) HIPCHK(hipEventElapsedTime(ms, e0, e1));
    return LF_OK;
}
