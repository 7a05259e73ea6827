/* lf_scan.h -- exclusive prefix sum of a batch-sized array in ONE launch (gfx950): decoupled look-back.
 *
 * Every stage of a chunk lays its output out with a prefix sum (hit offsets, request offsets, checkpoint bases, text
 * offsets ...): ~15 per chunk.  Through hipCUB each was two kernels (look-back state initialisation + scan) and a temporary
 * buffer query; with sixteen chunks in flight that was several hundred tiny launches per step competing for the hardware
 * queues.  Here a scan is one kernel: a tile of 2048 elements per 256-thread workgroup; tiles take their number from a
 * counter (so a tile only ever waits for tiles that are already running), publish (flag, epoch, value) in one 64-bit word --
 * flag 1 = the tile's own sum, 2 = the inclusive prefix up to and including the tile -- and look back over their
 * predecessors' words.  The status words need no clearing between launches: a word is valid only if it carries the
 * launch's epoch (14 bits; lfg_scan_ws clears the array when the epoch wraps or the array is new).  Flag, epoch and value
 * travel in ONE word, so the words are read and written with RELAXED agent-scope atomics (they bypass the non-coherent cache
 * levels by themselves): an acquire / release pair per poll would invalidate / write back the L1 and L2 of a CU that other
 * chunks' kernels are running on -- measured: 8 ms per step with sixteen chunks in flight.
 * Values are sums below 2^48.  Input is a functor of the element index (counts of another type, lengths derived from
 * records, ...), output u64. */
#ifndef LF_SCAN_H
#define LF_SCAN_H
#include "lf_gpu_common.h"

#define LF_SCAN_TILE 2048
struct lf_scan_ws_t { unsigned long long *status; unsigned int *counter; unsigned int epoch; };
/* lf_mem.hip: the lane's workspace number `which` (0..7) for a scan of n elements on `stream` (cleared when new / on epoch wrap) */
extern "C" int lfg_scan_ws(int device, int which, size_t n, void *stream, lf_scan_ws_t *ws);

template <class F>
__global__ void __launch_bounds__(256)
lf_scan_excl_kernel(F f, uint64_t *__restrict__ out, uint32_t n, unsigned long long *__restrict__ status, unsigned int *__restrict__ counter, uint32_t epoch)
{
    __shared__ uint32_t s_tile; __shared__ uint64_t s_wsum[4]; __shared__ uint64_t s_prefix;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n_tiles = gridDim.x;
    if (tid == 0) s_tile = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tid == 0 && tile == n_tiles - 1) atomicExch(counter, 0u);          /* every tile has its number: ready for the next launch */
    const uint64_t i0 = (uint64_t)tile * LF_SCAN_TILE + (uint64_t)tid * 8;
    uint64_t v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint64_t idx = i0 + k; v[k] = idx < n ? (uint64_t)f((uint32_t)idx) : 0ull; sum += v[k]; }
    uint64_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_wsum[w] = inc;
    __syncthreads();
    uint64_t wbase = 0;
    for (int j = 0; j < w; j++) wbase += s_wsum[j];
    const uint64_t total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    if (w == 0) {
        /* wavefront 0 publishes the tile's own sum at once, then looks back over 64 predecessors per trip: the nearest one that
         * already knows its inclusive prefix ends the walk, the tiles between it and us contribute their own sums; a word
         * that is not valid yet (wrong epoch / flag 0) is simply read again */
        const unsigned long long tag = ((unsigned long long)(epoch & 0x3fffu)) << 48, vmask = (1ull << 48) - 1;
        uint64_t prefix = 0;
        if (tile == 0) { if (lane == 0) __hip_atomic_store(&status[0], (2ull << 62) | tag | (total & vmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            if (lane == 0) __hip_atomic_store(&status[tile], (1ull << 62) | tag | (total & vmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t hi = (int64_t)tile - 1;                       /* nearest predecessor of the current window */
            for (;;) {
                const int64_t j = hi - lane;
                unsigned long long x = 0;
                if (j >= 0) x = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool valid = j < 0 || (((x >> 48) & 0x3fffu) == (epoch & 0x3fffu) && (x >> 62) != 0);
                const bool incl = j >= 0 && valid && (x >> 62) == 2;
                const uint64_t im = lf_ballot(incl), vm = lf_ballot(valid);
                /* lanes 0 .. first are usable when all of them are valid (first = nearest inclusive one, or 63 if none) */
                const int first = im ? __ffsll((long long)im) - 1 : 63;
                const uint64_t need = first == 63 ? ~0ull : ((2ull << first) - 1);
                if ((vm & need) != need) { __builtin_amdgcn_s_sleep(2); continue; }
                uint64_t part = (j >= 0 && lane <= first) ? (x & vmask) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                prefix += part;
                if (im || hi - 63 <= 0) break;                     /* reached an inclusive prefix, or tile 0's side of the array */
                hi -= 64;
            }
            if (lane == 0) __hip_atomic_store(&status[tile], (2ull << 62) | tag | ((prefix + total) & vmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();
    uint64_t run = s_prefix + wbase + inc - sum;
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint64_t idx = i0 + k; if (idx < n) { out[idx] = run; run += v[k]; } }
}

/* out[i] = sum of f(0 .. i-1), i < n.  One kernel on `stream`.  `which`: the caller's workspace number (a lane runs its stages
 * one after the other, but two modules never share a workspace). */
template <class F>
static inline int lf_scan_excl(int device, int which, hipStream_t stream, F f, uint64_t *out, size_t n)
{
    if (n == 0) return LF_OK;
    if (n >= (1ull << 32)) { lf_set_error("lf_scan_excl: %zu elements", n); return LF_ERR_ARG; }
    lf_scan_ws_t ws;
    const int rc = lfg_scan_ws(device, which, n, (void *)stream, &ws);
    if (rc != LF_OK) return rc;
    hipLaunchKernelGGL((lf_scan_excl_kernel<F>), dim3((unsigned)((n + LF_SCAN_TILE - 1) / LF_SCAN_TILE)), dim3(256), 0, stream, f, out, (uint32_t)n, ws.status, ws.counter, ws.epoch);
    return LF_OK;
}
struct lf_scan_u32 { const uint32_t *p; __device__ __forceinline__ uint64_t operator()(uint32_t i) const { return p[i]; } };
struct lf_scan_u64 { const uint64_t *p; __device__ __forceinline__ uint64_t operator()(uint32_t i) const { return p[i]; } };
#endif
