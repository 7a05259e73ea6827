/*
 * lf_rsweep.hip -- the forward pass of every edlib problem with at most 64 query blocks (n <= 4096, below edlib's
 * traceback switch): ONE kernel for all sizes (gfx950).
 *
 * Round 2 had six register-resident "lane" kernels (one lane per problem, NB = 1 .. 8 blocks walked as the lane's own
 * anti-diagonal) and three "sweep" kernels (G = 16 / 32 / 64 lanes per problem).  The lane kernels issue ~60 instructions per
 * 64-row block step but reached half the rate of the sweeps: 100 - 170 registers (3 - 4 wavefronts per SIMD) and a prologue
 * of 64 dependent byte loads + ~500 ALU operations per block to build the query's bit planes -- as long as the DP itself
 * for the short problems that make up most of the batch.  Here:
 *
 *   - a problem of nb blocks takes nb CONSECUTIVE LANES of a wavefront (one block per lane, the carry between blocks by DPP
 *     wave_shr:1); a wavefront holds floor(64 / nb) problems of the SAME nb, sorted by target length.  nb is a run-time
 *     value: 1, 2, 3, ... 64 -- no power-of-two size classes, no idle lanes beyond 64 mod nb;
 *   - the query's bit planes are not built per problem: lf_pack_planes_kernel turns the read batch into three bit arrays
 *     (code low bit, code high bit, "is one of ACGT") ONCE per chunk, coalesced; a block's planes are three unaligned 64-bit
 *     windows of those arrays (bit-reversed for queries walked backwards, complemented for reverse-strand chains): 6 loads
 *     in flight together + ~25 operations;
 *   - the four match masks of a lane's block (one per target code) live in a 2 KiB LDS table; the lane's target symbols come
 *     16 at a time straight from the 2-bit reference into a register (one unaligned 8-byte load per 16 steps, any direction
 *     / strand), so the inner loop has no ring, no staging, no barrier;
 *   - the step loop is unrolled over 16 steps (two checkpoint rows): symbol extraction, shifts and checkpoint addresses are
 *     immediates; ~45 instructions per block step (sweep kernels of round 2: ~80), ~50 registers: 8 wavefronts per SIMD.
 *
 * Output: distance / end column, and -- for lf_edlib_tb_kernel, one lane per path -- the wave's planes and one checkpoint row
 * per 32 steps: (Pv, Mv) of every lane in front of the row, then the 32 two-bit carries every lane received during it.
 */
#include "lf_edlib_common.h"
#include "lf_rsweep.h"
#include "lf_tb_core.h"
#include <type_traits>

/* ---- read batch -> bit planes: bit i of word (i >> 6) describes base i.  A wavefront transposes 64 x 64 bases through ballots. ---- */
__global__ void __launch_bounds__(256)
lf_pack_planes_kernel(const unsigned char *__restrict__ src, uint64_t n, uint64_t *__restrict__ lo, uint64_t *__restrict__ hi, uint64_t *__restrict__ valid, uint64_t n_words,
                      unsigned long long *__restrict__ lower_flag)
{
    const int lane = threadIdx.x & 63;
    const uint64_t w0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;                 /* first of this wavefront's 64 words */
    uint64_t mlo = 0, mhi = 0, mv = 0;
    bool lower = false;
    /* eight rows of 64 bases are requested together (their latency is paid once per eight), then balloted */
    for (int i0 = 0; i0 < 64; i0 += 8) {
        unsigned char ch[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint64_t p = (w0 + (uint64_t)(i0 + u)) * 64 + (uint64_t)lane; ch[u] = p < n ? src[p] : (unsigned char)0; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            bool ok; const uint32_t cd = lf_code_upper(ch[u], ok);
            { bool okl; (void)lf_code_upper((uint32_t)ch[u] & 0xDFu, okl); lower = lower || (okl && !ok); }
            const uint64_t bl = lf_ballot(ok && (cd & 1u)), bh = lf_ballot(ok && (cd & 2u)), bv = lf_ballot(ok);
            if (lane == i0 + u) { mlo = bl; mhi = bh; mv = bv; }
        }
    }
    const uint64_t w = w0 + (uint64_t)lane;
    if (w < n_words) { lo[w] = mlo; hi[w] = mhi; valid[w] = mv; }
    if (lower_flag != nullptr && lf_any(lower) && lane == 0) atomicOr(lower_flag, 1ull);
}

/* byte targets of the stage API -> the reference's 2-bit layout (four symbols per byte, first symbol in the top bits,
 * lib/bwa/bntseq.c:_set_pac); anything but upper-case ACGT becomes code 0 -- problems with such targets never read this
 * array (the host flags them: generic kernel, exact byte compare) */
__global__ void lf_pack_pac_kernel(const unsigned char *__restrict__ src, uint64_t n, uint8_t *__restrict__ pac)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b * 4 >= n + 64) return;
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) {
        const uint64_t p = b * 4 + (uint64_t)k;
        bool ok = true; uint32_t cd = 0;
        if (p < n) cd = lf_code_upper(src[p], ok);
        v |= (ok ? cd : 0u) << (6 - 2 * k);
    }
    pac[b] = (uint8_t)v;
}

template <bool TRACK>
__device__ __forceinline__ void lf_rsweep_body(const lf_rsw_args &A)
{
    __shared__ uint64_t s_peq[4 * 64];
    __shared__ int s_part[64];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= A.n_waves) return;
    /* the segments are sorted by blocks per problem and target length, ascending: handed out from the END, the long sweeps start first and the
     * short ones fill in behind them (the launch ends with the last short wave, not with a long one that started last) */
    const lf_rwave W = A.waves[A.wave0 + (A.rev ? A.n_waves - 1 - (int)blockIdx.x : (int)blockIdx.x)];
    const int G = (int)W.G;
    const int g = lane / G, gl = lane - g * G;
    const bool live = g < (int)W.count;
    const lf_aln_prob pr = A.probs[W.first + (live ? (uint32_t)g : 0u)];
    const uint32_t n = pr.n, m = live ? pr.m : 0u;
    const uint32_t nbk = (n + 63) >> 6;
    const bool mine = live && (uint32_t)gl < nbk;       /* G >= nbk for every problem of the wave (G = nbk: problems of one wave share it) */
    const int dq = (pr.flags & LF_F_QREV) ? -1 : 1, dt = (pr.flags & LF_F_TREV) ? -1 : 1;
    const bool cq = (pr.flags & LF_F_QCOMP) != 0, ct = (pr.flags & LF_F_TCOMP) != 0;

    /* ---- this lane's block: bit planes from the packed read batch ---- */
    uint64_t lo, hi, valid;
    {
        const int64_t r0 = (int64_t)gl * 64;            /* first row of the block */
        const int64_t p0 = dq > 0 ? pr.qstart + r0 : pr.qstart - r0 - 63;
        lo = lf_bits64(A.qlo, p0, A.q_words); hi = lf_bits64(A.qhi, p0, A.q_words); valid = lf_bits64(A.qvalid, p0, A.q_words);
        if (dq < 0) { lo = lf_brev64(lo); hi = lf_brev64(hi); valid = lf_brev64(valid); }
        if (cq) { lo = ~lo; hi = ~hi; }
        const int64_t rows = (int64_t)n - r0;
        const uint64_t rmask = !mine || rows <= 0 ? 0ull : rows >= 64 ? ~0ull : ((1ull << rows) - 1);
        valid &= rmask; lo &= valid; hi &= valid;
    }
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        const uint64_t slo = 0ull - (uint64_t)(c & 1u), shi = 0ull - (uint64_t)(c >> 1);
        s_peq[c * 64 + lane] = ~((lo ^ slo) | (hi ^ shi)) & valid;
    }
    const bool ck_on = live && pr.task == LF_TASK_PATH;
    const bool any_ck = lf_any(ck_on);
    lf_hist_t *wbase = A.ckpt + W.hist_base;
    if (any_ck) { uint64_t *pl = reinterpret_cast<uint64_t *>(wbase); pl[lane] = lo; pl[64 + lane] = hi; pl[128 + lane] = valid; }
    lf_hist_t *ck = wbase + LF_PLANE_ENTRIES;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();

    const uint32_t lastb = (n - 1) >> 6; const int lastbit = (int)((n - 1) & 63);
    const int steps = live ? (int)m + G - 1 : 0;
    const int steps_max = lf_wave_max_i32(steps);
    uint64_t Pv = ~0ull, Mv = 0;
    uint32_t hout = LF_HIN_PLUS1, cw = 0;
    int score = (int)n, best = (n & 63) ? (int)n : 0x7fffffff, best_c = 0;
    const bool first = gl == 0, is_last = (uint32_t)gl == lastb;
    const uint64_t *peq_l = s_peq + lane;
    /* 16 steps.  FAST: every block of the wavefront is inside its target for all 16 of them (most groups of 16: a lane is idle
     * only while it waits for its first column and after its last) -- no per-step range test, no exec-mask round trip:
     * 37 instead of 45 instructions per block step, and the kernel runs at the SIMDs' issue rate. */
    auto steps16 = [&](auto fast_tag, const int s0, const uint32_t V) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int64_t p = (int64_t)s0 - gl;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t from_left = lf_wave_shr1(hout);
            cw |= from_left << (2 * k);
            const uint32_t col0 = (uint32_t)((int)p + k);            /* column - 1; wraps for lanes that have not started */
            if (FAST || (mine && col0 < m)) {
                const uint32_t sym = (V >> (2 * k)) & 3u;
                const uint64_t Eq = peq_l[sym * 64];
                const uint32_t hin = first ? LF_HIN_PLUS1 : from_left;
                uint64_t ph, mh;
                hout = lf_myers_step(Pv, Mv, Eq, hin, ph, mh);
                if (TRACK) {
                    score += is_last ? lf_delta_at(ph, mh, lastbit) : 0;
                    const bool upd = is_last && score < best; best = upd ? score : best; best_c = upd ? (int)col0 + 1 : best_c;
                }
            }
            if (k == 15) {
                /* the 16 two-bit carries the lane received: one half of the row entry's carry word */
                if (any_ck) reinterpret_cast<uint32_t *>(ck + (size_t)(s0 >> 5) * LF_RROW + 64)[2 * lane + ((s0 >> 4) & 1)] = cw;
                cw = 0;
            }
        }
    };
    /* the lane's 16 target symbols of a group (stream position s - gl) are requested ONE GROUP AHEAD: the load right in front of its use was a stall per group */
    uint32_t Vn = lf_pac16(A.pac, pr.tstart + (int64_t)dt * (0 - (int64_t)gl), dt, ct, A.pac_syms);
    for (int s0 = 0; s0 < steps_max; s0 += 16) {
        const int64_t p = (int64_t)s0 - gl;
        const uint32_t V = Vn;
        Vn = lf_pac16(A.pac, pr.tstart + (int64_t)dt * (p + 16), dt, ct, A.pac_syms);
        /* a row starts: the state in front of it */
        if (any_ck && (s0 & 31) == 0) { lf_hist_t e; e.pv = Pv; e.ph = Mv; ck[(size_t)(s0 >> 5) * LF_RROW + lane] = e; }
        const bool partial = mine && (p < 0 || p + 15 >= (int64_t)m);       /* (lanes without a block compute on dead registers) */
        if (!lf_any(partial)) steps16(std::true_type{}, s0, V);
        else steps16(std::false_type{}, s0, V);
    }
    /* NW: D[n][m] = m + the vertical deltas of the last column, summed over the problem's lanes */
    {
        const uint32_t rows = (uint32_t)gl == lastb ? (uint32_t)lastbit + 1 : 64;
        const uint64_t msk = rows >= 64 ? ~0ull : ((1ull << rows) - 1);
        s_part[lane] = mine ? __popcll(Pv & msk) - __popcll(Mv & msk) : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    /* SHW (lib/edlib/edlib.cpp:583-618): smallest prefix distance, smallest column on ties -- followed by the lane of the last
     * block; every lane fetches its group's (all lanes active: the source of a shuffle must be) */
    int b_ed = 0, b_c = 0;
    if (TRACK) { const int src = (lane - gl + (int)lastb) & 63; b_ed = __shfl(best, src); b_c = __shfl(best_c, src); }
    if (live && gl == 0) {
        int ed = (int)m;
        for (uint32_t b = 0; b < nbk; b++) ed += s_part[lane + (int)b];
        int tl = (int)m;
        if (TRACK && pr.mode != 0) { ed = b_ed; tl = b_c; }
        A.out_ed[pr.id] = ed; A.out_end[pr.id] = tl - 1;
    }
}
template <bool TRACK>
__global__ void __launch_bounds__(64)
lf_edlib_rsweep_kernel(lf_rsw_args A) { lf_rsweep_body<TRACK>(A); }

void lf_rsweep_pack_planes(hipStream_t s, const unsigned char *d_src, uint64_t n_bytes, uint64_t *d_planes, uint64_t n_words, unsigned long long *lower_flag)
{
    if (!n_words) return;
    hipLaunchKernelGGL(lf_pack_planes_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, d_src, n_bytes, d_planes, d_planes + n_words, d_planes + 2 * n_words, n_words, lower_flag);
}
void lf_rsweep_pack_pac(hipStream_t s, const unsigned char *d_src, uint64_t n_bytes, uint8_t *d_pac)
{
    const uint64_t nb = (n_bytes + 64 + 3) / 4;
    hipLaunchKernelGGL(lf_pack_pac_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, d_src, n_bytes, d_pac);
}
void lf_rsweep_launch(hipStream_t s, bool track, lf_rsw_args A)
{
    if (A.n_waves <= 0) return;
    if (track) hipLaunchKernelGGL((lf_edlib_rsweep_kernel<true>), dim3((unsigned)A.n_waves), dim3(64), 0, s, A);
    else hipLaunchKernelGGL((lf_edlib_rsweep_kernel<false>), dim3((unsigned)A.n_waves), dim3(64), 0, s, A);
}
