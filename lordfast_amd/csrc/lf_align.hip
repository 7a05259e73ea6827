/*
 * lf_align.hip -- extension kernels (gfx950): edlib-equivalent edit-distance alignment and ksw_extend2.
 *
 * Reference replaced: edlibAlign (lib/edlib/edlib.cpp:101-221) with config {k=-1, NW|SHW, PATH}, and
 * ksw_extend2 (lib/bwa/ksw.c:380-478).  Integer only; results bit-identical.
 *
 * edlib kernels: Myers/Hyyro bit-vector DP, 64 query rows per 64-bit word.
 *   batching       descriptors -> 32-bit class keys -> counting sort (LDS histograms) -> (mode, nb) segments -> checkpoint bases
 *                  (one scan) -> problem array + wave table, all on the device (lf_desc_*_kernel below).
 *   forward pass   lf_rsweep.hip: nb CONSECUTIVE LANES per problem (nb = ceil(n / 64) = 1 .. 64, a run-time value), query bit
 *                  planes precomputed once per chunk, one checkpoint row per 16 sweep steps.  Leaves the distance / end
 *                  column, the rows and the planes in HBM.  (Leaves with 4096 < n <= 32768: lf_edlib_sweep_kernel<64, 4 / 8>
 *                  below, one wavefront per problem; targets with bytes outside ACGT: lf_edlib_generic_kernel.)
 *   traceback      needs two bits per cell -- Pv (vertical +1 => Up) and Ph (horizontal +1 => Left), else Diagonal (match
 *                  iff bytes equal), which is edlib's move priority (lib/edlib/edlib.cpp:950,984,1015) -- and gets them by
 *                  RECOMPUTING 16-column tiles of the ONE block the path is in from a checkpoint row and the stored carries:
 *                  lf_edlib_tb_kernel, one lane per path.  No history stream through HBM.
 *   Hirschberg     edlib's recursion for problems over its 1 MiB traceback switch: lf_hirsch.hip, breadth-first (level l of
 *                  all problems is one launch, 1 / 4 / 8 wavefronts per half of a node); its leaves are ordinary problems of
 *                  the kernels above.
 * ksw_extend2: lf_ksw_mw_kernel (a row's band over the four wavefronts of a workgroup, H / E in LDS rings) for bands of at most 255
 *                  columns, lf_ksw_kernel (one wavefront, 64 columns at a time) for wider ones.
 * No banding: the Ukkonen band of the reference only removes cells that cannot be on an optimal path, and lanes that skip
 * out-of-band blocks save no time, because their neighbours in the wavefront's lock-step sweep do not.
 */
#include "lf_gpu_common.h"
#include "lf_edlib_common.h"
#include "lf_hirsch.h"
#include <atomic>
void lf_htrial_learn(const uint32_t hist[2][16]);
void lf_htrial_pick(uint32_t trial16[2]);
#include "lf_rsweep.h"
#include "lf_tb_core.h"
#include "lf_scan.h"
#include <stddef.h>
#include <algorithm>
#include <type_traits>
#include <string.h>
#include <vector>
#include <chrono>
#include <numeric>
#include <limits.h>

/* ------------------------------------------------------------------------------------------------
 * traceback: ONE LANE PER PATH, for every problem of the one-block-per-lane forward kernel (lf_rsweep.hip: n <= 4096, below
 * edlib's Hirschberg switch -- which includes the leaves lf_hirsch.hip cuts the larger problems into).
 *
 * From (n, tl) the path is followed tile by tile: a tile = the 16 sweep steps of one checkpoint row, of the ONE block the
 * cell is in.  The lane restores that block's state in front of the tile (one 16-byte load), takes the 16 carries the block
 * received (one 4-byte load) and the tile's 16 target symbols (one 8-byte load) -- all three in flight together, one round
 * trip per tile -- and replays the tile in two halves of 8 columns, right half first: (Pv, Ph) of a half's columns go to the
 * lane's own LDS slots, the path is walked through the half one move per trip, then the left half is replayed from the same
 * checkpoint.  The kernel waits for memory most of its time, so a checkpoint row per 16 steps (half the round trips, half the
 * checkpoint traffic of the forward pass) is worth the 8 block steps that are replayed twice.  Same cells, same
 * Up -> Left -> Diagonal priority (lib/edlib/edlib.cpp:950,984,1015), same ops.
 * ---------------------------------------------------------------------------------------------- */
template <int HK>
__global__ void __launch_bounds__(64)
lf_edlib_tb_kernel(const lf_aln_prob *__restrict__ probs, int n_probs, lf_seqs S, int64_t pac_syms, const lf_hist_t *__restrict__ ckpt, uint8_t *__restrict__ ops,
                   const int32_t *__restrict__ out_end, uint32_t *__restrict__ out_len, int rev)
{
    __shared__ ulonglong2 s_tile[HK * 64];               /* (Pv, Ph) of a part's HK columns, [column][lane] */
    __shared__ uint64_t s_peq[4 * 64];                   /* the four match masks of the block the lane's path is in, [code][lane] */
    /* rev: the problems are sorted by size, ascending -- the long paths start first (see lf_rsweep_body) */
    const int lane = threadIdx.x, idx = (rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * 64 + lane;
    const bool live = idx < n_probs;
    const lf_aln_prob pr = probs[live ? idx : n_probs - 1];
    const bool want = live && pr.task == LF_TASK_PATH;
    const uint32_t tl = pr.mode == 0 ? pr.m : (uint32_t)(out_end[pr.id] + 1);
    /* where the forward kernel left this problem's data: the wave's planes, then its checkpoint rows; block b was lane lane0 + b */
    const lf_hist_t *wbase = ckpt + pr.hist_base;
    uint32_t len = 0;
    lf_tb_core<HK>(pr, want, tl, (int)pr.pad, lane, wbase + LF_PLANE_ENTRIES, reinterpret_cast<const uint64_t *>(wbase), s_peq, s_tile, S.pac, pac_syms, ops, len);
    if (live) out_len[pr.id] = len;
}

/* ------------------------------------------------------------------------------------------------
 * generic kernel: any n; per-lane state in HBM (aux words), private FULL history in HBM.  Only for leaf-size problems whose
 * query is longer than the sweep classes take (n > 32768, hence m <= 101 below edlib's traceback switch).
 *   aux layout per problem: nbk x {lo,hi,valid,Pv,Mv}
 * ---------------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64)
lf_edlib_generic_kernel(const lf_aln_prob *__restrict__ probs, int n_probs, lf_seqs S, lf_hist_t *__restrict__ hist, uint64_t *__restrict__ aux,
                        uint8_t *__restrict__ ops, int32_t *__restrict__ out_ed, int32_t *__restrict__ out_end,
                        uint32_t *__restrict__ out_len)
{
    const int gid = blockIdx.x * 64 + threadIdx.x;
    if (gid >= n_probs) return;
    const lf_aln_prob pr = probs[gid];
    const lf_qacc Q(S.q, pr.qstart, pr.flags); const lf_tacc T(S.t, S.pac, pr.tstart, pr.flags);
    const bool lazy = (pr.flags & LF_F_LAZYX) != 0;      /* diagonal moves are not classified here (lf_render_kernel does it) */
    const uint32_t n = pr.n, m = pr.m;
    const uint32_t nbk = (n + 63) >> 6;
    uint64_t *st = aux + pr.aux_off;       /* [b*5 + {0 lo,1 hi,2 valid,3 Pv,4 Mv}] */
    for (uint32_t b = 0; b < nbk; b++) {
        uint64_t lo = 0, hi = 0, valid = 0;
        for (int i = 0; i < 64; i++) { const uint32_t r = b * 64 + i; if (r < n) lf_plane_add(Q.get(r), i, lo, hi, valid); }
        st[b * 5 + 0] = lo; st[b * 5 + 1] = hi; st[b * 5 + 2] = valid; st[b * 5 + 3] = ~0ull; st[b * 5 + 4] = 0;
    }
    const uint32_t lastb = (n - 1) >> 6; const int lastbit = (int)((n - 1) & 63);
    int score = (int)n;
    int best = (n & 63) ? (int)n : 0x7fffffff, best_c = 0;
    lf_hist_t *h = hist + pr.hist_base;
    const bool want_path = pr.task == LF_TASK_PATH;
    for (uint32_t c = 1; c <= m; c++) {
        const unsigned char tc = T.get(c - 1);
        uint32_t hin = LF_HIN_PLUS1;
        for (uint32_t b = 0; b < nbk; b++) {
            uint64_t Pv = st[b * 5 + 3], Mv = st[b * 5 + 4];
            const uint64_t Eq = lf_eq_mask(tc, st[b * 5 + 0], st[b * 5 + 1], st[b * 5 + 2], Q, n, b);
            uint64_t ph, mh;
            hin = lf_myers_step(Pv, Mv, Eq, hin, ph, mh);
            st[b * 5 + 3] = Pv; st[b * 5 + 4] = Mv;
            if (b == lastb) score += (int)((ph >> lastbit) & 1) - (int)((mh >> lastbit) & 1);
            if (want_path) { lf_hist_t e; e.pv = Pv; e.ph = ph; h[(size_t)(c - 1) * nbk + b] = e; }
        }
        if (score < best) { best = score; best_c = (int)c; }
    }
    int ed, tl;
    if (pr.mode == 0) { ed = score; tl = (int)m; } else { ed = best; tl = best_c; }
    out_ed[pr.id] = ed;
    out_end[pr.id] = tl - 1;
    if (!want_path) { out_len[pr.id] = 0; return; }
    uint8_t *o = ops + pr.ops_off;
    const uint32_t cap = n + m;
    uint32_t w = cap, r = n, c = (uint32_t)tl;
    if (c == 0) { while (r) { o[--w] = 1; r--; } }
    while (r > 0 && c > 0) {
        const lf_hist_t e = h[(size_t)(c - 1) * nbk + ((r - 1) >> 6)];
        const int bit = (int)((r - 1) & 63);
        if ((e.pv >> bit) & 1) { o[--w] = 1; r--; }
        else if ((e.ph >> bit) & 1) { o[--w] = 2; c--; }
        else { o[--w] = (lazy || Q.get(r - 1) == T.get(c - 1)) ? 0 : 3; r--; c--; }
    }
    while (c > 0) { o[--w] = 2; c--; }
    while (r > 0) { o[--w] = 1; r--; }
    out_len[pr.id] = cap - w;            /* ops are END-aligned: o[cap - len .. cap) */
}

/* ------------------------------------------------------------------------------------------------
 * sweep classes: G = 16 / 32 / 64 LANES PER PROBLEM (64 / G problems per wavefront), KB blocks per lane.
 *
 * The lanes of a group sweep the DP matrix as an anti-diagonal: at step s lane l works on column s - l + 1 of its
 * KB blocks and takes the horizontal carry lane l-1 produced one step earlier (DPP wave_shr:1).  m + lanes - 1 steps
 * instead of m * blocks dependent ones.  Loop bounds are wave-uniform maxima, work is predicated per group.
 *
 *   forward    ed / end column; every K steps the state of all lanes (Pv, Mv per block, pending carry) is
 *              checkpointed: one (KB + 1/16) KiB line per wave instead of K x KB KiB of history
 *   traceback  tiles of K steps, last first: restore the checkpoint, replay the K steps with (Pv, Ph) going to LDS,
 *              follow the path through the tile (position (r, c) lives at step c - 1 + owner(r), which only decreases)
 *   G = 64     adds obtainAlignmentHirschberg (lib/edlib/edlib.cpp:1161-1330) ON THE DEVICE for problems above
 *              edlib's 1 MiB traceback switch (:1117-1119): a depth-first walk over (query range, target range, distance)
 *              triples on an LDS stack, right child first, so that the leaves' paths land end-aligned in one piece.
 *              A split = forward sweep over the left half of the target, backward sweep over the right half (both
 *              write their last DP column to HBM scratch), split row = first hit in the order rows 0..n-2, then -1,
 *              then n-1 (:1263-1289) found with one ballot per 64 rows.  No host round trip, no byte staging.
 * The target is staged through an LDS ring (refilled every TC/2 steps), so any target length runs out of LDS.
 * ---------------------------------------------------------------------------------------------- */

template <int G, int KB, int K, bool PAC>
__global__ void __launch_bounds__(64)
lf_edlib_sweep_kernel(const lf_aln_prob *__restrict__ probs, int n_probs, lf_seqs S, lf_hist_t *__restrict__ ckpt, uint64_t *__restrict__ aux,
                      uint8_t *__restrict__ ops, int32_t *__restrict__ out_ed, int32_t *__restrict__ out_end, uint32_t *__restrict__ out_len)
{
    constexpr int P = 64 / G;                      /* problems per wave */
    constexpr int TC = G == 64 ? 256 : 128;        /* LDS ring of target bytes per group (power of two; H = TC / 2 >= G + K - 1) */
    constexpr int H = TC / 2;
    __shared__ lf_hist_t s_tile[K * KB * 64];
    __shared__ unsigned char s_t[P * TC];
    /* match masks of the lane's block against the four target codes (KB = 1, targets from the 2-bit reference): one
     * ds_read_b64 per step, issued a step ahead, instead of ten ALU ops on the bit planes; every lane touches its own column only */
    constexpr bool PEQ = PAC && KB == 1;
    __shared__ uint64_t s_peq[PEQ ? 4 * 64 : 1];
    const int lane = threadIdx.x, g = lane / G, gl = lane % G;
    const int pi = (int)blockIdx.x * P + g;
    const bool live = pi < n_probs;
    const lf_aln_prob pr = probs[live ? pi : n_probs - 1];      /* a dead group shadows the last problem and stores nothing */
    const lf_qacc Q(S.q, pr.qstart, pr.flags); const lf_tacc T(S.t, S.pac, pr.tstart, pr.flags);
    const bool lazy = (pr.flags & LF_F_LAZYX) != 0;      /* diagonal moves are not classified here (lf_render_kernel does it) */
    const bool want_path = pr.task == LF_TASK_PATH;
    unsigned char *my_t = s_t + g * TC;
    /* all groups of a wave share one checkpoint area (base of the wave's first problem) */
    lf_hist_t *ck = ckpt + probs[(int)blockIdx.x * P].hist_base + (KB == 1 ? LF_PLANE_ENTRIES : 0);
    constexpr int ROW = 64 * KB + (KB == 1 ? 16 : 4);      /* lf_sweep_row(KB) */
    const uint64_t gmask = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1);

    /* geometry of the (sub)problem being swept: query rows [qlo, qlo + n), target columns [tlo, tlo + m), both walked
     * backwards when rev (right halves of the Hirschberg splits) */
    uint32_t qlo = 0, n = pr.n, tlo = 0, m = pr.m; bool rev = false;
    auto qget = [&](uint32_t i) -> unsigned char { return Q.get(rev ? qlo + n - 1 - i : qlo + i); };
    auto tget = [&](uint32_t j) -> unsigned char { return T.get(rev ? tlo + m - 1 - j : tlo + j); };

    uint64_t lo[KB], hi[KB], valid[KB], Pv[KB], Mv[KB];
    uint32_t nbk = 0, lastb = 0; int lastbit = 0, nl = 0, lane_last = 0;
    uint32_t hout_prev = LF_HIN_PLUS1;               /* carry bits (lf_myers_step) this lane hands to its right neighbour */
    uint32_t cw = 0;                                 /* KB = 1: the carries this lane received during the current K steps, two bits per step */
    int ring_lo = 0;                                 /* the LDS ring holds target columns [ring_lo, ring_lo + TC) (wave-uniform) */

    /* bit planes by ballot: for block b the lanes fetch its 64 query bytes (G at a time per group) and three wave
     * ballots give lo / hi / valid; the lane that owns block b keeps them */
    auto build_planes = [&]() {
        nbk = (n + 63) >> 6; lastb = (n - 1) >> 6; lastbit = (int)((n - 1) & 63);
        nl = (int)((nbk + KB - 1) / KB); lane_last = (int)(lastb / KB);
        const uint32_t nbk_max = G == 64 ? nbk : lf_wave_max_u32(nbk);
#pragma unroll
        for (int k = 0; k < KB; k++) { lo[k] = hi[k] = valid[k] = 0; }
        for (uint32_t b = 0; b < nbk_max; b++) {
#pragma unroll
            for (int sub = 0; sub < P; sub++) {
                const uint32_t r = b * 64 + (uint32_t)(sub * G + gl);
                int code = -1;
                {   /* unconditional load from a clamped row (see lf_edlib_kernel) */
                    const bool in = b < nbk && r < n; bool ok; const uint32_t cd = lf_code_upper(qget(in ? r : n - 1), ok); code = (in && ok) ? (int)cd : -1;
                }
                const uint64_t bl = lf_ballot(code >= 0 && (code & 1)), bh = lf_ballot(code >= 0 && (code & 2)), bv = lf_ballot(code >= 0);
                if ((uint32_t)gl == b / KB) {
                    const int slot = (int)(b % KB);
                    const uint64_t xl = ((bl >> (g * G)) & gmask) << (sub * G), xh = ((bh >> (g * G)) & gmask) << (sub * G), xv = ((bv >> (g * G)) & gmask) << (sub * G);
#pragma unroll
                    for (int k = 0; k < KB; k++) if (k == slot) { lo[k] |= xl; hi[k] |= xh; valid[k] |= xv; }
                }
            }
        }
        if (PEQ) {
#pragma unroll
            for (uint32_t cde = 0; cde < 4; cde++) s_peq[cde * 64 + lane] = lf_eq_tok<true>(cde, lo[0], hi[0], valid[0], qget, n, (uint32_t)gl);
        }
    };
    /* target columns [first, first + count) into the group's LDS ring */
    auto stage_window = [&](int first, int count) {
        for (int j = first + gl; j < first + count; j += G) if (j >= 0 && (uint32_t)j < m)
            my_t[j & (TC - 1)] = PAC ? (unsigned char)T.pac_code(rev ? tlo + m - 1 - (uint32_t)j : tlo + (uint32_t)j) : tget((uint32_t)j);
    };
    /* one sweep step s (TILE: (Pv, Ph) of the step go to LDS row s - s0) */
    int score = 0, best = 0, best_c = 0;
    /* the bottom-row score is followed column by column only when a problem of the wavefront asks for the best prefix (SHW);
     * the NW distance is read off the last column afterwards, and the Hirschberg passes use neither */
    bool track_shw = lf_any(live && pr.mode != 0);
    auto sweep_step = [&](int s, auto track_c, auto tile_c, int s0, uint32_t byte, uint64_t eq_ahead) {
        constexpr bool track = decltype(track_c)::value, tile = decltype(tile_c)::value;
        const uint32_t from_left = lf_wave_shr1(hout_prev);
        if (!tile && KB == 1) cw |= from_left << ((s & (K - 1)) * 2);      /* forward passes; scalar shift amount */
        const int c = s - gl + 1;
        if (gl < nl && c >= 1 && c <= (int)m) {
            const uint32_t tok = PAC ? byte : lf_tok_of_byte((unsigned char)byte);
            uint32_t hin = gl == 0 ? LF_HIN_PLUS1 : from_left;
#pragma unroll
            for (int k = 0; k < KB; k++) {
                const uint32_t b = (uint32_t)gl * KB + k;
                /* blocks past the problem's last one (padding of the lane's KB) compute on dead registers: no branch */
                const uint64_t Eq = (PEQ && !tile) ? eq_ahead : lf_eq_tok<PAC>(tok, lo[k], hi[k], valid[k], qget, n, b);
                uint64_t ph, mh;
                const uint32_t ho = lf_myers_step(Pv[k], Mv[k], Eq, hin, ph, mh);
                hin = (KB == 1 || b < nbk) ? ho : hin;
                if (track) score += (b == lastb) ? lf_delta_at(ph, mh, lastbit) : 0;
                if (tile) { lf_hist_t e; e.pv = Pv[k]; e.ph = ph; s_tile[((s - s0) * KB + k) * 64 + lane] = e; }
            }
            hout_prev = hin;
            if (track) { const bool upd = gl == lane_last && score < best; best = upd ? score : best; best_c = upd ? c : best_c; }
        }
    };
    /* forward pass of the current geometry; want_ck: checkpoint every K steps.  Leaves score (NW distance at lane_last),
     * best / best_c (SHW) behind; returns the wave-uniform number of steps. */
    auto forward = [&](bool want_ck) -> int {
#pragma unroll
        for (int k = 0; k < KB; k++) { Pv[k] = ~0ull; Mv[k] = 0; }
        hout_prev = LF_HIN_PLUS1; cw = 0;
        score = (int)n; best = (n & 63) ? (int)n : 0x7fffffff; best_c = 0;
        const int steps = (int)m + nl - 1;
        const int steps_max = G == 64 ? steps : lf_wave_max_i32(steps);
        /* the lane's target symbol is read from the LDS ring ONE STEP AHEAD: the ds_read latency hides behind the step's ALU chain */
        __syncthreads(); stage_window(0, H); __syncthreads(); ring_lo = 0;
        uint32_t sym = my_t[(0 - gl) & (TC - 1)];
        uint64_t eq = PEQ ? s_peq[(sym & 3u) * 64 + lane] : 0ull;
        for (int s = 0; s < steps_max; s++) {
            if (((s + 1) & (H - 1)) == 0) { __syncthreads(); stage_window(s + 1, H); __syncthreads(); ring_lo = s + 1 >= H ? s + 1 - H : 0; }
            const uint32_t sym_next = my_t[(s + 1 - gl) & (TC - 1)];
            if (track_shw) sweep_step(s, std::true_type(), std::false_type(), 0, sym, eq); else sweep_step(s, std::false_type(), std::false_type(), 0, sym, eq);
            sym = sym_next;
            if (PEQ) eq = s_peq[(sym & 3u) * 64 + lane];
            if (want_ck && ((s + 1) & (K - 1)) == 0) {
                const size_t j = (size_t)((s + 1) / K - 1);
                lf_hist_t *row = ck + j * ROW;
#pragma unroll
                for (int k = 0; k < KB; k++) { lf_hist_t e; e.pv = Pv[k]; e.ph = Mv[k]; row[k * 64 + lane] = e; }
                reinterpret_cast<unsigned char *>(row + 64 * KB)[lane] = (unsigned char)hout_prev;
                if (KB == 1) reinterpret_cast<uint16_t *>(row + 64)[32 + lane] = (uint16_t)cw;
            }
            if (((s + 1) & (K - 1)) == 0) cw = 0;
        }
        /* the carries of the last, partial row (lf_edlib_tb_kernel replays single blocks from them) */
        if (KB == 1 && want_ck && (steps_max & (K - 1)) != 0) reinterpret_cast<uint16_t *>(ck + (size_t)(steps_max / K) * ROW + 64)[32 + lane] = (uint16_t)cw;
        return steps_max;
    };
    /* traceback of the current geometry from (r = n, c = tl) through checkpointed tiles; ops go to `em` backwards */
    lf_emitter em; em.init(ops + pr.ops_off, pr.n + pr.m, live && want_path && gl == 0);
    auto traceback = [&](uint32_t tl, int steps_max, bool active) {
        uint32_t r = active ? n : 0, c = active ? tl : 0;
        auto step_of = [&](uint32_t rr, uint32_t cc) -> int { return (rr > 0 && cc > 0) ? (int)(cc - 1 + ((rr - 1) >> 6) / KB) : -1; };
        int scur = step_of(r, c);
        const int jmax = lf_wave_max_i32(scur >= 0 ? scur / K : -1);
        /* the checkpoint in front of tile j is loaded while tile j + 1 is replayed and walked (HBM latency off the chain) */
        uint64_t nPv[KB], nMv[KB]; uint32_t nh = LF_HIN_PLUS1;
        auto fetch_ck = [&](int j) {
            if (j <= 0) {
#pragma unroll
                for (int k = 0; k < KB; k++) { nPv[k] = ~0ull; nMv[k] = 0; }
                nh = LF_HIN_PLUS1;
            } else {
                const lf_hist_t *row = ck + (size_t)(j - 1) * ROW;
#pragma unroll
                for (int k = 0; k < KB; k++) { const lf_hist_t e = row[k * 64 + lane]; nPv[k] = e.pv; nMv[k] = e.ph; }
                nh = (uint32_t)reinterpret_cast<const unsigned char *>(row + 64 * KB)[lane];
            }
        };
        if (jmax >= 0) fetch_ck(jmax);
        for (int j = jmax; j >= 0; j--) {
            const int s0 = j * K;
#pragma unroll
            for (int k = 0; k < KB; k++) { Pv[k] = nPv[k]; Mv[k] = nMv[k]; }
            hout_prev = nh;
            if (j > 0) fetch_ck(j - 1);
            __syncthreads();                                   /* the previous tile's LDS rows have been read */
            /* target columns of this tile: the ring still holds them unless the walk has moved left of it */
            if (s0 - G + 1 < ring_lo) { ring_lo = s0 + K > H ? s0 + K - H : 0; stage_window(ring_lo, H); __syncthreads(); }      /* H >= G + K - 1 columns: the tile's and the next ones' */
            const int s1 = s0 + K < steps_max ? s0 + K : steps_max;
            for (int s = s0; s < s1; s++) sweep_step(s, std::false_type(), std::true_type(), s0, (uint32_t)my_t[(s - gl) & (TC - 1)], 0ull);
            __syncthreads();
            for (;;) {
                const bool act = scur >= s0;
                if (!lf_any(act)) break;
                if (act) {
                    const uint32_t blk = (r - 1) >> 6;
                    const lf_hist_t e = s_tile[((scur - s0) * KB + (int)(blk % KB)) * 64 + g * G + (int)(blk / KB)];
                    const int bit = (int)((r - 1) & 63);
                    const uint32_t up = (uint32_t)(e.pv >> bit) & 1u, lf = ((uint32_t)(e.ph >> bit) & 1u) & ~up, dg = (up | lf) ^ 1u;
                    uint32_t op = up ? 1u : (lf ? 2u : 0u);      /* Up -> Left -> Diagonal (lib/edlib/edlib.cpp:950,984,1015) */
                    if (!lazy) { if (dg && qget(r - 1) != tget(c - 1)) op = 3u; }
                    em.put(op);
                    r -= up | dg; c -= lf | dg;
                    scur = step_of(r, c);
                }
            }
        }
        if (active) {
            while (c > 0) { em.put(2); c--; }
            while (r > 0) { em.put(1); r--; }
        }
    };

    /* ---- the problem (always below edlib's traceback switch: larger ones were cut into leaves by lf_hirsch.hip) ---- */
    const bool root_leaf = true;
    build_planes();
    if (KB == 1 && want_path && root_leaf) {      /* for lf_edlib_tb_kernel */
        uint64_t *pl = reinterpret_cast<uint64_t *>(ck - LF_PLANE_ENTRIES);
        pl[lane] = lo[0]; pl[64 + lane] = hi[0]; pl[128 + lane] = valid[0];
    }
    int steps_max = forward(want_path && root_leaf);
    const int src_last = g * G + lane_last;
    int ed_nw = __shfl(score, src_last);
    const int ed_shw = __shfl(best, src_last), c_shw = __shfl(best_c, src_last);
    if (!track_shw) {       /* D[n][m] = m + the vertical deltas of the last column (every lane stopped at column m) */
        int mine = 0;
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const uint32_t b = (uint32_t)gl * KB + k;
            if (b < nbk) {
                const uint32_t rows = (b == lastb) ? (uint32_t)lastbit + 1 : 64;
                const uint64_t msk = rows >= 64 ? ~0ull : ((1ull << rows) - 1);
                mine += __popcll(Pv[k] & msk) - __popcll(Mv[k] & msk);
            }
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        ed_nw = (int)m + mine;
    }
    track_shw = false;
    int ed, tl;
    if (pr.mode == 0) { ed = ed_nw; tl = (int)m; } else { ed = ed_shw; tl = c_shw; }
    if (live && gl == 0) { out_ed[pr.id] = ed; out_end[pr.id] = tl - 1; }
    if (KB == 1 && root_leaf) return;             /* wave-uniform (G < 64: always): path and out_len come from lf_edlib_tb_kernel */
    if (!lf_any(want_path)) { if (live && gl == 0) out_len[pr.id] = 0; return; }
    if (root_leaf) {
        traceback((uint32_t)tl, steps_max, want_path);
    }
    em.flush();
    if (live && gl == 0) out_len[pr.id] = want_path ? pr.n + pr.m - em.w : 0;            /* ops are END-aligned: o[cap - len .. cap) */
}


/* ------------------------------------------------------------------------------------------------
 * classes and checkpoint layout
 *   1   lf_edlib_rsweep_kernel (lf_rsweep.hip): every problem with <= 64 query blocks (n <= 4096) below edlib's traceback
 *       switch -- one block per lane, floor(64 / nb) problems of the same nb per wavefront; path by lf_edlib_tb_kernel
 *   2,3 sweep kernel G 64 with KB 4 / 8 blocks per lane (n <= 16384 / 32768; below the switch such a problem has m < 800),
 *       forward + its own tile traceback
 *   0   generic lane kernel: n > 32768 (m <= 101 below the switch), and problems of the stage API whose target holds bytes
 *       other than upper-case ACGT (exact byte compare)
 *   4   not a problem of these kernels: an empty side (a pure run, written at once) or a problem above edlib's traceback
 *       switch (:1117-1119) -- a ROOT of the breadth-first Hirschberg levels (lf_hirsch.hip), whose leaves come back as
 *       extra problems of classes 0..3
 * Sort key (32 bits): class << 28 | mode << 23 | nb << 16 | m   (mode, nb: class 1 only; m < 37 450 below the switch).
 * A wave's checkpoints start at the hist_base of its first problem (16-byte entries).
 * ---------------------------------------------------------------------------------------------- */
#define LF_NCLASS 4
#define LF_CLASS_SKIP 4
__host__ __device__ __forceinline__ int lf_class_of(uint32_t n, uint32_t m, bool exotic)
{
    if (n == 0 || m == 0 || !lf_leaf(n, m)) return LF_CLASS_SKIP;
    if (exotic) return 0;
    if (n <= 4096) return 1; if (n <= 16384) return 2; if (n <= LF_SWEEP_MAX_N) return 3;
    return 0;
}
__host__ __device__ __forceinline__ uint32_t lf_class_key(const lf_aln_desc_t &x)
{
    const int c = lf_class_of(x.n, x.m, x.pad[0] != 0);
    const uint32_t m16 = x.m < 0xffffu ? x.m : 0xffffu;
    const uint32_t nbf = ((x.n + 63) >> 6) + (lf_small_prob(x.n, x.m) ? 64u : 0u);      /* small problems: segments of their own */
    return ((uint32_t)c << 28) | (c == 1 ? ((uint32_t)(x.mode ? 1u : 0u) << 23) | (nbf << 16) : 0u) | m16;
}
/* checkpoint entries of one rsweep wave: the planes, then one row per 32 steps (+ the partial last one) */
__host__ __device__ __forceinline__ uint64_t lf_rwave_entries(uint32_t nb, uint32_t m_max) { return LF_PLANE_ENTRIES + (uint64_t)((m_max + nb - 1 + LF_RSTEPS - 1) / LF_RSTEPS + 1) * (uint32_t)LF_RROW; }
/* ... of a G 64 / KB wave (one problem) */
__host__ __device__ __forceinline__ uint64_t lf_kbwave_entries(int kb, int k, uint32_t m) { return (((uint64_t)m + 64) / (uint32_t)k + 1) * (uint64_t)lf_sweep_row(kb); }

struct lf_rseg_tab { int lo[257]; int w0[257]; int cstart[8]; int n_waves[2]; };

/* ------------------------------------------------------------------------------------------------
 * Batches are binned, ordered and laid out ON THE GPU: one radix sort of 32-bit keys, a 256-thread kernel that finds the
 * (mode, nb) segments of class 1 and numbers their wavefronts, one scan for the checkpoint bases, one kernel that writes the
 * problem array and the wave table.  The host uploads 32-byte descriptors (or finds them in HBM: lf_walk.hip) and launches;
 * it does no per-problem work.  Items n .. n + n_h - 1 are the leaves the Hirschberg levels made out of the large problems.
 * ---------------------------------------------------------------------------------------------- */
struct lf_desc_src {
    const lf_aln_desc_t *d; const uint64_t *ops_off; uint64_t ops_total;       /* host arrays (or null: dev_desc / dev_opsoff) */
    const unsigned char *d_q, *d_t; const uint8_t *d_pac; int64_t pac_syms;   /* sequences in HBM: read batch + 2-bit reference; stage API: uploaded bytes + their 2-bit form */
    const uint64_t *d_planes; int64_t q_words;                                /* bit planes of d_q (lf_pack_planes_kernel) */
    bool pac;                                                                 /* descriptors address the reference itself (pipeline) */
    lf_hcount_t hc;                                                           /* the problems above edlib's traceback switch (known to whoever made the descriptors) */
};

/* Binning is a COUNTING sort, not a comparison / radix sort: what the launch layout needs is the problems grouped by class and,
 * inside class 1, by (mode, nb) with similar target lengths next to each other (a wavefront's sweep lasts as long as its longest
 * target) -- not a total order.  A bin = (class, mode, nb, bucket of the target length: 32 columns wide below 512, 256 above);
 * 4 100 bins.  Most problems of a chunk fall into a handful of bins (one or two blocks, short targets), so neither pass sends
 * its atomics to HBM one by one (a first version did: 28 ms per 100 k reads on a few hot counters): a workgroup counts its
 * 4 096 problems in an LDS histogram, adds the non-zero counters to the global ones (pass 1), and after a single-workgroup scan
 * reserves ONE range per non-zero bin and hands its places out through LDS (pass 2).  The order inside a bin is whatever the
 * atomics give -- problems are independent, results do not depend on it. */
#define LF_BIN_MB 32
#define LF_NBINS (1 + 2 * LF_SEG_NB_MAX * LF_BIN_MB + 3)
#define LF_BIN_ITEMS 4                       /* problems per thread of the two counting passes (1024 threads) */
__host__ __device__ __forceinline__ uint32_t lf_bin_of_key(uint32_t key)
{
    const uint32_t c = key >> 28;
    if (c == 0) return 0u;
    if (c != 1) return 1u + 2u * LF_SEG_NB_MAX * LF_BIN_MB + (c - 2u);
    const uint32_t mode = (key >> 23) & 1u, nb = (key >> 16) & 127u, m = key & 0xffffu;
    const uint32_t mb = m < 512u ? (m >> 5) : 16u + (((m - 512u) >> 8) < 15u ? ((m - 512u) >> 8) : 15u);
    return 1u + (mode * LF_SEG_NB_MAX + (nb - 1u)) * LF_BIN_MB + mb;
}
__global__ void lf_desc_keys_kernel(const lf_aln_desc_t *__restrict__ d, const uint64_t *__restrict__ ops_off, int n, const lf_aln_desc_t *__restrict__ hd, int n_h,
                                    uint32_t *__restrict__ keys, uint8_t *__restrict__ ops,
                                    int32_t *__restrict__ out_ed, int32_t *__restrict__ out_end, uint32_t *__restrict__ out_len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + n_h) return;
    const lf_aln_desc_t x = i < n ? d[i] : hd[i - n];
    keys[i] = lf_class_key(x);
    if (i < n && (x.n == 0 || x.m == 0)) {
        /* one side empty: no DP (lib/edlib/edlib.cpp:1096-1104).  n == 0: NW deletes the whole target, SHW takes the empty
         * prefix; m == 0: the query is inserted.  The run is end-aligned in the region like every other path. */
        const uint32_t len = x.n == 0 ? (x.mode == 0 ? x.m : 0u) : x.n;
        const uint8_t op = x.n == 0 ? 2 : 1;
        uint8_t *o = ops + ops_off[i] + (x.n + x.m - len);
        for (uint32_t k = 0; k < len; k++) o[k] = op;
        out_ed[i] = (int32_t)len; out_end[i] = x.n == 0 ? (int32_t)len - 1 : -1; out_len[i] = len;
    }
}
/* exclusive scan of the bin counters, in place (one workgroup; 24 580 counters): hist[b] becomes the first place of bin b */
__global__ void __launch_bounds__(1024)
lf_desc_binscan_kernel(uint32_t *__restrict__ hist)
{
    __shared__ uint32_t s_sum[1024];
    const int t = threadIdx.x;
    constexpr int PER = (LF_NBINS + 1023) / 1024;
    uint32_t loc[PER]; uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { const int b = t * PER + k; loc[k] = b < LF_NBINS ? hist[b] : 0u; acc += loc[k]; }
    s_sum[t] = acc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { const uint32_t v = t >= o ? s_sum[t - o] : 0u; __syncthreads(); s_sum[t] += v; __syncthreads(); }
    uint32_t base = t ? s_sum[t - 1] : 0u;
#pragma unroll
    for (int k = 0; k < PER; k++) { const int b = t * PER + k; if (b < LF_NBINS) hist[b] = base; base += loc[k]; }
}
/* pass 1 (SCATTER = false): per-workgroup LDS histogram -> global counters; pass 2 (SCATTER = true): the same histogram, one
 * range of the (scanned) global cursor per non-zero bin, places inside the range through a second LDS counter */
template <bool SCATTER>
__global__ void __launch_bounds__(1024)
lf_desc_count_kernel(const uint32_t *__restrict__ keys, int n, uint32_t *__restrict__ bins, uint32_t *__restrict__ keys2, uint32_t *__restrict__ vals2)
{
    __shared__ uint32_t s_cnt[LF_NBINS], s_base[SCATTER ? LF_NBINS : 1];
    const int t = threadIdx.x;
    for (int b = t; b < LF_NBINS; b += 1024) s_cnt[b] = 0;
    __syncthreads();
    const int i0 = (int)blockIdx.x * 1024 * LF_BIN_ITEMS;
    uint32_t key[LF_BIN_ITEMS], bin[LF_BIN_ITEMS], rank[LF_BIN_ITEMS];
#pragma unroll
    for (int k = 0; k < LF_BIN_ITEMS; k++) {
        const int i = i0 + k * 1024 + t;
        bin[k] = 0xffffffffu; key[k] = 0; rank[k] = 0;
        if (i < n) { key[k] = keys[i]; bin[k] = lf_bin_of_key(key[k]); rank[k] = atomicAdd(&s_cnt[bin[k]], 1u); }
    }
    __syncthreads();
    for (int b = t; b < LF_NBINS; b += 1024) {
        const uint32_t c = s_cnt[b];
        if (c) { const uint32_t g = atomicAdd(&bins[b], c); if (SCATTER) s_base[b] = g; }
    }
    if (!SCATTER) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LF_BIN_ITEMS; k++) {
        const int i = i0 + k * 1024 + t;
        if (i < n) { const uint32_t at = s_base[bin[k]] + rank[k]; keys2[at] = key[k]; vals2[at] = (uint32_t)i; }
    }
}
__device__ __forceinline__ int lf_lower_bound_u32(const uint32_t *__restrict__ keys, int n, uint32_t want)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
    return lo;
}
/* class starts; for class 1 the start of every (mode, nb) segment and the number of its first wavefront */
__global__ void __launch_bounds__(256)
lf_desc_segments_kernel(const uint32_t *__restrict__ keys, int n, lf_rseg_tab *__restrict__ tab)
{
    __shared__ int s_lo[257], s_w[256];
    const int t = threadIdx.x;
    s_lo[t] = lf_lower_bound_u32(keys, n, (1u << 28) | ((uint32_t)t << 16));
    if (t == 255) s_lo[256] = lf_lower_bound_u32(keys, n, 2u << 28);
    if (t < 6) tab->cstart[t] = lf_lower_bound_u32(keys, n, (uint32_t)t << 28);
    __syncthreads();
    const int nbf = t & 127, nb = lf_seg_blocks(nbf), cnt = s_lo[t + 1] - s_lo[t];
    s_w[t] = (nbf >= 1 && nbf <= LF_SEG_NB_MAX) ? (cnt + (64 / nb) - 1) / (64 / nb) : 0;
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int k = 0; k < 256; k++) { tab->lo[k] = s_lo[k]; tab->w0[k] = acc; acc += s_w[k]; if (k == 127) tab->n_waves[0] = acc; }
        tab->lo[256] = s_lo[256]; tab->w0[256] = acc; tab->n_waves[1] = acc - tab->n_waves[0];
    }
}
/* checkpoint entries are charged to the first problem of every wave */
__global__ void lf_desc_entries_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, const lf_aln_desc_t *__restrict__ d, int n0,
                                       const lf_aln_desc_t *__restrict__ hd, const lf_rseg_tab *__restrict__ tab, int n, uint64_t *__restrict__ ent, int fused_small)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t key = keys[j];
    const int c = (int)(key >> 28);
    uint64_t e = 0;
    if (c == 1) {
        const int t = (int)((key >> 16) & 0xffu), nb = lf_seg_blocks(t & 127), P = 64 / nb, rel = j - tab->lo[t];
        if (((t & 127) <= 64 || !fused_small) && rel % P == 0) {      /* (the fused kernel keeps the small problems' rows in LDS) the wave's checkpoint rows: as many as its LONGEST target needs (a bin holds a range of lengths, in no particular order) */
            int last = j + P - 1; if (last > tab->lo[t + 1] - 1) last = tab->lo[t + 1] - 1;
            uint32_t mmax = 0;
            for (int k = j; k <= last; k++) { const uint32_t mk = keys[k] & 0xffffu; mmax = mk > mmax ? mk : mmax; }
            e = lf_rwave_entries((uint32_t)nb, mmax);
        }
    } else if (c != LF_CLASS_SKIP) {
        const uint32_t i = vals[j];
        const lf_aln_desc_t x = i < (uint32_t)n0 ? d[i] : hd[i - n0];
        e = c == 0 ? (uint64_t)x.m * ((x.n + 63) >> 6) : c == 2 ? lf_kbwave_entries(4, 4, x.m) : lf_kbwave_entries(8, 2, x.m);      /* class 0: full history, m rows of nb entries */
    }
    ent[j] = e;
}
__global__ void lf_desc_build_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, const lf_aln_desc_t *__restrict__ d, int n0,
                                     const lf_aln_desc_t *__restrict__ hd, const uint64_t *__restrict__ ops_off, const uint64_t *__restrict__ hops_off,
                                     const lf_rseg_tab *__restrict__ tab, const uint64_t *__restrict__ base, int n, int pac,
                                     lf_aln_prob *__restrict__ probs, lf_rwave *__restrict__ waves, uint64_t *__restrict__ aux_words_total)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t key = keys[j];
    const int c = (int)(key >> 28);
    if (c == LF_CLASS_SKIP) return;
    const uint32_t i = vals[j];
    const lf_aln_desc_t x = i < (uint32_t)n0 ? d[i] : hd[i - n0];
    lf_aln_prob p;
    p.qstart = x.qstart; p.tstart = x.tstart; p.ops_off = i < (uint32_t)n0 ? ops_off[i] : hops_off[i - n0];
    p.hist_base = base[j]; p.aux_off = 0; p.pad = 0;
    if (c == 1) {
        const int t = (int)((key >> 16) & 0xffu), nb = lf_seg_blocks(t & 127), P = 64 / nb, rel = j - tab->lo[t], slot = rel % P;
        p.hist_base = base[j - slot];
        p.pad = (uint8_t)(slot * nb);                       /* the problem's first lane in its wavefront */
        if (slot == 0) {
            lf_rwave w; w.first = (uint32_t)j; const int left = tab->lo[t + 1] - j; w.count = (uint16_t)(left < P ? left : P); w.G = (uint16_t)nb; w.hist_base = p.hist_base;      /* (small segments: no rows in HBM, hist_base unused) */
            waves[tab->w0[t] + rel / P] = w;
        }
    } else if (c == 0) {
        p.aux_off = atomicAdd((unsigned long long *)aux_words_total, (unsigned long long)((x.n + 63) >> 6) * 5);      /* private state words of the generic kernel */
    }
    p.n = x.n; p.m = x.m; p.id = i; p.mode = x.mode; p.task = LF_TASK_PATH;
    p.flags = (uint8_t)((x.flags & ~LF_F_TPAC) | ((pac || !x.pad[0]) ? LF_F_TPAC : 0u));
    probs[j] = p;
}

/* HIP-event brackets of the last alignment batch of the calling thread: forward sweeps (lf_edlib_rsweep_kernel, both modes),
 * traceback (lf_edlib_tb_kernel), Hirschberg levels (incl. their per-level readbacks), binning (keys, sort, segments, scan, build).
 * Alone on the GPU only when the classes run on one stream (LF_SERIAL_CLASSES: bench.py's exclusive pass). */
static thread_local float t_breakdown[4];
/* A device-planned round (lfg_edlib_desc_dev) hands nothing to the host: its consumer, lf_walk_emit_kernel, is queued behind it on the same stream, and
 * the call returns without waiting.  Its event brackets are read when somebody asks -- lf_pipeline.c does after lfg_walk_emit's wait. */
static thread_local struct { bool on, class1; hipEvent_t e0, e1, eb, bd[4]; float ms; } t_pend;
static void breakdown_resolve(void)
{
    if (!t_pend.on) return;
    t_pend.on = false;
    if (hipEventElapsedTime(&t_pend.ms, t_pend.e0, t_pend.e1) != hipSuccess) { (void)hipGetLastError(); t_pend.ms = 0; return; }
    if (t_pend.class1) { (void)hipEventElapsedTime(&t_breakdown[0], t_pend.bd[1], t_pend.bd[2]); (void)hipEventElapsedTime(&t_breakdown[1], t_pend.bd[2], t_pend.bd[3]); }
    (void)hipEventElapsedTime(&t_breakdown[2], t_pend.e0, t_pend.bd[0]); (void)hipEventElapsedTime(&t_breakdown[3], t_pend.bd[0], t_pend.eb);
    (void)hipGetLastError();
}
extern "C" void lfg_edlib_breakdown(float *out4) { breakdown_resolve(); for (int k = 0; k < 4; k++) out4[k] = t_breakdown[k]; }
extern "C" float lfg_edlib_round_ms(void) { breakdown_resolve(); return t_pend.ms; }      /* e0 .. e1 of the calling thread's last device-planned round */

/* dev_desc / dev_opsoff != nullptr: the descriptors are already on the device (lfg_walk_plan) and the results stay there
 * (res_dev[0..2] = ed, end column, path length arrays) */
static int run_edlib_desc_gpu(int device, int n, const lf_desc_src *D, int32_t *ed, int32_t *endloc, uint8_t *ops, uint32_t *ops_len,
                              int ops_slot, void **ops_dev, void **desc_dev, float *ms,
                              const lf_aln_desc_t *dev_desc = nullptr, const uint64_t *dev_opsoff = nullptr, void **res_dev = nullptr)
{
    if (ms) *ms = 0;
    for (int k = 0; k < 4; k++) t_breakdown[k] = 0;
    t_pend.on = false; t_pend.ms = 0;
    if (n == 0) return LF_OK;
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)lfg_lane_stream(device, 1);
    if (!s) return LF_ERR_HIP;
    const bool serial_classes = lf_env_set("LF_SERIAL_CLASSES") != 0;     /* profiling aid: one class at a time (read per call: bench.py switches it) */
    hipEvent_t bd[5]; for (int k = 0; k < 5; k++) { bd[k] = (hipEvent_t)lfg_lane_event(device, 4 + k); if (!bd[k]) return LF_ERR_HIP; }
    hipStream_t cs[LF_NCLASS];
    for (int k = 0; k < LF_NCLASS; k++) { cs[k] = serial_classes ? s : (hipStream_t)lfg_lane_stream(device, 2 + k); if (!cs[k]) return LF_ERR_HIP; }
    hipEvent_t cdone[LF_NCLASS], e0 = (hipEvent_t)lfg_lane_event(device, 12), e1 = (hipEvent_t)lfg_lane_event(device, 13), eb = (hipEvent_t)lfg_lane_event(device, 14);
    for (int k = 0; k < LF_NCLASS; k++) { cdone[k] = (hipEvent_t)lfg_lane_event(device, k); if (!cdone[k]) return LF_ERR_HIP; }
    if (!e0 || !e1 || !eb) return LF_ERR_HIP;
    const lf_hcount_t HC = D->hc;
    const size_t hcap = HC.roots ? (size_t)HC.cap : 0, NN = (size_t)n + hcap;      /* upper bound of the items to bin */
    if (NN >= (1ull << 31)) { lf_set_error("edlib batch too large (%zu problems)", NN); return LF_ERR_ARG; }
#define DSLOT(T, k, bytes) (T *)lfg_dev_slot(device, LF_DS_ALN0 + (k), (bytes))
    /* the descriptors stay with the round's paths when those stay in HBM (lazy paths are resolved against them later) */
    lf_aln_desc_t *d_desc = dev_desc ? const_cast<lf_aln_desc_t *>(dev_desc)
                          : ops ? DSLOT(lf_aln_desc_t, 0, (size_t)n * sizeof(lf_aln_desc_t))
                                : (lf_aln_desc_t *)lfg_dev_slot(device, ops_slot + 1, (size_t)n * sizeof(lf_aln_desc_t));
    if (desc_dev) *desc_dev = d_desc;
    uint64_t *d_opsoff = dev_opsoff ? const_cast<uint64_t *>(dev_opsoff) : DSLOT(uint64_t, 1, (size_t)n * 8);
    uint32_t *d_keys = DSLOT(uint32_t, 3, NN * 4), *d_keys2 = DSLOT(uint32_t, 8, NN * 4);
    uint32_t *d_vals = DSLOT(uint32_t, 9, NN * 4), *d_vals2 = DSLOT(uint32_t, 10, NN * 4);
    uint64_t *d_ent = DSLOT(uint64_t, 11, NN * 8), *d_base = DSLOT(uint64_t, 12, NN * 8 + 8);
    lf_aln_prob *d_probs = DSLOT(lf_aln_prob, 13, NN * sizeof(lf_aln_prob));
    lf_rwave *d_waves = DSLOT(lf_rwave, 0 + 17, (NN + 256) * sizeof(lf_rwave));
    /* device-planned rounds keep their results in slots of their own: the host-planned rounds that follow (rare chains)
     * must not overwrite what lf_walk_emit_kernel still reads */
    const int rs = dev_desc ? 18 : 4;
    int32_t *d_ed = DSLOT(int32_t, rs, NN * 4), *d_end = DSLOT(int32_t, rs + 1, NN * 4);
    uint32_t *d_len = DSLOT(uint32_t, rs + 2, NN * 4);
    uint8_t *d_ops = ops ? DSLOT(uint8_t, 7, D->ops_total + 64) : (uint8_t *)lfg_dev_slot(device, ops_slot, D->ops_total + 64);
    if (res_dev) { res_dev[0] = d_ed; res_dev[1] = d_end; res_dev[2] = d_len; }
    if (ops_dev) *ops_dev = d_ops;
    lf_rseg_tab *d_tab = DSLOT(lf_rseg_tab, 14, sizeof(lf_rseg_tab));
    uint64_t *d_misc = DSLOT(uint64_t, 15, 64);
    lf_rseg_tab *h_tab = (lf_rseg_tab *)lfg_pin_slot(LF_PS_ALN_PROB + 2, sizeof(lf_rseg_tab) + 64);
    if (!d_desc || !d_opsoff || !d_keys || !d_keys2 || !d_vals || !d_vals2 || !d_ent || !d_base || !d_probs || !d_waves || !d_ed || !d_end || !d_len || !d_ops || !d_tab || !d_misc || !h_tab) return LF_ERR_NOMEM;
    uint32_t *d_bins = (uint32_t *)lfg_dev_slot(device, LF_DS_ALN0 + 16, (size_t)LF_NBINS * 4 + 256);
    if (!d_bins) return LF_ERR_NOMEM;

    if (!dev_desc) {
        HIPCHK(hipMemcpyAsync(d_desc, D->d, (size_t)n * sizeof(lf_aln_desc_t), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_opsoff, D->ops_off, (size_t)n * 8, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemsetAsync(d_misc, 0, 64, s));
    HIPCHK(hipEventRecord(e0, s));

    /* ---- the problems above edlib's traceback switch: breadth-first Hirschberg levels (lf_hirsch.hip) ---- */
    lf_hargs HA; memset(&HA, 0, sizeof HA);
    uint32_t n_h = 0, n_roots = 0;
    lf_aln_desc_t *d_hdesc = nullptr; uint64_t *d_hopsoff = nullptr;
    if (HC.roots) {
        const size_t q_cap = hcap;
        lf_hctl *d_ctl = DSLOT(lf_hctl, 21, sizeof(lf_hctl));
        lf_hroot *d_roots = DSLOT(lf_hroot, 22, (size_t)HC.roots * sizeof(lf_hroot));
        lf_hseg *d_segs = (lf_hseg *)lfg_dev_slot(device, LF_DS_ALN0 + 23, hcap * sizeof(lf_hseg));
        lf_hnode *d_q = (lf_hnode *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 0, 2 * LF_HQ * q_cap * sizeof(lf_hnode));
        d_hdesc = (lf_aln_desc_t *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 1, hcap * sizeof(lf_aln_desc_t));
        d_hopsoff = (uint64_t *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 2, hcap * 8);
        const uint64_t aux_cap = 6 * (HC.sum_n / 64 + HC.roots + hcap) + 64, hcar_cap = 2 * HC.sum_m + 64 * hcap + 4096 + 1024 * (HC.sum_n / 32768 + 2);      /* (lf_hband_reserve: at most sum_n / 32768 nodes of a level sweep in super-bands) */
        uint64_t *d_haux = (uint64_t *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 3, aux_cap * 8);
        uint8_t *d_hcar = (uint8_t *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 4, hcar_cap);
        lf_hctl *h_ctl = (lf_hctl *)lfg_pin_slot(LF_PS_ALN_PROB + 1, sizeof(lf_hctl));
        if (!d_ctl || !d_roots || !d_segs || !d_q || !d_hdesc || !d_hopsoff || !d_haux || !d_hcar || !h_ctl) return LF_ERR_NOMEM;
        HIPCHK(hipMemsetAsync(d_ctl, 0, sizeof(lf_hctl), s));
        HA.S.q = D->d_q; HA.S.t = D->d_t; HA.S.pac = D->d_pac; HA.pac_syms = D->pac_syms;
        lf_htrial_pick(HA.trial16);
        HA.trial_min = LF_HTRIAL_MIN_ROWS;
        if (const char *e_ = lf_env("LF_HIRSCH_TRIAL")) {      /* test hook: fixed bounds "nw,shw", optionally ",rows" (only roots above that many rows try) */
            unsigned a_ = 0, b_ = 0, c_ = 0; const int got_ = sscanf(e_, "%u,%u,%u", &a_, &b_, &c_);
            if (got_ >= 2 && a_ <= 16 && b_ <= 16) { HA.trial16[0] = a_; HA.trial16[1] = b_; if (got_ == 3) HA.trial_min = c_; }
        }
        { const long hb_ = lf_env_long("LF_HIRSCH_BAND", 1); HA.no_band = hb_ == 0 ? 1u : hb_ == 64 ? 2u : hb_ == 3 ? 4u : 0u;      /* (read per call: the tests switch it) */
          /* the lane-group queues (lf_hband_group_kernel) are for levels that are bound by VALU issue -- far more nodes than the GPU holds wavefronts; a call with few roots is
           * bound by its longest chains and by the number of launches per level, and four more queues only add launches (C4, 11 k roots per call: 202 -> 213 ms per step with
           * them).  LF_HIRSCH_BAND=16 keeps them on for any call (tests). */
          if (hb_ != 16 && HC.roots < 16384) HA.no_band |= 2u; }
        HA.qlo = D->d_planes; HA.qhi = D->d_planes + D->q_words; HA.qvalid = D->d_planes + 2 * D->q_words; HA.q_words = D->q_words;
        HA.q_cap = (uint32_t)q_cap; HA.ctl = d_ctl; HA.roots = d_roots; HA.segs = d_segs; HA.hdesc = d_hdesc; HA.hopsoff = d_hopsoff; HA.hleaf_cap = (uint32_t)hcap;
        HA.aux = d_haux; HA.aux_cap = aux_cap; HA.hcar = d_hcar; HA.hcar_cap = hcar_cap;
        HA.ops = d_ops; HA.out_ed = d_ed; HA.out_end = d_end; HA.out_len = d_len; HA.n_desc = (uint32_t)n;
        auto queue = [&](int par, int kbc) { return d_q + ((size_t)par * LF_HQ + kbc) * q_cap; };
        int par = 0;
        for (int k = 0; k < LF_HQ; k++) HA.q_out[k] = queue(0, k);
        HA.out_par = 0;
        lf_hirsch_launch_roots(s, D->pac, d_desc, d_opsoff, n, HA);
        const bool hdbg = lf_env_set("LF_HIRSCH_DEBUG") != 0;      /* per call: roots and their sizes; per level: nodes by class, milliseconds since the previous readback */
        std::chrono::steady_clock::time_point hd_t0 = std::chrono::steady_clock::now();
        if (hdbg) fprintf(stderr, "[lf] hirschberg: %llu roots, sum n %llu, sum m %llu (of %d problems)\n", (unsigned long long)HC.roots, (unsigned long long)HC.sum_n, (unsigned long long)HC.sum_m, n);
        for (int level = 0; level < 64; level++) {
            HIPCHK(hipMemcpyAsync(h_ctl, d_ctl, sizeof(lf_hctl), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (hdbg) {
                const std::chrono::steady_clock::time_point t1 = std::chrono::steady_clock::now();
                fprintf(stderr, "[lf]   level %d: unbanded %u / %u / %u (<= 4096 / <= 16384 / more rows), banded NW %u / %u / %u / %u (1 / 2 / 4 / 8 wavefronts per half), SHW %u / %u / %u / %u / %u (1 .. 16 wavefronts), sixteen-lane NW %u SHW %u, thirty-two-lane NW %u SHW %u, trials %u failed %u, leaves so far %u, %.3f ms\n", level,
                        h_ctl->q_n[par][0], h_ctl->q_n[par][1], h_ctl->q_n[par][2], h_ctl->q_n[par][3], h_ctl->q_n[par][4], h_ctl->q_n[par][5], h_ctl->q_n[par][6],
                        h_ctl->q_n[par][7], h_ctl->q_n[par][8], h_ctl->q_n[par][9], h_ctl->q_n[par][10], h_ctl->q_n[par][11], h_ctl->q_n[par][12], h_ctl->q_n[par][13], h_ctl->q_n[par][14], h_ctl->q_n[par][15], h_ctl->n_trial, h_ctl->n_trial_failed, h_ctl->n_hleaf,
                        std::chrono::duration<double, std::milli>(t1 - hd_t0).count());
                hd_t0 = t1;
            }
            if (h_ctl->fail > 1) { lf_set_error("edlib Hirschberg levels: scratch bound exceeded (code %u)", h_ctl->fail); return LF_ERR_HIP; }
            n_roots = h_ctl->n_roots; n_h = h_ctl->n_hleaf;
            uint32_t cntq[LF_HQ], cnt_all = 0;
            for (int k = 0; k < LF_HQ; k++) { cntq[k] = h_ctl->q_n[par][k]; cnt_all += cntq[k]; }
            /* statistics of the lane's calls (lf_stats_t.hirsch_*): lane values 4 .. 6 (0: plane words, lf_seed.hip; 2, 3: lines of the SAM buffers, lf_sam.hip) */
            { uint64_t nb_ = 0, nu_ = 0; for (int k = 0; k < LF_HQ; k++) (k >= LF_HQ_NW0 ? nb_ : nu_) += cntq[k];
              const uint64_t mx_ = lfg_lane_value(device, 4); if (h_ctl->max_root_n > mx_) lfg_lane_set_value(device, 4, h_ctl->max_root_n);
              lfg_lane_set_value(device, 5, lfg_lane_value(device, 5) + nb_); lfg_lane_set_value(device, 6, lfg_lane_value(device, 6) + nu_); }
            if (cnt_all == 0) break;
            /* next level's counters and this level's scratch cursors */
            HIPCHK(hipMemsetAsync((char *)d_ctl + offsetof(lf_hctl, q_n) + (size_t)(par ^ 1) * sizeof(h_ctl->q_n[0]), 0, sizeof(h_ctl->q_n[0]), s));
            HIPCHK(hipMemsetAsync((char *)d_ctl + offsetof(lf_hctl, aux_used), 0, 16, s));
            for (int k = 0; k < LF_HQ; k++) HA.q_out[k] = queue(par ^ 1, k);
            HA.out_par = (uint32_t)(par ^ 1);
            /* the classes with the longest sweeps first: wide bands, unbanded large queries, ... */
            static const int order[LF_HQ] = { 2, 1, 11, 6, 10, 5, 9, 4, 8, 3, 7, 0, 14, 15, 12, 13 };
            /* A level's queues are independent, and each lasts as long as its longest node: one after the other on one stream a level cost the SUM of its
             * queues' longest sweeps (round 6 has twelve queues where round 5 had three).  They go out side by side on the lane's class streams -- idle
             * until the binning below -- and the level's stream waits for all of them. */
            hipStream_t ls[6]; int n_ls = 1; ls[0] = s;
            if (!serial_classes) { for (int k = 0; k < LF_NCLASS && n_ls < 6; k++) ls[n_ls++] = cs[k]; hipStream_t x6 = (hipStream_t)lfg_lane_stream(device, 6); if (x6 && n_ls < 6) ls[n_ls++] = x6; }
            hipEvent_t lfork = (hipEvent_t)lfg_lane_event(device, 28); if (!lfork) return LF_ERR_HIP;
            bool used_ls[6] = { false, false, false, false, false, false }; int slot = 0;
            HIPCHK(hipEventRecord(lfork, s));
            for (int o = 0; o < LF_HQ; o++) {
                const int k = order[o];
                if (!cntq[k]) continue;
                const int u = slot % n_ls; slot++;
                if (u != 0 && !used_ls[u]) HIPCHK(hipStreamWaitEvent(ls[u], lfork, 0));
                used_ls[u] = true;
                HA.q_in = queue(par, k); HA.n_in = cntq[k]; lf_hirsch_launch_level(ls[u], D->pac, k, HA);
            }
            for (int u = 1; u < n_ls; u++) if (used_ls[u]) {
                hipEvent_t ld = (hipEvent_t)lfg_lane_event(device, 20 + u); if (!ld) return LF_ERR_HIP;
                HIPCHK(hipEventRecord(ld, ls[u])); HIPCHK(hipStreamWaitEvent(s, ld, 0));
            }
            par ^= 1;
        }
        lf_htrial_learn(h_ctl->ratio_hist);
        if (hdbg) {
            fprintf(stderr, "[lf]   trial bounds %u / %u sixteenths (NW / SHW); roots above 4096 rows by 16 * distance / rows, NW:", HA.trial16[0], HA.trial16[1]);
            for (int r = 0; r < 16; r++) fprintf(stderr, " %u", h_ctl->ratio_hist[0][r]);
            fprintf(stderr, "  SHW:");
            for (int r = 0; r < 16; r++) fprintf(stderr, " %u", h_ctl->ratio_hist[1][r]);
            fprintf(stderr, "\n");
        }
        HIPCHK(hipGetLastError());
        if (n_roots != HC.roots) { lf_set_error("edlib Hirschberg levels: %u roots found, %llu announced", n_roots, (unsigned long long)HC.roots); return LF_ERR_ARG; }
    }
    const int N = n + (int)n_h;
    HIPCHK(hipEventRecord(bd[0], s));                      /* Hirschberg levels: e0 .. bd[0]; binning: bd[0] .. eb */

    const unsigned gb = (unsigned)((N + 255) / 256);
    HIPCHK(hipMemsetAsync(d_bins, 0, (size_t)LF_NBINS * 4, s));
    hipLaunchKernelGGL(lf_desc_keys_kernel, dim3(gb), dim3(256), 0, s, d_desc, d_opsoff, n, d_hdesc, (int)n_h, d_keys, d_ops, d_ed, d_end, d_len);
    const unsigned gc = (unsigned)((N + 1024 * LF_BIN_ITEMS - 1) / (1024 * LF_BIN_ITEMS));
    hipLaunchKernelGGL(lf_desc_count_kernel<false>, dim3(gc), dim3(1024), 0, s, (const uint32_t *)d_keys, N, d_bins, (uint32_t *)nullptr, (uint32_t *)nullptr);
    hipLaunchKernelGGL(lf_desc_binscan_kernel, dim3(1), dim3(1024), 0, s, d_bins);
    hipLaunchKernelGGL(lf_desc_count_kernel<true>, dim3(gc), dim3(1024), 0, s, (const uint32_t *)d_keys, N, d_bins, d_keys2, d_vals2);
    hipLaunchKernelGGL(lf_desc_segments_kernel, dim3(1), dim3(256), 0, s, d_keys2, N, d_tab);
    /* (A fused forward + traceback kernel for the problems of at most two query blocks and 127 columns -- rows in LDS, paths walked by the same wavefront --
     * was built and measured in round 5: 14 KB of LDS and 112 registers left its forward pass three wavefronts per SIMD, the step's alignment kernels took 24.3
     * instead of 21.1 ms, profiles/r05_search/.  It is not in the tree; the binning still gives such problems segments of their own, block field 65 ..) */
    const int fused_small = 0;
    hipLaunchKernelGGL(lf_desc_entries_kernel, dim3(gb), dim3(256), 0, s, d_keys2, d_vals2, d_desc, n, d_hdesc, d_tab, N, d_ent, fused_small);
    { lf_scan_u64 f; f.p = d_ent; const int src = lf_scan_excl(device, 4, s, f, d_base, (size_t)N); if (src != LF_OK) return src; }
    hipLaunchKernelGGL(lf_desc_build_kernel, dim3(gb), dim3(256), 0, s, d_keys2, d_vals2, d_desc, n, d_hdesc, d_opsoff, d_hopsoff, d_tab, d_base, N, D->pac ? 1 : 0, d_probs, d_waves, d_misc);
    /* (into PINNED memory: an asynchronous copy into pageable memory -- a stack variable -- makes the runtime wait for the stream inside the
     * call, spinning, instead of in lf_stream_wait) */
    uint64_t *tail = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(h_tab) + ((sizeof(lf_rseg_tab) + 15) & ~(size_t)15));
    HIPCHK(hipMemcpyAsync(h_tab, d_tab, sizeof(lf_rseg_tab), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&tail[0], d_base + (N - 1), 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&tail[1], d_ent + (N - 1), 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&tail[2], d_misc, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const uint64_t aux_total = tail[2];
    const uint64_t hist_entries = tail[0] + tail[1];
    lf_hist_t *d_hist = DSLOT(lf_hist_t, 2, hist_entries * sizeof(lf_hist_t) + 64);
    uint64_t *d_aux = (uint64_t *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 9, aux_total * 8 + 64);
    if (!d_hist || !d_aux) return LF_ERR_NOMEM;
#undef DSLOT
    int cstart[LF_NCLASS + 2]; for (int k = 0; k < LF_NCLASS + 2; k++) cstart[k] = h_tab->cstart[k];
    const int nw_nw = h_tab->n_waves[0], nw_shw = h_tab->n_waves[1];
    HIPCHK(hipEventRecord(eb, s));
    auto cnt = [&](int c) { return cstart[c + 1] - cstart[c]; };
    for (int k = 0; k < LF_NCLASS; k++) if (cnt(k) > 0 && cs[k] != s) HIPCHK(hipStreamWaitEvent(cs[k], eb, 0));
    lf_seqs S; S.q = D->d_q; S.t = D->d_t; S.pac = D->d_pac;
    /* largest problems first: their few long-running waves start while the rest fills the chip */
    if (cnt(3) > 0) hipLaunchKernelGGL((lf_edlib_sweep_kernel<64, 8, 2, true>), dim3((unsigned)cnt(3)), dim3(64), 0, cs[3], d_probs + cstart[3], cnt(3), S, d_hist, d_aux, d_ops, d_ed, d_end, d_len);
    if (cnt(2) > 0) hipLaunchKernelGGL((lf_edlib_sweep_kernel<64, 4, 4, true>), dim3((unsigned)cnt(2)), dim3(64), 0, cs[2], d_probs + cstart[2], cnt(2), S, d_hist, d_aux, d_ops, d_ed, d_end, d_len);
    if (cnt(0) > 0) hipLaunchKernelGGL(lf_edlib_generic_kernel, dim3((unsigned)((cnt(0) + 63) / 64)), dim3(64), 0, cs[0], d_probs + cstart[0], cnt(0), S, d_hist, d_aux, d_ops, d_ed, d_end, d_len);
    if (cnt(1) > 0) {
        lf_rsw_args RA;
        RA.probs = d_probs; RA.waves = d_waves; RA.qlo = D->d_planes; RA.qhi = D->d_planes + D->q_words; RA.qvalid = D->d_planes + 2 * D->q_words; RA.q_words = D->q_words;
        RA.pac = D->d_pac; RA.pac_syms = D->pac_syms; RA.ckpt = d_hist; RA.out_ed = d_ed; RA.out_end = d_end;
        /* segments t = mode * 128 + block field; the small problems' segments (block field 65 .. 64 + LF_SMALL_NB) close each mode's range:
         * their waves go to the fused kernel (rows in LDS, paths walked by the same wavefront), everything else to forward + traceback */
        const int w_shw0 = h_tab->w0[128], w_end = h_tab->w0[256];
        const int p_shw0 = h_tab->lo[128], p_end = h_tab->lo[256];
        (void)nw_nw; (void)nw_shw;
        RA.ops = d_ops; RA.out_len = d_len;
        RA.rev = true;                                       /* the long problems of a launch first */
        /* The two modes' problems are independent: NW (the pieces between anchors) sweeps and walks its paths on the class stream, SHW (the pieces in front
         * of the first / behind the last anchor) on a stream of its own.  Each mode's traceback ends with ONE long path (1.3 and 0.9 ms in a chunk of
         * 1 600 or 6 250 reads alike, profiles/r05_chain/): back to back they were a quarter of a small chunk's launch chain.  (One stream when the
         * classes are serialized for the exclusive timings, and for the fused small-problem kernel's A / B.) */
        hipStream_t cb = cs[1];
        hipEvent_t shw_done = nullptr;
        if (!serial_classes) {
            cb = (hipStream_t)lfg_lane_stream(device, 6); shw_done = (hipEvent_t)lfg_lane_event(device, 9);
            if (!cb || !shw_done) return LF_ERR_HIP;
            HIPCHK(hipStreamWaitEvent(cb, eb, 0));
        }
        HIPCHK(hipEventRecord(bd[1], cs[1]));
        RA.wave0 = 0; RA.n_waves = w_shw0; lf_rsweep_launch(cs[1], false, RA);
        RA.wave0 = w_shw0; RA.n_waves = w_end - w_shw0; lf_rsweep_launch(cb, true, RA);
        HIPCHK(hipEventRecord(bd[2], cs[1]));
        {
            /* eight columns per replayed part of a 32-step tile (four parts, 8 KiB of LDS per wavefront; parts of 4 and of 16 columns measured 12.0 and 10.1 ms
             * against 9.4, round 4); the NW paths on the class stream, the SHW paths on theirs */
            const int r0[2] = { cstart[1], p_shw0 }, r1[2] = { p_shw0, p_end };
            for (int q = 0; q < 2; q++) {
                const int np = r1[q] - r0[q];
                if (np <= 0) continue;
                hipStream_t ts = q == 0 ? cs[1] : cb;
                hipLaunchKernelGGL(lf_edlib_tb_kernel<8>, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, ts, d_probs + r0[q], np, S, D->pac_syms, d_hist, d_ops, d_end, d_len, RA.rev);
            }
        }
        if (cb != cs[1]) { HIPCHK(hipEventRecord(shw_done, cb)); HIPCHK(hipStreamWaitEvent(cs[1], shw_done, 0)); }
        HIPCHK(hipEventRecord(bd[3], cs[1]));
    }
    for (int k = 0; k < LF_NCLASS; k++) if (cnt(k) > 0 && cs[k] != s) { HIPCHK(hipEventRecord(cdone[k], cs[k])); HIPCHK(hipStreamWaitEvent(s, cdone[k], 0)); }
    if (n_roots) lf_hirsch_launch_stitch(s, HA, n_roots);      /* the roots' pieces move together once their leaves have paths */
    HIPCHK(hipEventRecord(e1, s));
    if (ed) {
        HIPCHK(hipMemcpyAsync(ed, d_ed, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(endloc, d_end, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(ops_len, d_len, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    }
    if (ops) HIPCHK(hipMemcpyAsync(ops, d_ops, D->ops_total, hipMemcpyDeviceToHost, s));
    if (dev_desc && !ed && !ops) {
        HIPCHK(hipGetLastError());
        t_pend.on = true; t_pend.class1 = cnt(1) > 0; t_pend.e0 = e0; t_pend.e1 = e1; t_pend.eb = eb; for (int k = 0; k < 4; k++) t_pend.bd[k] = bd[k];
        return LF_OK;
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (ms) HIPCHK(hipEventElapsedTime(ms, e0, e1));
    if (cnt(1) > 0) { HIPCHK(hipEventElapsedTime(&t_breakdown[0], bd[1], bd[2])); HIPCHK(hipEventElapsedTime(&t_breakdown[1], bd[2], bd[3])); }
    HIPCHK(hipEventElapsedTime(&t_breakdown[2], e0, bd[0])); HIPCHK(hipEventElapsedTime(&t_breakdown[3], bd[0], eb));
    /* (a device-planned round reports a missing Hirschberg split through lf_walk_emit_kernel: ed == -2 makes the job rare) */
    if (ed && n_roots) for (int i = 0; i < n; i++) if (ed[i] == -2) { lf_set_error("edlib kernel: no Hirschberg split row for descriptor %d (n %u, m %u)", i, D->d ? D->d[i].n : 0, D->d ? D->d[i].m : 0); return LF_ERR_HIP; }
    return LF_OK;
}

/* ---- trial bounds of the Hirschberg roots (lf_hirsch.h): chosen from what the process has seen.  g_hratio[mode][r] counts the roots above 4096 rows whose
 * distance was r / 16 .. (r + 1) / 16 of their rows (halved now and then, so that a new kind of input takes over); a mode's bound is the first sixteenth
 * below which nine roots in ten ended, no trial at all when that is above 5 / 16 (the band of such a bound covers most of the matrix and the sweep's length --
 * its latency -- is the same with or without it).  Before anything was seen: NW 4 / 16, SHW none. ---- */
static std::atomic<uint32_t> g_hratio[2][16];
void lf_htrial_learn(const uint32_t hist[2][16])
{
    for (int k = 0; k < 2; k++) {
        uint32_t tot = 0;
        for (int r = 0; r < 16; r++) if (hist[k][r]) tot += g_hratio[k][r].fetch_add(hist[k][r], std::memory_order_relaxed) + hist[k][r];
        if (tot > (1u << 20)) for (int r = 0; r < 16; r++) g_hratio[k][r].store(g_hratio[k][r].load(std::memory_order_relaxed) / 2, std::memory_order_relaxed);
    }
}
void lf_htrial_pick(uint32_t trial16[2])
{
    for (int k = 0; k < 2; k++) {
        uint32_t c[16], tot = 0;
        for (int r = 0; r < 16; r++) { c[r] = g_hratio[k][r].load(std::memory_order_relaxed); tot += c[r]; }
        if (tot < 8) { trial16[k] = k == 0 ? 4u : 0u; continue; }
        uint32_t cum = 0; int r = 0;
        for (; r < 16; r++) { cum += c[r]; if ((uint64_t)cum * 10 >= (uint64_t)tot * 9) break; }
        trial16[k] = r + 1 > 5 ? 0u : (uint32_t)(r + 1);
    }
}

/* what the Hirschberg levels need to know about a batch before anything runs (buffer bounds): counted where the descriptors are made */
static lf_hcount_t count_hroots(int n, const lf_aln_desc_t *d)
{
    lf_hcount_t c; memset(&c, 0, sizeof c);
    for (int i = 0; i < n; i++) if (d[i].n && d[i].m && !lf_leaf(d[i].n, d[i].m)) { c.roots++; c.cap += lf_hroot_cap(d[i].n, d[i].m); c.sum_n += d[i].n; c.sum_m += d[i].m; }
    return c;
}

/* the pipeline's sequences: the read batch lfg_seed left in HBM (bytes + bit planes) and the 2-bit reference */
static int pipeline_src(const struct lf_index *ix, lf_desc_src *D)
{
    lf_dev_state *st = (lf_dev_state *)ix->dev;
    if (!st) { lf_set_error("index is not on a device"); return LF_ERR_NO_DEVICE; }
    D->d_q = (const unsigned char *)lfg_dev_slot(ix->device, LF_DS_SEED0 + 0, 0);
    D->d_planes = (const uint64_t *)lfg_dev_slot(ix->device, LF_DS_SEED0 + 14, 0);
    D->q_words = (int64_t)lfg_lane_value(ix->device, 0);
    D->d_t = nullptr; D->d_pac = st->view.pac; D->pac_syms = ix->l_pac; D->pac = true;
    if (!D->d_q || !D->d_planes || D->q_words < 2) { lf_set_error("no resident read batch"); return LF_ERR_ARG; }
    return LF_OK;
}

extern "C" int lfg_edlib_desc(const struct lf_index *ix, int n, const lf_aln_desc_t *d, const uint64_t *ops_off, uint64_t ops_total,
                              int32_t *ed, int32_t *endloc, uint8_t *ops, uint32_t *ops_len, int ops_slot, void **ops_dev, void **desc_dev, float *ms)
{
    lf_desc_src D;
    const int rc = pipeline_src(ix, &D);
    if (rc != LF_OK) return rc;
    D.d = d; D.ops_off = ops_off; D.ops_total = ops_total;
    D.hc = count_hroots(n, d);
    return run_edlib_desc_gpu(ix->device, n, &D, ed, endloc, ops, ops_len, ops_slot, ops_dev, desc_dev, ms);
}

extern "C" int lfg_edlib_desc_dev(const struct lf_index *ix, int n, const void *d_desc, const void *d_opsoff, uint64_t ops_total, const lf_hcount_t *hc, int ops_slot,
                                  void **ops_dev, void **ed_dev, void **end_dev, void **len_dev, float *ms)
{
    lf_desc_src D;
    const int rc0 = pipeline_src(ix, &D);
    if (rc0 != LF_OK) return rc0;
    D.d = nullptr; D.ops_off = nullptr; D.ops_total = ops_total;
    D.hc = *hc;
    void *res[3] = { nullptr, nullptr, nullptr };
    const int rc = run_edlib_desc_gpu(ix->device, n, &D, nullptr, nullptr, nullptr, nullptr, ops_slot, ops_dev, nullptr, ms,
                                      (const lf_aln_desc_t *)d_desc, (const uint64_t *)d_opsoff, res);
    if (ed_dev) *ed_dev = res[0];
    if (end_dev) *end_dev = res[1];
    if (len_dev) *len_dev = res[2];
    return rc;
}

/* problems given as byte strings in host memory (the stage API lf_edlib_batch and the edlibAlign drop-in): the strings are
 * uploaded once, the queries become bit planes and the targets 2-bit codes like the pipeline's, and the problems are
 * described exactly like the pipeline's requests -- the same binning, kernels and Hirschberg levels.  A target that holds
 * anything but upper-case ACGT keeps its bytes and goes through the generic kernel (exact byte compare, any alphabet). */
extern "C" int lfg_edlib(int device, int n, const char *q, const uint64_t *qoff, const char *t, const uint64_t *toff,
                         const uint8_t *mode, int32_t *ed, int32_t *endloc, uint8_t *ops, uint32_t *ops_len, float *ms)
{
    if (ms) *ms = 0;
    if (n == 0) return LF_OK;
    if (lfg_device_count() <= device) { lf_set_error("no gfx950 device %d visible (no CPU path)", device); return LF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)lfg_lane_stream(device, 1);
    if (!s) return LF_ERR_HIP;
    const uint64_t qbytes = qoff[n], tbytes = toff[n], qw = lf_plane_words(qbytes);
    unsigned char *d_q = (unsigned char *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 5, qbytes + 64), *d_t = (unsigned char *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 6, tbytes + 64);
    uint64_t *d_planes = (uint64_t *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 7, 3 * qw * 8);
    uint8_t *d_tpac = (uint8_t *)lfg_dev_slot(device, LF_DS_HIRSCH0 + 8, tbytes / 4 + 128);
    if (!d_q || !d_t || !d_planes || !d_tpac) return LF_ERR_NOMEM;
    std::vector<lf_aln_desc_t> dd((size_t)n); std::vector<uint64_t> oo((size_t)n);
    for (int i = 0; i < n; i++) {
        lf_aln_desc_t &x = dd[(size_t)i]; memset(&x, 0, sizeof x);
        x.qstart = (int64_t)qoff[i]; x.tstart = (int64_t)toff[i]; x.n = (uint32_t)(qoff[i + 1] - qoff[i]); x.m = (uint32_t)(toff[i + 1] - toff[i]);
        x.mode = mode ? mode[i] : 0;
        for (uint64_t k = toff[i]; k < toff[i + 1]; k++) { const char c = t[k]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') { x.pad[0] = 1; break; } }
        oo[(size_t)i] = qoff[i] + toff[i];
    }
    HIPCHK(hipMemcpyAsync(d_q, q, qbytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_t, t, tbytes, hipMemcpyHostToDevice, s));
    lf_rsweep_pack_planes(s, d_q, qbytes, d_planes, qw);
    lf_rsweep_pack_pac(s, d_t, tbytes, d_tpac);
    lf_desc_src D;
    D.d = dd.data(); D.ops_off = oo.data(); D.ops_total = qbytes + tbytes;
    D.d_q = d_q; D.d_t = d_t; D.d_pac = d_tpac; D.pac_syms = (int64_t)tbytes > 0 ? (int64_t)tbytes : 1; D.d_planes = d_planes; D.q_words = (int64_t)qw; D.pac = false;
    D.hc = count_hroots(n, dd.data());
    return run_edlib_desc_gpu(device, n, &D, ed, endloc, ops, ops_len, 0, nullptr, nullptr, ms);
}

/* ------------------------------------------------------------------------------------------------
 * ksw_extend2 (lib/bwa/ksw.c:380-478), clip matrix +2/-16/N=0 (src/LordFAST.cpp:82-85,178-187).
 * Rare branch of the reference (clip / split tests), but its problems are long (band 40 / 100, up to thousands of rows).
 * One wavefront per problem; rows stay sequential, the band of a row (<= 2w + 1 columns) is processed 64 columns at a
 * time:
 *   - M(i,j) and E(i,j) depend on the previous row only: one lane per column;
 *   - F runs along the row:  f(j+1) = max(f(j) - e_ins, max(M(j) - oe_ins, 0)).  With B(l) = t(l) + (l+1) e_ins this is
 *     f(k) = max(carry, max_{l<k} B(l)) - k e_ins: one exclusive max-scan (DPP) per tile;
 *   - the row maximum with the reference's tie rule (the LAST column attaining it: `mj = m > h ? mj : j`) is a wave
 *     max of (h << 32 | j);
 *   - H(i,j) is stored one column to the right (the reference's eh[j].h = h1 trick): a wave_shr:1 move.
 * The band trimming, z-drop and maximum bookkeeping are the reference's scalar code on wave-uniform values.
 * ---------------------------------------------------------------------------------------------- */
/* signed inclusive max-scan / 64-bit max-reduction over the wavefront by DPP row shifts and broadcasts (no LDS) */
__device__ __forceinline__ int lf_wave_incl_max_i32(int v)
{
    int x = v;
    const int lo = INT_MIN;
    x = max(x, __builtin_amdgcn_update_dpp(lo, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(lo, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(lo, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(lo, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(lo, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(lo, x, 0x143, 0xc, 0xf, false));
    return x;
}
struct lf_ksw_prob { uint64_t qoff, toff, ws_off; int32_t qlen, tlen, o_del, e_del, o_ins, e_ins, w, zdrop, h0, id; };

/* the band width ksw_extend2 really uses (lib/bwa/ksw.c:409-416) */
__host__ __device__ __forceinline__ int lf_ksw_band(int qlen, int o_del, int e_del, int o_ins, int e_ins, int w)
{
    int max_ins = (int)((double)(qlen * 2 - o_ins) / e_ins + 1.);
    if (max_ins < 1) max_ins = 1;
    if (w > max_ins) w = max_ins;
    int max_del = (int)((double)(qlen * 2 - o_del) / e_del + 1.);
    if (max_del < 1) max_del = 1;
    if (w > max_del) w = max_del;
    return w;
}
#define LF_KSW_MW_MAXW 127       /* lf_ksw_mw_kernel: a row's band (<= 2 w + 1 columns) fits the 256 threads of its workgroup and a 256-entry ring */

template <bool LDS>
__global__ void __launch_bounds__(64)
lf_ksw_kernel(const lf_ksw_prob *__restrict__ probs, int n_probs, const uint8_t *__restrict__ qs, const uint8_t *__restrict__ ts,
              int32_t *__restrict__ ws, int32_t *__restrict__ out_score, int32_t *__restrict__ out_qle, int32_t *__restrict__ out_tle, int lds_q, int wide_only)
{
    extern __shared__ __attribute__((aligned(16))) int32_t s_he[];
    const int gid = blockIdx.x, lane = threadIdx.x;
    if (gid >= n_probs) return;
    const lf_ksw_prob pr = probs[gid];
    if (LDS != (pr.qlen <= lds_q)) return;            /* the other instantiation serves this problem */
    if (wide_only && lf_ksw_band(pr.qlen, pr.o_del, pr.e_del, pr.o_ins, pr.e_ins, pr.w) <= LF_KSW_MW_MAXW) return;      /* lf_ksw_mw_kernel does */
    const uint8_t *q = qs + pr.qoff, *t = ts + pr.toff;
    const int qlen = pr.qlen, tlen = pr.tlen, o_del = pr.o_del, e_del = pr.e_del, o_ins = pr.o_ins, e_ins = pr.e_ins;
    const int zdrop = pr.zdrop, h0 = pr.h0;
    int w = pr.w;
    int32_t *H, *E;
    if (LDS) { H = s_he; E = s_he + qlen + 2; } else { H = ws + pr.ws_off; E = H + qlen + 2; }
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    /* first row (lib/bwa/ksw.c:404-407): h0, h0 - oe_ins, then -e_ins per column while positive */
    for (int j = lane; j <= qlen + 1; j += 64) {
        int v = 0;
        if (j == 0) v = h0;
        else if (j <= qlen && h0 > oe_ins) { const long long x = (long long)h0 - oe_ins - (long long)(j - 1) * e_ins; v = (j == 1 || x + e_ins > e_ins) ? (int)(x > 0 ? x : 0) : 0; }
        H[j] = v; E[j] = 0;
    }
    __syncthreads();
    w = lf_ksw_band(qlen, o_del, e_del, o_ins, e_ins, w);
    int mx = h0, max_i = -1, max_j = -1, beg = 0, end = qlen;
    for (int i = 0; i < tlen; ++i) {
        const int tc = t[i];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int h1;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        int f = 0, m = 0, mj = -1;
        unsigned long long best = 0;                     /* this lane's (h << 32 | j) maximum over the row's tiles */
        for (int base = beg; base < end; base += 64) {
            const int j = base + lane;
            const bool act = j < end;
            const int cnt = end - base < 64 ? end - base : 64;
            int M = 0, e = 0, tins = 0;
            if (act) {
                M = H[j]; e = E[j];
                const int qc = q[j];
                const int sc = (tc > 3 || qc > 3) ? 0 : (tc == qc ? 2 : -16);
                M = M ? M + sc : 0;
                tins = M - oe_ins; if (tins < 0) tins = 0;
            }
            /* f entering column base + k: max(carry, max_{l<k} B(l)) - k e_ins */
            const int Bv = act ? tins + (lane + 1) * e_ins : INT_MIN / 2;
            const int incl = lf_wave_incl_max_i32(Bv);
            int G = __builtin_amdgcn_update_dpp(INT_MIN, incl, 0x138, 0xf, 0xf, false);      /* exclusive: lane l takes lanes < l */
            G = G > f ? G : f;
            const int fin = G - lane * e_ins;
            int h = M > e ? M : e;
            h = h > fin ? h : fin;
            if (!act) h = 0;
            /* H(i, j) goes to slot j + 1 of the next row: slot j takes the h of column j - 1 (h1 for the first column) */
            int hleft = lf_wave_shr1(h); if (lane == 0) hleft = h1;
            if (act) {
                H[j] = hleft;
                int tt = M - oe_del; if (tt < 0) tt = 0;
                int e2 = e - e_del; if (e2 < tt) e2 = tt;
                E[j] = e2;
            }
            /* row maximum, last column on ties: kept per lane over the tiles (a lane's columns ascend), reduced ONCE per row */
            { const unsigned long long key = act ? (((unsigned long long)(unsigned)h << 32) | (unsigned)j) : 0ull; best = key > best ? key : best; }
            /* carries into the next tile */
            const int last = __builtin_amdgcn_readfirstlane(cnt - 1);
            h1 = __builtin_amdgcn_readlane(h, last);
            int fout = fin - e_ins; if (fout < tins) fout = tins;
            f = __builtin_amdgcn_readlane(fout, last);
        }
        if (beg < end) {
            /* wave maximum of (h << 32 | j): the high words first, then the low words of the lanes that tie on them (DPP, no LDS) */
            const uint32_t bh = (uint32_t)(best >> 32), hmax = lf_wave_max_u32(bh);
            const uint32_t bl = bh == hmax ? (uint32_t)best : 0u, lmax = lf_wave_max_u32(bl);
            m = (int)hmax; mj = (int)lmax;
        }
        if (lane == 0) { H[end] = h1; E[end] = 0; }
        __syncthreads();
        if (m == 0) break;
        if (m > mx) { mx = m; max_i = i; max_j = mj; }
        else if (zdrop > 0) {
            if (i - max_i > mj - max_j) { if (mx - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (mx - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        /* trim the band: first / last column of [beg, end] that is not all zero (lib/bwa/ksw.c:462-465) */
        int nb = end;
        for (int base = beg; base < end; base += 64) {
            const int j = base + lane;
            const unsigned long long nz = lf_ballot(j < end && (H[j] != 0 || E[j] != 0));
            if (nz) { nb = base + (__ffsll((long long)nz) - 1); break; }
        }
        int ne = nb - 1;
        for (int top = end; top >= nb; top -= 64) {
            const int j = top - lane;
            const unsigned long long nz = lf_ballot(j >= nb && (H[j] != 0 || E[j] != 0));
            if (nz) { ne = top - (__ffsll((long long)nz) - 1); break; }
        }
        beg = nb;
        end = ne + 2 < qlen ? ne + 2 : qlen;
        __syncthreads();
    }
    if (lane == 0) { out_score[pr.id] = mx; out_qle[pr.id] = max_j + 1; out_tle[pr.id] = max_i + 1; }
}

/* ------------------------------------------------------------------------------------------------
 * ksw_extend2 with the row spread over the FOUR wavefronts of a workgroup: the problems of the clip tests are few and long
 * (thousands of dependent rows), so what counts is the time of one row.  lf_ksw_kernel walks a row's band tile by tile (up
 * to four 64-column tiles, each a round trip to LDS, a DPP scan, a carry to the next tile) and spends ~1.3 us per row; here
 * thread k owns column beg + k of the row (band <= 2 w + 1 <= 255 columns: w <= LF_KSW_MW_MAXW, all of lordFAST's calls):
 *   phase 1   M, E, the insertion candidates; max-scan inside the wavefront; the tile's aggregate to LDS        | barrier
 *   phase 2   f entering the tile from the aggregates of the tiles to its left (carry(t + 1) = max(carry(t), A(t)) - 64 e_ins),
 *             h, the new E; H(i, j) is stored one slot to the right by its own thread; per wavefront: the row maximum with the
 *             reference's tie rule, and the first / last slot of the NEXT row's band that is not all zero (from h and e in
 *             registers: slot p holds h(p - 1) and e(p)), to LDS                                              | barrier
 *   phase 3   every thread combines the four wavefronts' results: z-drop, maximum bookkeeping and band trimming are the
 *             reference's scalar code on workgroup-uniform values
 * Two barriers per row, no second pass over the row for the trimming, the read's bases in LDS.  Same arithmetic, same results.
 * Measured (profiles/r04_ksw/): 1.0 us per row alone (lf_ksw_kernel: 1.5 - 1.9 with the reference's w = 100), config C4's ksw
 * 162 -> 98 ms per step.  A one-wavefront version of the same ring design (four tiles in registers, all loads up front, no
 * barrier) issues 2.8 x fewer instructions per row but takes 1.65 us per row: the row is a chain of dependent steps, not an
 * instruction count, and C4's launches of a few hundred problems each are bound by that chain too.
 * ---------------------------------------------------------------------------------------------- */
template <bool LDS>
__global__ void __launch_bounds__(256)
lf_ksw_mw_kernel(const lf_ksw_prob *__restrict__ probs, int n_probs, const uint8_t *__restrict__ qs, const uint8_t *__restrict__ ts,
                 int32_t *__restrict__ out_score, int32_t *__restrict__ out_qle, int32_t *__restrict__ out_tle, int lds_q, int r4_on)
{
    extern __shared__ __attribute__((aligned(16))) int32_t s_he[];      /* H ring, E ring (256 slots each), then the read's bases (LDS instantiation) */
    __shared__ int s_A[4], s_mn[4], s_mxp[4];
    __shared__ uint32_t s_bh[4], s_bl[4];
    const int gid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (gid >= n_probs) return;
    const lf_ksw_prob pr = probs[gid];
    if (LDS != (pr.qlen <= lds_q)) return;            /* the other instantiation serves this problem */
    const int qlen = pr.qlen, tlen = pr.tlen, o_del = pr.o_del, e_del = pr.e_del, o_ins = pr.o_ins, e_ins = pr.e_ins;
    const int zdrop = pr.zdrop, h0 = pr.h0;
    const int w = lf_ksw_band(qlen, o_del, e_del, o_ins, e_ins, pr.w);
    if (w > LF_KSW_MW_MAXW) return;                   /* lf_ksw_kernel does */
    if (r4_on && w <= 120 /* LF_KSW_R4_MAXW */) return;      /* lf_ksw_r4_kernel does */
    const uint8_t *q = qs + pr.qoff, *t = ts + pr.toff;
    /* The H / E arrays of the reference (qlen + 2 entries) are only ever touched inside the row's band [beg, end], beg never
     * decreases and end - beg <= 2 w + 1 <= 255: entry j lives in slot j & 255 of a ring, 2 KiB per problem whatever the read's
     * length (the full arrays took 48 KiB of LDS per workgroup -- two problems per CU -- or went through HBM for reads above
     * 6 000 bases: a round trip to memory per row).  Rows write [beg, end] without gaps, so the entries above the highest `end`
     * so far still hold the first row's values (lib/bwa/ksw.c:404-407), which are a closed form: no initialisation pass. */
    int32_t *H = s_he, *E = s_he + 256;
    const uint8_t *Q = q;
    if (LDS) {
        uint8_t *sq = reinterpret_cast<uint8_t *>(s_he + 512);
        for (int j = tid; j < qlen; j += 256) sq[j] = q[j];
        Q = sq;
    }
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    auto h_first = [&](int j) -> int {               /* h0, h0 - oe_ins, then -e_ins per column while positive */
        if (j == 0) return h0;
        if (j > qlen || h0 <= oe_ins) return 0;
        const long long x = (long long)h0 - oe_ins - (long long)(j - 1) * e_ins;
        return (int)(x > 0 ? x : 0);
    };
    int hiw = -1;                                     /* entries [0, hiw] have been written */
    __syncthreads();
    int mx = h0, max_i = -1, max_j = -1, beg = 0, end = qlen;
    int tc = tlen > 0 ? (int)t[0] : 0;
    for (int i = 0; i < tlen; ++i) {
        const int tc_next = i + 1 < tlen ? (int)t[i + 1] : 0;      /* requested a row ahead */
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int h1;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        const int j = beg + tid;
        const bool act = j < end;
        /* ---- phase 1 ---- */
        int M = 0, e = 0, tins = 0;
        if (act) {
            M = H[j & 255]; e = E[j & 255];
            if (j > hiw) { M = h_first(j); e = 0; }   /* (the columns that enter the band: one or two per row) */
            const int qc = Q[j];
            const int sc = (tc > 3 || qc > 3) ? 0 : (tc == qc ? 2 : -16);
            M = M ? M + sc : 0;
            tins = M - oe_ins; if (tins < 0) tins = 0;
        }
        const int Bv = act ? tins + (lane + 1) * e_ins : INT_MIN / 2;
        const int incl = lf_wave_incl_max_i32(Bv);
        const int excl = __builtin_amdgcn_update_dpp(INT_MIN, incl, 0x138, 0xf, 0xf, false);      /* lane l takes lanes < l */
        if (lane == 63) s_A[wv] = incl;
        __syncthreads();
        /* ---- phase 2 ---- */
        int carry = 0;                                   /* f entering the row's first column */
#pragma unroll
        for (int u = 0; u < 3; u++) if (u < wv) { const int a = s_A[u]; carry = (a > carry ? a : carry) - 64 * e_ins; }
        const int G = excl > carry ? excl : carry;
        const int fin = G - lane * e_ins;
        int h = M > e ? M : e;
        h = h > fin ? h : fin;
        if (!act) h = 0;
        int e2 = 0;
        if (act) {
            H[(j + 1) & 255] = h;                        /* H(i, j) goes to entry j + 1 of the next row (the reference's eh[j].h = h1 trick) */
            int tt = M - oe_del; if (tt < 0) tt = 0;
            e2 = e - e_del; if (e2 < tt) e2 = tt;
            E[j & 255] = e2;
            if (j + 1 == end) E[end & 255] = 0;
        }
        if (tid == 0) { H[beg & 255] = h1; if (beg >= end) E[end & 255] = 0; }
        hiw = end > hiw ? end : hiw;
        {
            /* row maximum, the LAST column on ties (`mj = m > h ? mj : j`): the high words first, then the columns of the lanes that tie */
            const uint32_t bh = act ? (uint32_t)h : 0u, hmax = lf_wave_max_u32(bh);
            const uint32_t bl = (act && bh == hmax) ? (uint32_t)j : 0u, lmax = lf_wave_max_u32(bl);
            /* slots of the next row that are not all zero: slot j + 1 through h, slot j through e2 */
            const unsigned long long ba = lf_ballot(act && h != 0), bb = lf_ballot(act && e2 != 0);
            if (lane == 0) {
                const int base = beg + 64 * wv;
                int mn = INT_MAX, mxp = INT_MIN;
                if (ba) { mn = base + (__ffsll((long long)ba) - 1) + 1; mxp = base + (63 - __clzll((long long)ba)) + 1; }
                if (bb) { const int a0 = base + (__ffsll((long long)bb) - 1), a1 = base + (63 - __clzll((long long)bb)); mn = a0 < mn ? a0 : mn; mxp = a1 > mxp ? a1 : mxp; }
                s_bh[wv] = hmax; s_bl[wv] = lmax; s_mn[wv] = mn; s_mxp[wv] = mxp;
            }
        }
        __syncthreads();
        /* ---- phase 3: workgroup-uniform ---- */
        uint32_t hm = 0, lm = 0; int mn = h1 != 0 ? beg : INT_MAX, mxp = h1 != 0 ? beg : INT_MIN;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t a = s_bh[u], b = s_bl[u];
            if (a > hm) { hm = a; lm = b; } else if (a == hm && b > lm) lm = b;
            const int x = s_mn[u], y = s_mxp[u];
            mn = x < mn ? x : mn; mxp = y > mxp ? y : mxp;
        }
        const int m = beg < end ? (int)hm : 0, mj = beg < end ? (int)lm : -1;
        tc = tc_next;
        if (m == 0) break;
        if (m > mx) { mx = m; max_i = i; max_j = mj; }
        else if (zdrop > 0) {
            if (i - max_i > mj - max_j) { if (mx - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (mx - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        /* trim the band: first / last slot of [beg, end] that is not all zero (lib/bwa/ksw.c:462-465) */
        const int nb = mn == INT_MAX ? end : mn, ne = mxp == INT_MIN ? end - 1 : mxp;
        beg = nb;
        end = ne + 2 < qlen ? ne + 2 : qlen;
    }
    if (tid == 0) { out_score[pr.id] = mx; out_qle[pr.id] = max_j + 1; out_tle[pr.id] = max_i + 1; }
}

/* ------------------------------------------------------------------------------------------------
 * ksw_extend2 with the row's band IN REGISTERS: one wavefront per problem, lane l owns the four consecutive columns base + 4 l ... of a
 * 256-column window that holds the band [beg, end] (w <= LF_KSW_R4_MAXW).  What a clip test costs is the time of ONE row times thousands
 * of dependent rows, and a row of lf_ksw_mw_kernel is ~300 instructions per wavefront around two workgroup barriers and three LDS round
 * trips (1.0 us).  Here a row has no LDS access and no barrier at all:
 *   - H (the reference's eh[j].h: H(i-1, j-1), the diagonal predecessor of column j) and E stay in the registers of the column's owner;
 *     "eh[j].h = h1" -- every column takes the h of its left neighbour -- is a move inside the lane and ONE DPP wave_shr for the lane's
 *     first column; columns outside [beg, end] keep their old values exactly like the reference's array does;
 *   - F: a lane folds its four columns (B(j) = max(M - oe_ins, 0) + (j + 1) e_ins, prefix maxima), ONE exclusive max-scan over the lanes'
 *     aggregates gives every lane the gap that enters it;
 *   - row maximum with the reference's tie rule (last column wins) = one wave maximum of (h << 8 | column in the window);
 *   - band trimming: a lane's first / last slot that is not all zero, one ballot, two readlanes;
 *   - the window follows the band: when the band's left edge has moved 64 columns (or the right edge nears the window's end) the
 *     registers move down by whole lanes (ds_bpermute) and the lanes that become free take the closed form of the first row
 *     (lib/bwa/ksw.c:404-407): columns above the highest `end` so far have never been written.
 * Same arithmetic, same results (tests/test_gpu_stages.py::test_ksw_golden_and_fuzz).  Wider bands: lf_ksw_mw_kernel / lf_ksw_kernel.
 * ---------------------------------------------------------------------------------------------- */
#define LF_KSW_R4_MAXW 120
__global__ void __launch_bounds__(256)
lf_ksw_r4_kernel(const lf_ksw_prob *__restrict__ probs, int n_probs, const uint8_t *__restrict__ qs, const uint8_t *__restrict__ ts,
                 int32_t *__restrict__ out_score, int32_t *__restrict__ out_qle, int32_t *__restrict__ out_tle)
{
    /* four problems per workgroup, one per wavefront (they never talk to each other): a workgroup's wavefronts go to the four SIMDs of a CU, so a
     * launch of a few hundred problems gets a SIMD per problem -- single-wavefront workgroups were packed several to a SIMD (512 problems: 2.8 us
     * per row instead of 0.8) */
    const int gid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (gid >= n_probs) return;
    const lf_ksw_prob pr = probs[gid];
    const int qlen = pr.qlen, tlen = pr.tlen, o_del = pr.o_del, e_del = pr.e_del, o_ins = pr.o_ins, e_ins = pr.e_ins;
    const int zdrop = pr.zdrop, h0 = pr.h0;
    const int w = lf_ksw_band(qlen, o_del, e_del, o_ins, e_ins, pr.w);
    if (w > LF_KSW_R4_MAXW) return;                   /* lf_ksw_mw_kernel / lf_ksw_kernel do */
    const uint8_t *q = qs + pr.qoff, *t = ts + pr.toff;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    constexpr int NEG = INT_MIN / 2;
    auto h_first = [&](int j) -> int {               /* h0, h0 - oe_ins, then -e_ins per column while positive */
        if (j == 0) return h0;
        if (j > qlen || h0 <= oe_ins) return 0;
        const long long x = (long long)h0 - oe_ins - (long long)(j - 1) * e_ins;
        return (int)(x > 0 ? x : 0);
    };
    int base = 0;                                     /* first column of the window, a multiple of 4 */
    int Hd[4], E[4], qc[4];
    auto fresh = [&]() {
#pragma unroll
        for (int k = 0; k < 4; k++) { Hd[k] = h_first(base + 4 * lane + k); E[k] = 0; }
    };
    auto load_q = [&]() {
#pragma unroll
        for (int k = 0; k < 4; k++) { const int j = base + 4 * lane + k; qc[k] = j < qlen ? (int)q[j] : 4; }
    };
    fresh(); load_q();
    int mx = h0, max_i = -1, max_j = -1, beg = 0, end = qlen;
    int tc = tlen > 0 ? (int)t[0] : 0;
    for (int i = 0; i < tlen; ++i) {
        const int tc_next = i + 1 < tlen ? (int)t[i + 1] : 0;      /* requested a row ahead */
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        /* the window follows the band (wave-uniform): slots up to end + 1 are written this row, end grows by at most two per row */
        if (beg - base >= 64 || end + 4 >= base + 256) {
            const int sl = (beg - base) >> 2;
            if (sl > 0) {
                const int src = (lane + sl) & 63;
#pragma unroll
                for (int k = 0; k < 4; k++) { Hd[k] = __shfl(Hd[k], src); E[k] = __shfl(E[k], src); }
                base += 4 * sl;
                if (lane >= 64 - sl) fresh();
                load_q();
            }
        }
        int h1;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        const int j0 = base + 4 * lane, r0 = 4 * lane;            /* the lane's first column, absolute / inside the window */
        bool act[4]; int M[4], tin[4], B[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = j0 + k;
            act[k] = j >= beg && j < end;
            const int sc = (tc > 3 || qc[k] > 3) ? 0 : (tc == qc[k] ? 2 : -16);
            int m0 = Hd[k] ? Hd[k] + sc : 0;
            M[k] = act[k] ? m0 : 0;
            int ti = M[k] - oe_ins; tin[k] = ti < 0 ? 0 : ti;
            B[k] = act[k] ? tin[k] + (r0 + k + 1) * e_ins : NEG;
        }
        /* f entering column j = max over the columns l < j of B(l), minus (j + 1 - 1) e_ins; at least the row's initial 0 decayed (never above E >= 0) */
        const int P0 = B[0], P1 = P0 > B[1] ? P0 : B[1], P2 = P1 > B[2] ? P1 : B[2], P3 = P2 > B[3] ? P2 : B[3];
        const int incl = lf_wave_incl_max_i32(P3);
        int G = __builtin_amdgcn_update_dpp(NEG, incl, 0x138, 0xf, 0xf, false);      /* exclusive: lane l takes lanes < l */
        G = G > 0 ? G : 0;
        int h[4], e2[4];
        {
            const int pre[4] = { G, G > P0 ? G : P0, G > P1 ? G : P1, G > P2 ? G : P2 };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int fin = pre[k] - (r0 + k) * e_ins;
                int hh = M[k] > E[k] ? M[k] : E[k];
                hh = hh > fin ? hh : fin;
                h[k] = act[k] ? hh : 0;
                int tt = M[k] - oe_del; tt = tt < 0 ? 0 : tt;
                int ee = E[k] - e_del; ee = ee < tt ? tt : ee;
                e2[k] = ee;
            }
        }
        /* row maximum, the LAST column on ties (`mj = m > h ? mj : j`): (h << 8 | column in the window); slots of the NEXT row that are not all
         * zero: slot j + 1 through h(j), slot j through e2(j) -- the lane's first and last */
        uint32_t key = 0; int lmn = INT_MAX, lmx = INT_MIN;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t kk = act[k] ? (((uint32_t)h[k] << 8) | (uint32_t)(r0 + k)) : 0u;
            key = kk > key ? kk : key;
            if (act[k] && e2[k] != 0) lmx = j0 + k;
            if (act[k] && h[k] != 0) lmx = j0 + k + 1;
        }
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            if (act[k] && h[k] != 0) lmn = j0 + k + 1;
            if (act[k] && e2[k] != 0) lmn = j0 + k;
        }
        const uint32_t kmax = lf_wave_max_u32(key);
        const unsigned long long nzb = lf_ballot(lmn != INT_MAX);
        int mn = h1 != 0 ? beg : INT_MAX, mxp = h1 != 0 ? beg : INT_MIN;
        if (nzb) {
            const int a = __builtin_amdgcn_readlane(lmn, __ffsll((long long)nzb) - 1), b = __builtin_amdgcn_readlane(lmx, 63 - __clzll((long long)nzb));
            mn = a < mn ? a : mn; mxp = b > mxp ? b : mxp;
        }
        /* the next row's arrays: column j takes the h of column j - 1 (column beg: h1), E(j) the new e; E(end) = 0; everything else stays */
        {
            int hleft0 = lf_wave_shr1((uint32_t)h[3]);                       /* the left neighbour lane's last column */
            const int hl[4] = { hleft0, h[0], h[1], h[2] };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = j0 + k;
                Hd[k] = j == beg ? h1 : ((j > beg && j <= end) ? hl[k] : Hd[k]);
                E[k] = act[k] ? e2[k] : (j == end ? 0 : E[k]);
            }
        }
        const int m = beg < end ? (int)(kmax >> 8) : 0, mj = beg < end ? base + (int)(kmax & 255u) : -1;
        tc = tc_next;
        if (m == 0) break;
        if (m > mx) { mx = m; max_i = i; max_j = mj; }
        else if (zdrop > 0) {
            if (i - max_i > mj - max_j) { if (mx - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (mx - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        /* trim the band: first / last slot of [beg, end] that is not all zero (lib/bwa/ksw.c:462-465) */
        const int nb = mn == INT_MAX ? end : mn, ne = mxp == INT_MIN ? end - 1 : mxp;
        beg = nb;
        end = ne + 2 < qlen ? ne + 2 : qlen;
    }
    if (lane == 0) { out_score[pr.id] = mx; out_qle[pr.id] = max_j + 1; out_tle[pr.id] = max_i + 1; }
}

extern "C" int lfg_ksw(int device, int n, const uint8_t *q, const uint64_t *qoff, const uint8_t *t, const uint64_t *toff,
                       const int32_t *prm, int32_t *score, int32_t *qle, int32_t *tle, float *ms)
{
    if (ms) *ms = 0;
    if (n == 0) return LF_OK;
    if (lfg_device_count() <= device) { lf_set_error("no gfx950 device %d visible (no CPU path)", device); return LF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(device));
    std::vector<lf_ksw_prob> P((size_t)n);
    size_t ws = 0;
    for (int i = 0; i < n; i++) {
        lf_ksw_prob &p = P[i];
        p.qoff = qoff[i]; p.toff = toff[i]; p.qlen = (int32_t)(qoff[i + 1] - qoff[i]); p.tlen = (int32_t)(toff[i + 1] - toff[i]);
        p.o_del = prm[7 * i]; p.e_del = prm[7 * i + 1]; p.o_ins = prm[7 * i + 2]; p.e_ins = prm[7 * i + 3];
        p.w = prm[7 * i + 4]; p.zdrop = prm[7 * i + 5]; p.h0 = prm[7 * i + 6]; p.id = i;
        p.ws_off = ws; ws += 2 * ((size_t)p.qlen + 2);
    }
    /* persistent slots and the lane's stream: no hipMalloc / hipFree (hipFree synchronises the whole device, which
     * would stall the other chunks in flight) */
#define KSLOT(T, k, bytes) (T *)lfg_dev_slot(device, LF_DS_KSW0 + (k), (bytes))
    uint8_t *d_q = KSLOT(uint8_t, 0, qoff[n] + 16), *d_t = KSLOT(uint8_t, 1, toff[n] + 16);
    lf_ksw_prob *d_p = KSLOT(lf_ksw_prob, 2, P.size() * sizeof(lf_ksw_prob));
    int32_t *d_ws = KSLOT(int32_t, 3, ws * 4 + 16), *d_s = KSLOT(int32_t, 4, (size_t)n * 4), *d_ql = KSLOT(int32_t, 5, (size_t)n * 4), *d_tl = KSLOT(int32_t, 6, (size_t)n * 4);
#undef KSLOT
    if (!d_q || !d_t || !d_p || !d_ws || !d_s || !d_ql || !d_tl) return LF_ERR_NOMEM;
    hipStream_t s = (hipStream_t)lfg_lane_stream(device, 15);
    if (!s) return LF_ERR_HIP;
    hipEvent_t e0 = (hipEvent_t)lfg_lane_event(device, 40), e1 = (hipEvent_t)lfg_lane_event(device, 41);
    if (!e0 || !e1) return LF_ERR_HIP;
    HIPCHK(hipMemcpyAsync(d_q, q, qoff[n], hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_t, t, toff[n], hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_p, P.data(), P.size() * sizeof(lf_ksw_prob), hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(e0, s));
    int qmax = 0, n_wide = 0, n_narrow = 0, qmax_wide = 0, qmax_narrow = 0;
    for (int i = 0; i < n; i++) {
        qmax = std::max(qmax, P[i].qlen);
        const bool wide = lf_ksw_band(P[i].qlen, P[i].o_del, P[i].e_del, P[i].o_ins, P[i].e_ins, P[i].w) > LF_KSW_MW_MAXW;
        if (wide) { n_wide++; qmax_wide = std::max(qmax_wide, P[i].qlen); } else { n_narrow++; qmax_narrow = std::max(qmax_narrow, P[i].qlen); }
    }
    const int lds_q = std::min(qmax, 6000);
    /* bands of at most 2 x 120 + 1 columns (every call of the reference: w = 100): the register kernel; wider ones: the four-wavefront LDS kernel, then the general one */
    const bool r4 = true;
    bool r4_all = r4;
    if (r4) {
        for (int i = 0; i < n && r4_all; i++) r4_all = lf_ksw_band(P[i].qlen, P[i].o_del, P[i].e_del, P[i].o_ins, P[i].e_ins, P[i].w) <= LF_KSW_R4_MAXW;
        hipLaunchKernelGGL(lf_ksw_r4_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, (const lf_ksw_prob *)d_p, n, (const uint8_t *)d_q, (const uint8_t *)d_t, d_s, d_ql, d_tl);
    }
    if (n_narrow && !(r4 && r4_all)) {
        /* H / E rings, and the read's bases up to 32 KiB, in LDS */
        const int mw_q = std::min(qmax_narrow, 32768);
        hipLaunchKernelGGL(lf_ksw_mw_kernel<true>, dim3((unsigned)n), dim3(256), (size_t)2048 + (size_t)mw_q + 16, s, (const lf_ksw_prob *)d_p, n, (const uint8_t *)d_q,
                           (const uint8_t *)d_t, d_s, d_ql, d_tl, mw_q, r4 ? 1 : 0);
        if (qmax_narrow > mw_q) hipLaunchKernelGGL(lf_ksw_mw_kernel<false>, dim3((unsigned)n), dim3(256), (size_t)2048, s, (const lf_ksw_prob *)d_p, n, (const uint8_t *)d_q,
                           (const uint8_t *)d_t, d_s, d_ql, d_tl, mw_q, r4 ? 1 : 0);
    }
    if (n_wide) {
        const int wide_only = 1;
        hipLaunchKernelGGL(lf_ksw_kernel<true>, dim3((unsigned)n), dim3(64), (size_t)(2 * (lds_q + 2)) * 4, s, (const lf_ksw_prob *)d_p, n, (const uint8_t *)d_q,
                           (const uint8_t *)d_t, d_ws, d_s, d_ql, d_tl, lds_q, wide_only);
        if (qmax_wide > lds_q) hipLaunchKernelGGL(lf_ksw_kernel<false>, dim3((unsigned)n), dim3(64), 0, s, (const lf_ksw_prob *)d_p, n, (const uint8_t *)d_q,
                           (const uint8_t *)d_t, d_ws, d_s, d_ql, d_tl, lds_q, wide_only);
    }
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipMemcpyAsync(score, d_s, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(qle, d_ql, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(tle, d_tl, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (ms) HIPCHK(hipEventElapsedTime(ms, e0, e1));
    return LF_OK;
}
