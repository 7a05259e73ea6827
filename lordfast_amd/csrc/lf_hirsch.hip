/*
 * lf_hirsch.hip -- edlib's Hirschberg recursion (obtainAlignmentHirschberg, lib/edlib/edlib.cpp:1161-1330), BREADTH-FIRST
 * over all problems of a batch (gfx950).
 *
 * edlib switches from the plain traceback to Hirschberg's divide and conquer when the traceback data of a problem would
 * exceed 1 MiB (:1117-1119).  The recursion picks a different optimal path than the plain traceback on ties, so it has to be
 * followed node for node.  Round 2 did that depth-first inside ONE wavefront per problem (LDS stack): a few hundred
 * wavefronts per launch on a 1024-SIMD chip, each running ~2 m dependent sweep steps.  Here the recursion is turned inside
 * out: level l of ALL problems is one launch,
 *
 *   - a node is one 128-thread workgroup: wavefront 0 sweeps the left half of the target forwards, wavefront 1 the right
 *     half backwards (both strings reversed) -- at the same time; each leaves the last DP column as per-block (Pv, Mv, base)
 *     triples (24 B per 64 rows instead of 256 B of scores);
 *   - wavefront 0 then finds the split row in the reference's order -- rows 0 .. n-2 ascending, then -1, then n-1
 *     (:1263-1289), one ballot per 64 rows -- and emits the two children: a child that is small enough for the plain
 *     traceback (:1117-1119) becomes an ordinary leaf problem of the size-class kernels (lf_align.hip), a child with an empty
 *     side is a pure run, anything else is queued for the next level;
 *   - an NW root needs no distance pass of its own: D[n][m] = min over rows r of F[r] + R[n - r], which the first split
 *     computes anyway; an SHW root runs one distance sweep first (end column, :141-168) as a node of kind 1.
 *
 * Every piece (leaf path, run) lands end-aligned in its own part of the root's ops region; lf_hirsch_stitch_kernel closes
 * the gaps once the leaves' tracebacks are done, so that the root looks like any other problem (ops end-aligned, out_len).
 * Queries of any length: a wavefront sweeps 64 x KB blocks (KB = 1 / 4 / 8 registers-resident blocks per lane) and walks
 * longer queries band by band (32 768 rows), the two-bit carries that cross a band boundary going through HBM.
 */
#include "lf_hirsch.h"
#include <stdlib.h>
#include <type_traits>

#define LF_H_TC 256          /* LDS ring of target symbols per wavefront (power of two) */
#define LF_H_H  128

/* wavefront-local LDS ordering: DS operations of one wavefront execute in order; this only stops the compiler from moving
 * them across (two wavefronts of a node run different numbers of ring refills: no workgroup barrier may sit in the sweep) */
__device__ __forceinline__ void lf_wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

#define LF_H_B   16          /* multi-wavefront sweeps: steps between two workgroup barriers */
#define LF_H_LAG 96          /* ... and the number of steps a sub-band's wavefront runs behind the one above it (64 lanes + 2 B) */

/* One sweep of q[0..n) x t[0..m) by the W wavefronts of a node's half (anti-diagonal: lane l is l columns behind lane l - 1,
 * carry by DPP; wavefront wsub owns the blocks [wsub, wsub + 1) x 64 KB of every super-band of 64 KB W blocks and runs
 * LF_H_LAG steps behind wavefront wsub - 1, whose last lane leaves its two-bit carries in an LDS ring (cw_out -> cw_in); a
 * workgroup barrier every LF_H_B steps makes them visible long before they are read.  Round 3 gave a node's half to ONE
 * wavefront with KB = 4 / 8 blocks per lane: KB dependent block steps per sweep step -- the latency of every Hirschberg level,
 * and the levels of the few long problems are what a chunk with a clipped chain waits for).
 * blk_out != null: the last DP column as 3 u64 per 64-row block: Pv, Mv, D[64 b][m].
 * track: the SHW result (smallest prefix distance, smallest column on ties; the empty prefix only competes when n % 64 != 0,
 * lib/edlib/edlib.cpp:595,615).  idle: a wavefront group without a sweep of its own walks through the same barriers.
 * EVERY wavefront of the workgroup has to call this with the same n, m_max, track (barrier counts). */
template <int KB, int W, bool PAC>
__device__ __forceinline__ void lf_hsweep(const lf_qacc &Q, const lf_tacc &T, const int64_t pac_syms, const uint32_t n, const uint32_t m, const uint32_t m_max, const bool track, const bool idle,
                                          const int wsub, unsigned char *ring, unsigned char *cring, const unsigned char *cw_in, unsigned char *cw_out,
                                          uint64_t *peq, uint8_t *hc_a, uint8_t *hc_b, uint64_t *blk_out, int *s_tot, int *s_shw, int &shw_best, int &shw_c)
{
    constexpr bool PEQ = PAC && KB == 1;
    constexpr int TC = LF_H_TC, H = LF_H_H;
    constexpr uint32_t SB = 64u * KB * W;        /* blocks of a super-band */
    const int lane = threadIdx.x & 63;
    auto qget = [&](uint32_t r) -> unsigned char { return Q.get(r); };
    const uint32_t nbk = (n + 63) >> 6, lastb = (n - 1) >> 6;
    const int lastbit = (int)((n - 1) & 63);
    int col_base = (int)m;                       /* D[first row of the super-band][m] */
    int score = (int)n, best = (n & 63) ? (int)n : 0x7fffffff, best_c = 0;
    uint8_t *hc_in = hc_a, *hc_out = hc_b;
    bool had_last = false;
    for (uint32_t b0 = 0; b0 < nbk; b0 += SB) {
        const uint32_t sb0 = b0 + (uint32_t)wsub * 64u * KB;          /* this wavefront's sub-band */
        const uint32_t band_blocks = (idle || sb0 >= nbk) ? 0u : (nbk - sb0 < 64u * KB ? nbk - sb0 : 64u * KB);
        const int nl = (int)((band_blocks + KB - 1) / KB);
        const bool more_sb = b0 + SB < nbk;
        const bool from_hbm = wsub == 0 && b0 != 0, from_wave = wsub > 0;
        const bool to_wave = W > 1 && wsub + 1 < W && band_blocks && sb0 + 64u * KB < nbk, to_hbm = wsub == W - 1 && more_sb && band_blocks;
        const bool last_band = band_blocks && lastb >= sb0 && lastb < sb0 + 64u * KB;
        const int lane_last = last_band ? (int)((lastb - sb0) / KB) : -1;
        had_last = had_last || last_band;
        /* bit planes of the band's blocks by ballot: 64 query bytes per block, three ballots, the owning lane keeps them */
        uint64_t lo[KB], hi[KB], valid[KB], Pv[KB], Mv[KB];
#pragma unroll
        for (int k = 0; k < KB; k++) { lo[k] = hi[k] = valid[k] = 0; Pv[k] = ~0ull; Mv[k] = 0; }
        for (uint32_t b = 0; b < band_blocks; b++) {
            const uint32_t r = (sb0 + b) * 64 + (uint32_t)lane;
            bool ok; const uint32_t cd = lf_code_upper(qget(r < n ? r : n - 1), ok);
            const int code = (r < n && ok) ? (int)cd : -1;
            const uint64_t bl = lf_ballot(code >= 0 && (code & 1)), bh = lf_ballot(code >= 0 && (code & 2)), bv = lf_ballot(code >= 0);
            if ((uint32_t)lane == b / KB) {
                const int slot = (int)(b % KB);
#pragma unroll
                for (int k = 0; k < KB; k++) if (k == slot) { lo[k] = bl; hi[k] = bh; valid[k] = bv; }
            }
        }
        if (PEQ && band_blocks) {
#pragma unroll
            for (uint32_t cde = 0; cde < 4; cde++) peq[cde * 64 + lane] = lf_eq_tok<true>(cde, lo[0], hi[0], valid[0], qget, n, sb0 + (uint32_t)lane);
        }
        auto stage = [&](int first, int count) {
            for (int j = first + lane; j < first + count; j += 64) if (j >= 0 && (uint32_t)j < m) {
                ring[j & (TC - 1)] = PAC ? (unsigned char)T.pac_code((uint32_t)j) : T.get((uint32_t)j);
                if (from_hbm) cring[j & (TC - 1)] = hc_in[j];
            }
        };
        /* the carry entering lane 0 at the column with 0-based index x */
        auto carry_in = [&](int x) -> uint32_t { return from_wave ? (uint32_t)cw_in[x & 63] : from_hbm ? (uint32_t)cring[x & (TC - 1)] : LF_HIN_PLUS1; };
        const int my_steps = nl > 0 ? (int)m + nl - 1 : 0;
        const int t0 = wsub * LF_H_LAG;
        const int steps_total = W > 1 ? (int)m_max + 63 + (W - 1) * LF_H_LAG : my_steps;
        bool swept = false;
        if constexpr (PEQ) { {
            swept = true;
            /* One block per lane, 2-bit targets (every query of the mapping pipeline; round 5: also the queries above 4096 W rows, super-band by
             * super-band -- the carries that leave a super-band go to HBM in the same packed form, lf_hband_bytes): the step
             * loop of the forward kernel (lf_rsweep.hip) -- 16 steps unrolled, the lane's 16 target symbols in one register
             * straight from the 2-bit reference, match masks from the LDS table, no range test in the groups of 16 steps during
             * which every lane is inside the target; the carries between the wavefronts of a half travel as ONE 32-bit word per 16
             * steps.  ~40 instructions per step instead of ~150 in the general loop below: a level lasts as long as the sweep of
             * its longest node, which is a chain of dependent steps, and the chunks with long clipped tails (config C4) wait for
             * exactly that. */
            const uint64_t *peq_l = peq + lane;
            const uint32_t *cw32_in = reinterpret_cast<const uint32_t *>(cw_in); uint32_t *cw32_out = reinterpret_cast<uint32_t *>(cw_out);
            const uint32_t *h32_in = reinterpret_cast<const uint32_t *>(hc_in); uint32_t *h32_out = reinterpret_cast<uint32_t *>(hc_out);      /* one word per 16 steps of the super-band's last lane */
            const bool is_last = last_band && lane == lane_last;
            const int n_groups = (steps_total + 15) >> 4;
            uint32_t hout = LF_HIN_PLUS1, acc = 0;
            auto steps16 = [&](auto fast_tag, const int sl0, const uint32_t V, const uint32_t cin16) {
                constexpr bool FAST = decltype(fast_tag)::value;
                const int p0 = sl0 - lane;
                /* the group's sixteen match masks first, all in flight together (round 6): read inside the steps, each LDS access sat on the dependent chain of
                 * a wavefront that has its SIMD to itself -- 0.25 us per column for one wavefront, 0.45 - 0.55 for four / eight (profiles/r06_hirsch/) */
                uint64_t EQ[16];
#pragma unroll
                for (int k = 0; k < 16; k++) EQ[k] = peq_l[((V >> (2 * k)) & 3u) * 64];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t from_left = lf_wave_shr1(hout);
                    const uint32_t col0 = (uint32_t)(p0 + k);                /* column - 1; wraps for lanes that have not started */
                    if (FAST || (lane < nl && col0 < m)) {
                        const uint64_t Eq = EQ[k];
                        const uint32_t hin = lane == 0 ? ((cin16 >> (2 * k)) & 3u) : from_left;
                        uint64_t ph, mh;
                        hout = lf_myers_step(Pv[0], Mv[0], Eq, hin, ph, mh);
                        if (track) {
                            score += is_last ? lf_delta_at(ph, mh, lastbit) : 0;
                            const bool upd = is_last && score < best; best = upd ? score : best; best_c = upd ? (int)col0 + 1 : best_c;
                        }
                    }
                    acc |= hout << (2 * k);
                }
            };
            /* the 16 target symbols of the group that starts at local step sl (and, below a super-band boundary, the carry words of its columns) are
             * requested one group ahead: a global load right in front of its use was the other stall of the chain */
            auto fetchV = [&](int sl) -> uint32_t { return lf_pac16(T.pac, T.start + (int64_t)T.dir * ((int64_t)sl - lane), T.dir, T.comp, pac_syms); };
            const int g_first = t0 >> 4;                                      /* the first group with sl0 >= 0 */
            uint32_t Vn = fetchV(0), ha_n = 0, hb_n = 0;
            if (from_hbm && my_steps > 0) { ha_n = h32_in[3]; hb_n = h32_in[4]; }
            for (int g = 0; g < n_groups; g++) {
                if (W > 1) __syncthreads();
                const int sl0 = 16 * g - t0;                                 /* LF_H_LAG is a multiple of 16 */
                if (sl0 < 0 || sl0 >= my_steps) continue;
                (void)g_first;
                const uint32_t V = Vn, ha = ha_n, hb = hb_n;
                Vn = fetchV(sl0 + 16);
                if (from_hbm && sl0 + 16 < my_steps) { const int gl = (sl0 + 16) >> 4; ha_n = h32_in[gl + 3]; hb_n = h32_in[gl + 4]; }
                /* carries entering lane 0: +1 per column above the first block; else what the last lane of the wavefront above left
                 * for the same columns 63 steps (3 groups and 15 steps) later in ITS numbering */
                uint32_t cin16 = 0x55555555u;
                if (from_wave) { const int gl = sl0 >> 4; const uint32_t a = cw32_in[(gl + 3) & 7], b = cw32_in[(gl + 4) & 7]; cin16 = (a >> 30) | (b << 2); }
                else if (from_hbm) cin16 = (ha >> 30) | (hb << 2);             /* the same columns, left by the super-band above */
                acc = 0;
                if (sl0 >= nl - 1 && sl0 + 15 <= (int)m - 1) steps16(std::true_type{}, sl0, V, cin16);
                else steps16(std::false_type{}, sl0, V, cin16);
                if (to_wave && lane == 63) cw32_out[(sl0 >> 4) & 7] = acc;
                if (to_hbm && lane == 63) h32_out[sl0 >> 4] = acc;
            }
        } }
        if (!swept) {
        uint32_t hout_prev = LF_HIN_PLUS1, sym = 0, cin = LF_HIN_PLUS1;
        uint64_t eq = 0;
        for (int s = 0; s < steps_total; s++) {
            if (W > 1 && (s & (LF_H_B - 1)) == 0) __syncthreads();
            const int sl = s - t0;
            if (sl < 0 || sl >= my_steps) continue;
            if (sl == 0) {
                lf_wave_lds_sync(); stage(0, H); lf_wave_lds_sync();
                sym = ring[(0 - lane) & (TC - 1)];
                eq = PEQ ? peq[(sym & 3u) * 64 + lane] : 0ull;
                cin = carry_in(0);
            }
            if (((sl + 1) & (H - 1)) == 0) { lf_wave_lds_sync(); stage(sl + 1, H); lf_wave_lds_sync(); }
            const uint32_t sym_next = ring[(sl + 1 - lane) & (TC - 1)];
            const uint32_t cin_next = carry_in(sl + 1);
            const uint32_t from_left = lf_wave_shr1(hout_prev);
            const int c = sl - lane + 1;
            /* between the step at which the last lane enters its first column and the one at which lane 0 leaves its last, every
             * lane with blocks is inside the target: no per-lane range test (lanes without blocks compute on dead registers) */
            auto body = [&]() {
                const uint32_t tok = PAC ? sym : lf_tok_of_byte((unsigned char)sym);
                uint32_t hin = lane == 0 ? cin : from_left;
#pragma unroll
                for (int k = 0; k < KB; k++) {
                    const uint32_t b = sb0 + (uint32_t)lane * KB + k;
                    const uint64_t Eq = PEQ ? eq : lf_eq_tok<PAC>(tok, lo[k], hi[k], valid[k], qget, n, b);
                    uint64_t ph, mh;
                    const uint32_t ho = lf_myers_step(Pv[k], Mv[k], Eq, hin, ph, mh);
                    hin = (KB == 1 || b < nbk) ? ho : hin;      /* blocks past the last one compute on dead registers */
                    if (track) score += (b == lastb) ? lf_delta_at(ph, mh, lastbit) : 0;
                }
                hout_prev = hin;
                if (track && last_band) { const bool upd = lane == lane_last && score < best; best = upd ? score : best; best_c = upd ? c : best_c; }
                if (lane == 63) {                               /* the carry leaving the sub-band's last block (only full sub-bands pass one on) */
                    if (to_hbm) hc_out[c - 1] = (uint8_t)hin;
                    if (to_wave) cw_out[(c - 1) & 63] = (unsigned char)hin;
                }
            };
            if (sl >= nl - 1 && sl <= (int)m - 1) body();
            else if (lane < nl && c >= 1 && c <= (int)m) body();
            sym = sym_next; cin = cin_next;
            if (PEQ) eq = peq[(sym & 3u) * 64 + lane];
        }
        }
        /* last column of the sub-band: D[r][m] = D[first row][m] + vertical deltas */
        int mine = 0, part[KB];
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const uint32_t b = sb0 + (uint32_t)lane * KB + k;
            part[k] = 0;
            if (lane < nl && b < nbk) {
                const uint32_t rows = (b == lastb) ? (uint32_t)lastbit + 1 : 64;
                const uint64_t msk = rows >= 64 ? ~0ull : ((1ull << rows) - 1);
                part[k] = __popcll(Pv[k] & msk) - __popcll(Mv[k] & msk);
            }
            mine += part[k];
        }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const int wave_tot = __shfl(incl, 63);
        int pre = 0, all = wave_tot;
        if (W > 1) {
            if (lane == 0) s_tot[wsub] = wave_tot;
            __syncthreads();
            all = 0;
#pragma unroll
            for (int u = 0; u < W; u++) { const int x = s_tot[u]; pre += u < wsub ? x : 0; all += x; }
        }
        int base = col_base + pre + incl - mine;
        if (blk_out) {
#pragma unroll
            for (int k = 0; k < KB; k++) {
                const uint32_t b = sb0 + (uint32_t)lane * KB + k;
                if (lane < nl && b < nbk) { blk_out[3 * (size_t)b] = Pv[k]; blk_out[3 * (size_t)b + 1] = Mv[k]; blk_out[3 * (size_t)b + 2] = (uint64_t)(int64_t)base; }
                base += part[k];
            }
        }
        col_base += all;
        if (more_sb) { __threadfence(); uint8_t *t = hc_in; hc_in = hc_out; hc_out = t; }
        if (W > 1) __syncthreads();              /* s_tot is reused; the carries that left the super-band are in HBM */
    }
    if (track) {
        if (W > 1) {
            if (had_last) {
                const int src = (int)((lastb % (64u * KB)) / KB);
                const int b1 = __shfl(best, src), b2 = __shfl(best_c, src);
                if (lane == 0) { s_shw[0] = b1; s_shw[1] = b2; }
            }
            __syncthreads();
            shw_best = s_shw[0]; shw_c = s_shw[1];
        } else {
            const int src = (int)((lastb % (64u * KB)) / KB);
            shw_best = __shfl(best, src); shw_c = __shfl(best_c, src);
        }
    }
}

/* D[x][last column] from the per-block triples; x = 0: the first row (`zero`) */
__device__ __forceinline__ int lf_hcol(const uint64_t *__restrict__ B, uint32_t x, int zero)
{
    if (x == 0) return zero;
    const uint32_t b = (x - 1) >> 6, bit = (x - 1) & 63;
    const uint64_t pv = B[3 * (size_t)b], mv = B[3 * (size_t)b + 1];
    const uint64_t msk = bit == 63 ? ~0ull : ((2ull << bit) - 1);
    return (int)(int64_t)B[3 * (size_t)b + 2] + __popcll(pv & msk) - __popcll(mv & msk);
}

/* the roots above 4096 rows by distance / rows: the next calls' trial bounds are chosen from these counts (lf_align.hip) */
__device__ __forceinline__ void lf_hratio(const lf_hargs &A, int shw, int ed, uint32_t n)
{
    if (n <= 4096) return;
    uint32_t r = (uint32_t)(16ull * (uint32_t)ed / n); if (r > 15) r = 15;
    atomicAdd(&A.ctl->ratio_hist[shw][r], 1u);
}
/* registers a finished piece of a root's path (wave-uniform arguments; lane 0 writes) */
__device__ __forceinline__ void lf_hpiece(const lf_hargs &A, uint32_t root, uint64_t off, uint32_t cap, uint32_t len)
{
    if ((threadIdx.x & 63) == 0) {
        lf_hroot *R = &A.roots[root];
        const uint32_t slot = atomicAdd(&R->count, 1u);
        if (slot < R->seg_cap) { lf_hseg e; e.off = off; e.cap = cap; e.len = len; A.segs[R->seg_off + slot] = e; }
        else atomicExch(&A.ctl->fail, 2u);
    }
}

/* a child (or the truncated SHW root): run, leaf problem of the size classes, or a node of the next level */
__device__ __forceinline__ void lf_hfinalize(const lf_hargs &A, const lf_hnode &P, uint32_t qo, uint32_t cn, uint32_t to, uint32_t cm, int best, uint64_t off)
{
    const int lane = threadIdx.x & 63;
    if (cn == 0 && cm == 0) return;
    if (cn == 0 || cm == 0) {            /* lib/edlib/edlib.cpp:1096-1104: all deletions / all insertions */
        const uint32_t len = cn + cm; const uint8_t op = cn == 0 ? 2 : 1;
        for (uint32_t i = (uint32_t)lane; i < len; i += 64) A.ops[off + i] = op;
        lf_hpiece(A, P.root, off, len, len);
        return;
    }
    const int64_t dq = (P.flags & LF_F_QREV) ? -1 : 1, dt = (P.flags & LF_F_TREV) ? -1 : 1;
    if (lf_leaf(cn, cm)) {
        uint32_t j = 0;
        if (lane == 0) j = atomicAdd(&A.ctl->n_hleaf, 1u);
        j = (uint32_t)__shfl((int)j, 0);
        if (j >= A.hleaf_cap) { if (lane == 0) atomicExch(&A.ctl->fail, 3u); return; }
        if (lane == 0) {
            lf_aln_desc_t d;
            d.qstart = P.qstart + dq * (int64_t)qo; d.tstart = P.tstart + dt * (int64_t)to; d.n = cn; d.m = cm;
            d.flags = (uint8_t)(P.flags & ~LF_F_TPAC); d.mode = 0;
            for (int z = 0; z < 6; z++) d.pad[z] = 0;
            d.pad[0] = P.pad & 1u;                              /* stage API: the root's target holds bytes other than ACGT */
            A.hdesc[j] = d; A.hopsoff[j] = off;
        }
        lf_hpiece(A, P.root, off, cn + cm, 0x80000000u | j);
        return;
    }
    if (lane == 0) {
        uint32_t k0;
        const int kbc = lf_hqueue_of(cn, cm, best, 0, (P.pad & 1u) | A.no_band, 0, 0, &k0);      /* (its distance is known: a band of exactly that width, no trial) */
        const uint32_t idx = atomicAdd(&A.ctl->q_n[A.out_par][kbc], 1u);
        if (idx >= A.q_cap) { atomicExch(&A.ctl->fail, 4u); return; }
        lf_hnode c;
        c.qstart = P.qstart + dq * (int64_t)qo; c.tstart = P.tstart + dt * (int64_t)to; c.ops_off = off; c.n = cn; c.m = cm;
        c.best = best; c.root = P.root; c.flags = P.flags; c.kind = 0; c.is_root = 0; c.pad = P.pad & 1u; c.k0 = k0;
        A.q_out[kbc][idx] = c;
    }
}
/* a root whose trial bound was too small goes back to the queue.  `found` is what its sweep found inside the band -- the cost of a real path, an upper bound of the
 * distance -- so the band of `found` holds an optimal path and the second sweep cannot fail; nothing connected inside the band (LF_HB_FAR): the whole matrix */
__device__ __forceinline__ void lf_hrequeue_bound(const lf_hargs &A, const lf_hnode &P, int found)
{
    if ((threadIdx.x & 63) == 0) {
        uint32_t k0;
        const uint64_t whole = (uint64_t)P.n + P.m;
        const uint32_t t = found >= 0 && (uint64_t)found < whole ? (uint32_t)found : (uint32_t)whole;
        const int kbc = lf_hqueue_of_bound(P.n, P.m, -1, P.kind, P.pad | (A.no_band & 2u), t, &k0);
        const uint32_t idx = atomicAdd(&A.ctl->q_n[A.out_par][kbc], 1u);
        atomicAdd(&A.ctl->n_trial_failed, 1u);
        if (idx >= A.q_cap) { atomicExch(&A.ctl->fail, 4u); return; }
        lf_hnode c = P; c.k0 = k0;
        A.q_out[kbc][idx] = c;
    }
}

template <int KB, int W, bool PAC>
__global__ void __launch_bounds__(128 * W)
lf_hirsch_level_kernel(lf_hargs A)
{
    __shared__ unsigned char s_ring[2 * W][LF_H_TC], s_cring[2 * W][LF_H_TC], s_cw[2 * W][64];
    __shared__ uint64_t s_peq[2 * W][(PAC && KB == 1) ? 256 : 1];
    __shared__ int s_tot[2][W], s_shw[2][2];
    __shared__ unsigned long long s_base[2];
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int w = wave / W, wsub = wave % W;             /* w: the node's half; wsub: the wavefront's sub-band */
    if (blockIdx.x >= A.n_in) return;
    if (W > 1 && !(A.no_band & 4u)) __builtin_amdgcn_s_setprio(3);
    const lf_hnode P = A.q_in[blockIdx.x];
    const uint32_t n = P.n, m = P.m, nbk = (n + 63) >> 6;
    const bool banded = nbk > 64u * KB * W;
    const uint32_t lw = m / 2, rw = m - lw;
    if (threadIdx.x == 0) {
        s_base[0] = P.kind == 0 ? atomicAdd(&A.ctl->aux_used, 6ull * nbk) : 0ull;
        s_base[1] = banded ? atomicAdd(&A.ctl->hcar_used, lf_hband_reserve(m)) : 0ull;
    }
    __syncthreads();
    const bool aux_ok = P.kind != 0 || s_base[0] + 6ull * nbk <= A.aux_cap, hc_ok = !banded || s_base[1] + lf_hband_reserve(m) <= A.hcar_cap;
    if (!aux_ok || !hc_ok) { if (threadIdx.x == 0) { atomicExch(&A.ctl->fail, 5u); A.out_ed[A.roots[P.root].desc] = -2; } return; }
    uint64_t *Fb = A.aux + s_base[0], *Rb = Fb + 3 * (size_t)nbk;
    uint8_t *hc = A.hcar + s_base[1];
    static_assert(LF_H_LAG == 96, "lf_hband_bytes (lf_hirsch.h) counts the lag");
    constexpr bool PK = PAC && KB == 1;                   /* packed carries (lf_hsweep's 16-step loop) */
    const size_t HB = (size_t)lf_hband_bytes(m);
    const int64_t dq = (P.flags & LF_F_QREV) ? -1 : 1, dt = (P.flags & LF_F_TREV) ? -1 : 1;
    const uint32_t desc = A.roots[P.root].desc;
    const unsigned char *cw_in = wsub > 0 ? s_cw[wave - 1] : nullptr;

    if (P.kind == 1) {
        /* SHW root: distance and end column first (lib/edlib/edlib.cpp:141-168), then the path of q vs t[0 .. end].  The
         * wavefronts of the first half sweep; the others only keep the barriers company */
        if (W == 1 && w != 0) return;
        const lf_qacc Q(A.S.q, P.qstart, P.flags); const lf_tacc T(A.S.t, A.S.pac, P.tstart, P.flags | (PAC ? LF_F_TPAC : 0u));
        int ed = 0, tl = 0;
        lf_hsweep<KB, W, PAC>(Q, T, A.pac_syms, n, m, m, true, w != 0, wsub, s_ring[wave], s_cring[wave], cw_in, s_cw[wave], s_peq[wave], hc, PK ? hc + HB : hc + m + 32, nullptr, s_tot[w], s_shw[w], ed, tl);
        if (wave != 0) return;
        if (lane == 0) { A.out_ed[desc] = ed; A.out_end[desc] = tl - 1; lf_hratio(A, 1, ed, n); }
        lf_hfinalize(A, P, 0, n, 0, (uint32_t)tl, ed, P.ops_off);
        return;
    }
    {   /* the two half sweeps, W wavefronts each */
        int d0, d1;
        if (w == 0) {
            const lf_qacc Q(A.S.q, P.qstart, P.flags); const lf_tacc T(A.S.t, A.S.pac, P.tstart, P.flags | (PAC ? LF_F_TPAC : 0u));
            if (lw || W > 1) lf_hsweep<KB, W, PAC>(Q, T, A.pac_syms, n, lw, rw, false, lw == 0, wsub, s_ring[wave], s_cring[wave], cw_in, s_cw[wave], s_peq[wave], hc, PK ? hc + HB : hc + (lw + 16), Fb, s_tot[0], s_shw[0], d0, d1);
        } else {
            /* both strings backwards: element i = original element (len - 1 - i) */
            const unsigned fl = P.flags ^ (LF_F_QREV | LF_F_TREV);
            const lf_qacc Q(A.S.q, P.qstart + dq * (int64_t)(n - 1), fl); const lf_tacc T(A.S.t, A.S.pac, P.tstart + dt * (int64_t)(m - 1), fl | (PAC ? LF_F_TPAC : 0u));
            lf_hsweep<KB, W, PAC>(Q, T, A.pac_syms, n, rw, rw, false, false, wsub, s_ring[wave], s_cring[wave], cw_in, s_cw[wave], s_peq[wave], PK ? hc + 2 * HB : hc + 2 * (lw + 16), PK ? hc + 3 * HB : hc + 2 * (lw + 16) + (rw + 16), Rb, s_tot[1], s_shw[1], d0, d1);
        }
    }
    __threadfence_block();
    __syncthreads();
    if (wave != 0) return;
    /* F[x] = dist(q[0..x), t[0..lw)), R[x] = dist(last x of q, t[lw..m)) */
    auto F = [&](uint32_t x) -> int { return lw ? lf_hcol(Fb, x, (int)lw) : (int)x; };
    auto R = [&](uint32_t x) -> int { return lf_hcol(Rb, x, (int)rw); };
    int best = P.best;
    if (best < 0) {                          /* NW root: D[n][m] = min over rows r of F[r] + R[n - r] */
        int mn = 0x7fffffff;
        for (uint32_t base = 0; base <= n; base += 64) { const uint32_t r = base + (uint32_t)lane; if (r <= n) { const int v = F(r) + R(n - r); mn = v < mn ? v : mn; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(mn, o); mn = v < mn ? v : mn; }
        best = mn;
        if (lane == 0 && P.is_root) { A.out_ed[desc] = best; A.out_end[desc] = (int)m - 1; lf_hratio(A, 0, best, n); }
    }
    /* split row (lib/edlib/edlib.cpp:1263-1289): first qi in 0 .. n-2 with F[qi+1] + R[n-qi-1] == best, else -1, else n-1 */
    int split = -2, ls = 0, rs = 0;
    for (uint32_t base = 0; base + 2 <= n && split == -2; base += 64) {
        const uint32_t qi = base + (uint32_t)lane;
        const bool hit = qi + 2 <= n && F(qi + 1) + R(n - qi - 1) == best;
        const uint64_t bm = lf_ballot(hit);
        if (bm) split = (int)(base + (uint32_t)(__ffsll((long long)bm) - 1));
    }
    if (split >= 0) { ls = F((uint32_t)split + 1); rs = R(n - (uint32_t)split - 1); }
    else if ((int)lw + R(n) == best) { split = -1; ls = (int)lw; rs = R(n); }
    else if (F(n) + (int)rw == best) { split = (int)n - 1; ls = F(n); rs = (int)rw; }
    else { if (lane == 0) { atomicExch(&A.ctl->fail, 1u); A.out_ed[desc] = -2; } return; }      /* cannot happen for a consistent distance */
    const uint32_t ul = (uint32_t)(split + 1);
    lf_hfinalize(A, P, 0, ul, 0, lw, ls, P.ops_off);
    lf_hfinalize(A, P, ul, n - ul, lw, rw, rs, P.ops_off + ul + lw);
}

/* ================================================================================================
 * BANDED LEVELS (round 6).  What a level costs is the sweep of its LONGEST node -- a chain of m / 2 + blocks dependent steps -- and a node of more than 4096
 * rows used to be swept by 4 or 8 wavefronts per half, every block of every column, a workgroup barrier per 16 steps (0.27 - 0.5 us per column).  Below the
 * root a node's distance is known exactly (its side of `left + right == best`), so only the diagonals of lf_hband_nw can hold an optimal path: at 15 % error a
 * sixth of the matrix, and -- what matters for the chain -- few enough blocks per column for ONE wavefront, without a barrier:
 *
 *   - block b lives on lane b mod 64 W of the half's W wavefronts (W = 1 for bands of up to 4066 diagonals, 2 / 4 above) and handles column j at step
 *     j + skew(b), skew(b) = b for W = 1: the carry into block b + 1 is a DPP move from the lane on the left one step later, wave_ror:1 taking it from lane 63
 *     round to lane 0; with W > 1 the wavefront below runs 33 steps behind and takes its carries from an LDS ring (one word per 16 steps, as in lf_hsweep);
 *   - a lane keeps its block while the block is inside the band (columns [64 b + dlo, 64 b + 63 + dhi]), then takes block b + 64 W: blocks change hands at the
 *     boundaries of 16-step groups only, the new block starts at the group's first column from a column of +1s whose bottom value is the left neighbour's
 *     bottom value one column earlier + 64 (edlib's new block, lib/edlib/edlib.cpp:527-541);
 *   - +1 enters a block at the columns at which the block above has left the band (:500: `hout = 1`);
 *   - every lane follows the value at the bottom of its block (+ / - of its carries out, summed per group from the packed carry word): the last column leaves
 *     (Pv, Mv, value above the block) triples exactly like the unbanded sweep, for the blocks that are inside the band there -- rows outside read as "far";
 *   - match masks: four LDS words per lane, rebuilt from the read batch's bit planes when the lane changes blocks (planes of the NEXT block are fetched when a
 *     block starts, the next group's 16 target symbols while this group is stepped).
 * Roots do not know their distance: they try the band of lf_htrial_nw / _shw (k0) -- min (F + R) <= k0, resp. the SHW minimum <= k0, proves the result exact --
 * and go back to the queue for the unbanded sweep otherwise.  tests/models/hband_model.cpp is this schedule lane by lane on the host, checked against the
 * full matrix (tests/test_hband_model.py).
 * ================================================================================================ */
__device__ __forceinline__ uint32_t lf_wave_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C /* wave_ror:1 */, 0xf, 0xf, true); }
#define LF_HB_FAR (1 << 28)

template <int W, bool TRACK>
__device__ __forceinline__ void lf_hband_sweep(const lf_hargs &A, const int64_t qstart, const int64_t tstart, const unsigned flags, const uint32_t n, const int mm, const lf_hband B,
                                               const bool idle, const int nG, const int wsub, uint64_t *peq, const uint32_t *cw_in, const int *sc_in, uint32_t *cw_out, int *sc_out,
                                               uint64_t *blk_out, int *s_shw)
{
    constexpr int LAG = LF_HB_LAG(W), NL = 64 * W;
    const int lane = threadIdx.x & 63;
    const int dq = (flags & LF_F_QREV) ? -1 : 1, dt = (flags & LF_F_TREV) ? -1 : 1;
    const bool cq = (flags & LF_F_QCOMP) != 0, ct = (flags & LF_F_TCOMP) != 0;
    const int nbk = (int)((n + 63) >> 6), lastb = (int)((n - 1) >> 6), lastbit = (int)((n - 1) & 63);
    int nbA = idle || mm <= 0 ? 0 : (mm - 1 - B.dlo) / 64 + 1; if (nbA > nbk) nbA = nbk;
    auto skew = [&](int b) { return (b & 63) + LAG * (b >> 6); };
    auto jlo = [&](int b) { const int x = 64 * b + B.dlo; return x < 0 ? 0 : x; };
    auto jhi = [&](int b) { const int x = 64 * b + 63 + B.dhi; return x > mm - 1 ? mm - 1 : x; };
    /* bit planes of block bb out of the packed read batch (lf_rsweep.hip does the same for its one block per lane) */
    auto planes = [&](int bb, uint64_t &lo, uint64_t &hi, uint64_t &valid) {
        const int64_t r0 = (int64_t)bb * 64;
        const int64_t p0 = dq > 0 ? qstart + r0 : qstart - r0 - 63;
        lo = lf_bits64(A.qlo, p0, A.q_words); hi = lf_bits64(A.qhi, p0, A.q_words); valid = lf_bits64(A.qvalid, p0, A.q_words);
        if (dq < 0) { lo = lf_brev64(lo); hi = lf_brev64(hi); valid = lf_brev64(valid); }
        if (cq) { lo = ~lo; hi = ~hi; }
        const int64_t rows = (int64_t)n - r0;
        const uint64_t rmask = rows <= 0 ? 0ull : rows >= 64 ? ~0ull : ((1ull << rows) - 1);
        valid &= rmask; lo &= valid; hi &= valid;
    };
    int b = -1, nbnext = wsub * 64 + lane, skw = 0, sc = 0;
    uint32_t jend = 0, jtop = 0, hout = LF_HIN_PLUS1;
    uint64_t Pv = ~0ull, Mv = 0, nlo = 0, nhi = 0, nvalid = 0;
    int score = 0, best = 0x7fffffff, best_c = 0;
    if (nbnext < nbA) planes(nbnext, nlo, nhi, nvalid);
    uint64_t *peq_l = peq + lane;
    auto finish = [&]() {
        if (blk_out && jend == (uint32_t)mm) {
            blk_out[3 * (size_t)b] = Pv; blk_out[3 * (size_t)b + 1] = Mv;
            blk_out[3 * (size_t)b + 2] = (uint64_t)(int64_t)(sc - (__popcll(Pv) - __popcll(Mv)));
        }
        if (TRACK && b == lastb) { s_shw[0] = best; s_shw[1] = best_c; }
        b = -1; jend = 0;
    };
    /* the 16 target symbols of the group that starts at step s, for the block the lane will hold then */
    auto fetch = [&](int s, int sk) -> uint32_t { return lf_pac16(A.S.pac, tstart + (int64_t)dt * ((int64_t)s - sk), dt, ct, A.pac_syms); };
    uint32_t Vn = 0;
    {   /* group 0: blocks that start in it */
        const bool act0 = nbnext < nbA && 16 > jlo(nbnext) + skew(nbnext);
        Vn = fetch(0, act0 ? skew(nbnext) : 0);
    }
    for (int G = 0; G < nG; G++) {
        if (W > 1) __syncthreads();
        if (idle) continue;
        const int s0 = 16 * G;
        /* the left neighbour as it was at the end of the previous group (all lanes, before anything changes) */
        int sc_left; uint32_t h_left, cin16 = 0;
        if (W == 1) { sc_left = (int)lf_wave_ror1((uint32_t)sc); h_left = lf_wave_ror1(hout); }
        else {
            sc_left = (int)lf_wave_shr1((uint32_t)sc); h_left = lf_wave_shr1(hout);
            const uint32_t a = cw_in[(G + 5) & 7], c = cw_in[(G + 6) & 7];      /* groups G - 3 and G - 2 of the wavefront above: the same columns, 33 steps earlier */
            cin16 = (a >> 30) | (c << 2);
            if (lane == 0) { sc_left = sc_in[(G + 5) & 7]; h_left = cin16 & 3u; }
        }
        if (b >= 0 && s0 > (int)jend - 1 + skw) finish();
        if (b < 0 && nbnext < nbA && s0 + 16 > jlo(nbnext) + skew(nbnext)) {
            b = nbnext; nbnext += NL;
            skw = skew(b); jend = (uint32_t)(jhi(b) + 1);
            { const int x = 64 * b + B.dhi; jtop = b == 0 ? 0u : (uint32_t)(x > mm ? mm : x); }
            Pv = ~0ull; Mv = 0;
            sc = s0 - skw <= 0 ? 64 * (b + 1) : sc_left - ((int)(h_left & 1u) - (int)(h_left >> 1)) + 64;
            if (TRACK && b == lastb) { score = sc - (63 - lastbit); best = (n & 63) ? (int)n : 0x7fffffff; best_c = 0; }
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const uint64_t slo = 0ull - (uint64_t)(c & 1u), shi = 0ull - (uint64_t)(c >> 1);
                peq_l[c * 64] = ~((nlo ^ slo) | (nhi ^ shi)) & nvalid;
            }
            if (nbnext < nbA) planes(nbnext, nlo, nhi, nvalid);
        }
        const uint32_t V = Vn;
        {   /* next group's symbols: the block the lane will hold then is a matter of geometry, not of data */
            const int s1 = s0 + 16;
            const bool done1 = b >= 0 && s1 > (int)jend - 1 + skw;
            const bool act1 = (b < 0 || done1) && nbnext < nbA && s1 + 16 > jlo(nbnext) + skew(nbnext);
            Vn = fetch(s1, act1 ? skew(nbnext) : skw);
        }
        uint32_t acc = 0;
        if (lf_any(jend != 0u)) {
            const int p0 = s0 - skw;
            const bool is_last = TRACK && b == lastb;
            /* the group's sixteen match masks first, all in flight together: inside the steps an LDS read sits on the dependent chain of a wavefront that
             * has the SIMD to itself (the unbanded one-wavefront sweep takes 0.25 us per column that way) */
            uint64_t EQ[16];
#pragma unroll
            for (int k = 0; k < 16; k++) EQ[k] = peq_l[((V >> (2 * k)) & 3u) * 64];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t from_left = W == 1 ? lf_wave_ror1(hout) : lf_wave_shr1(hout);
                const uint32_t col0 = (uint32_t)(p0 + k);                    /* wraps for columns in front of the target */
                if (col0 < jend) {
                    const uint64_t Eq = EQ[k];
                    uint32_t hin = (W > 1 && lane == 0) ? ((cin16 >> (2 * k)) & 3u) : from_left;
                    hin = col0 < jtop ? hin : LF_HIN_PLUS1;
                    uint64_t ph, mh;
                    hout = lf_myers_step(Pv, Mv, Eq, hin, ph, mh);
                    acc |= hout << (2 * k);
                    if (TRACK) {
                        score += is_last ? lf_delta_at(ph, mh, lastbit) : 0;
                        const bool upd = is_last && score < best; best = upd ? score : best; best_c = upd ? (int)col0 + 1 : best_c;
                    }
                }
            }
        }
        sc += __popc(acc & 0x55555555u) - __popc(acc & 0xAAAAAAAAu);
        if (W > 1 && lane == 63) { cw_out[G & 7] = acc; sc_out[G & 7] = sc; }
    }
    if (b >= 0) finish();
}

/* groups of 16 steps a half needs (the same function of the geometry in every wavefront of the workgroup: barrier counts) */
template <int W>
__device__ __forceinline__ int lf_hband_groups(uint32_t n, int mm, lf_hband B)
{
    if (mm <= 0) return 0;
    const int nbk = (int)((n + 63) >> 6);
    int nbA = (mm - 1 - B.dlo) / 64 + 1; if (nbA > nbk) nbA = nbk;
    const int bl = nbA - 1;
    int jh = 64 * bl + 63 + B.dhi; if (jh > mm - 1) jh = mm - 1;
    return (jh + (bl & 63) + LF_HB_LAG(W) * (bl >> 6)) / 16 + 1;
}

/* D[x][last column] of a banded half: rows whose block is outside the band at the last column are "far" */
__device__ __forceinline__ int lf_hcol_band(const uint64_t *__restrict__ Bk, uint32_t x, int zero, int mm, lf_hband B, int nbk)
{
    if (x == 0) return zero;
    const int b = (int)((x - 1) >> 6);
    int nbA = (mm - 1 - B.dlo) / 64 + 1; if (nbA > nbk) nbA = nbk;
    if (b >= nbA || 64 * b + 63 + B.dhi < mm - 1) return LF_HB_FAR;
    return lf_hcol(Bk, x, zero);
}

template <int W, bool SHW>
__global__ void __launch_bounds__(SHW ? 64 * W : 128 * W)
lf_hband_level_kernel(lf_hargs A)
{
    constexpr int NW_ = SHW ? W : 2 * W;                  /* wavefronts of the workgroup: an SHW root has one half */
    __shared__ uint64_t s_peq[NW_][256];
    __shared__ uint32_t s_cw[NW_][8];
    __shared__ int s_scr[NW_][8], s_shw[2];
    __shared__ unsigned long long s_base;
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int w = wave / W, wsub = wave % W;             /* w: the node's half; wsub: the wavefront inside it */
    if (blockIdx.x >= A.n_in) return;
    if (W > 1 && !(A.no_band & 4u)) __builtin_amdgcn_s_setprio(W >= 8 ? 3 : W == 4 ? 2 : 1);
    const lf_hnode P = A.q_in[blockIdx.x];
    const uint32_t n = P.n, m = P.m, nbk = (n + 63) >> 6;
    const uint32_t lw = m / 2, rw = m - lw;
    const uint32_t desc = A.roots[P.root].desc;
    if (threadIdx.x == 0) {
        s_base = !SHW ? atomicAdd(&A.ctl->aux_used, 6ull * nbk) : 0ull;
        s_shw[0] = (n & 63) ? (int)n : 0x7fffffff; s_shw[1] = 0;
    }
    __syncthreads();
    if (!SHW && s_base + 6ull * nbk > A.aux_cap) { if (threadIdx.x == 0) { atomicExch(&A.ctl->fail, 5u); A.out_ed[desc] = -2; } return; }
    uint64_t *Fb = A.aux + s_base, *Rb = Fb + 3 * (size_t)nbk;
    const int64_t dq = (P.flags & LF_F_QREV) ? -1 : 1, dt = (P.flags & LF_F_TREV) ? -1 : 1;
    const int pw = w * W + (wsub + W - 1) % W;            /* the wavefront whose last lane hands its carries to this one's lane 0 */

    if constexpr (SHW) {
        /* SHW root inside the band of its bound: distance and end column (lib/edlib/edlib.cpp:141-168) */
        const int64_t k64 = (int64_t)P.k0; const int k = (int)k64;
        const lf_hband B = lf_hband_shw(k);
        const int64_t mme64 = (int64_t)n + k; const int mme = (int)(mme64 < (int64_t)m ? mme64 : (int64_t)m);
        const int nG = lf_hband_groups<W>(n, mme, B);
        lf_hband_sweep<W, true>(A, P.qstart, P.tstart, P.flags, n, mme, B, false, nG, wsub, s_peq[wave], s_cw[pw], s_scr[pw], s_cw[wave], s_scr[wave], nullptr, s_shw);
        __syncthreads();
        if (wave != 0) return;
        const int ed = s_shw[0], tl = s_shw[1];
        if (ed > k) { lf_hrequeue_bound(A, P, ed); return; }
        if (lane == 0) { A.out_ed[desc] = ed; A.out_end[desc] = tl - 1; lf_hratio(A, 1, ed, n); }
        lf_hfinalize(A, P, 0, n, 0, (uint32_t)tl, ed, P.ops_off);
        return;
    }
    const int k = P.best >= 0 ? P.best : (int)P.k0;
    const lf_hband B = lf_hband_nw(n, m, k);
    {
        const int gF = lf_hband_groups<W>(n, (int)lw, B), gR = lf_hband_groups<W>(n, (int)rw, B);
        const int nG = W > 1 ? (gF > gR ? gF : gR) : (w == 0 ? gF : gR);
        if (w == 0) {
            lf_hband_sweep<W, false>(A, P.qstart, P.tstart, P.flags, n, (int)lw, B, lw == 0, nG, wsub, s_peq[wave], s_cw[pw], s_scr[pw], s_cw[wave], s_scr[wave], Fb, s_shw);
        } else {
            /* both strings backwards: element i = original element (len - 1 - i) */
            const unsigned fl = P.flags ^ (LF_F_QREV | LF_F_TREV);
            lf_hband_sweep<W, false>(A, P.qstart + dq * (int64_t)(n - 1), P.tstart + dt * (int64_t)(m - 1), fl, n, (int)rw, B, false, nG, wsub, s_peq[wave], s_cw[pw], s_scr[pw], s_cw[wave], s_scr[wave], Rb, s_shw);
        }
    }
    __threadfence_block();
    __syncthreads();
    if (wave != 0) return;
    auto F = [&](uint32_t x) -> int { return lw ? lf_hcol_band(Fb, x, (int)lw, (int)lw, B, (int)nbk) : (int)x; };
    auto R = [&](uint32_t x) -> int { return lf_hcol_band(Rb, x, (int)rw, (int)rw, B, (int)nbk); };
    int best = P.best;
    if (best < 0) {                          /* a root inside its trial band: min (F + R) <= k0 is the distance, anything larger proves nothing */
        int mn = 0x7fffffff;
        for (uint32_t base = 0; base <= n; base += 64) { const uint32_t r = base + (uint32_t)lane; if (r <= n) { const int v = F(r) + R(n - r); mn = v < mn ? v : mn; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(mn, o); mn = v < mn ? v : mn; }
        if (mn > k) { lf_hrequeue_bound(A, P, mn); return; }
        best = mn;
        if (lane == 0 && P.is_root) { A.out_ed[desc] = best; A.out_end[desc] = (int)m - 1; lf_hratio(A, 0, best, n); }
    }
    /* split row (lib/edlib/edlib.cpp:1263-1289), exactly as in the unbanded kernel: rows outside the band cannot satisfy the equality */
    int split = -2, ls = 0, rs = 0;
    for (uint32_t base = 0; base + 2 <= n && split == -2; base += 64) {
        const uint32_t qi = base + (uint32_t)lane;
        const bool hit = qi + 2 <= n && F(qi + 1) + R(n - qi - 1) == best;
        const uint64_t bm = lf_ballot(hit);
        if (bm) split = (int)(base + (uint32_t)(__ffsll((long long)bm) - 1));
    }
    if (split >= 0) { ls = F((uint32_t)split + 1); rs = R(n - (uint32_t)split - 1); }
    else if ((int)lw + R(n) == best) { split = -1; ls = (int)lw; rs = R(n); }
    else if (F(n) + (int)rw == best) { split = (int)n - 1; ls = F(n); rs = (int)rw; }
    else { if (lane == 0) { atomicExch(&A.ctl->fail, 1u); A.out_ed[desc] = -2; } return; }
    const uint32_t ul = (uint32_t)(split + 1);
    lf_hfinalize(A, P, 0, ul, 0, lw, ls, P.ops_off);
    lf_hfinalize(A, P, ul, n - ul, lw, rw, rs, P.ops_off + ul + lw);
}

/* ---- SIXTEEN / THIRTY-TWO LANES PER HALF (round 6): most nodes of a batch are small -- C5's gaps between sparse anchors are 2 000 - 5 000 rows at 10 % error, tens of
 * thousands per chunk and level, its roots 7 000 rows inside a trial band of 1 300 diagonals -- and their band needs a handful of blocks per column: a half of such a node on a whole
 * wavefront leaves most of it idle, and with that many nodes a level is bound by VALU issue (a step is ~40 instructions, four cycles each on a SIMD), not by the longest chain.
 * Here a half is a group of L = 16 or 32 lanes (block b on lane b mod L of the group; the carry round the group by DPP row_ror:1, resp. wave_ror:1 with lanes 0 and 32 set right by two
 * v_readlane; everything else as in lf_hband_sweep with W = 1): a wavefront sweeps the two halves of 32 / L nodes, or 64 / L SHW roots, at once.  What is uniform per wavefront in
 * lf_hband_sweep (the node's sizes, band, strings) is per lane here. ---- */
template <int L>
__device__ __forceinline__ uint32_t lf_group_ror1(uint32_t v)
{
    if constexpr (L == 16) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121 /* row_ror:1 */, 0xf, 0xf, true);
    else {
        const uint32_t r = lf_wave_ror1(v);                /* lane i <- lane i - 1, lane 0 <- lane 63 */
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
        const int lane = threadIdx.x & 63;
        return lane == 0 ? a : lane == 32 ? b : r;
    }
}
template <int L, bool TRACK>
__device__ __forceinline__ void lf_hband_sweepL(const lf_hargs &A, const int64_t qstart, const int64_t tstart, const unsigned flags, const uint32_t n, const int mm, const lf_hband B,
                                                 const bool idle, const int nG, uint64_t *peq, uint64_t *blk_out, int *s_shw)
{
    const int lane = threadIdx.x & 63, gl = lane & (L - 1);
    const int dq = (flags & LF_F_QREV) ? -1 : 1, dt = (flags & LF_F_TREV) ? -1 : 1;
    const bool cq = (flags & LF_F_QCOMP) != 0, ct = (flags & LF_F_TCOMP) != 0;
    const int nbk = (int)((n + 63) >> 6), lastb = (int)((n - 1) >> 6), lastbit = (int)((n - 1) & 63);
    int nbA = idle || mm <= 0 ? 0 : (mm - 1 - B.dlo) / 64 + 1; if (nbA > nbk) nbA = nbk;
    auto jlo = [&](int b) { const int x = 64 * b + B.dlo; return x < 0 ? 0 : x; };
    auto jhi = [&](int b) { const int x = 64 * b + 63 + B.dhi; return x > mm - 1 ? mm - 1 : x; };
    auto planes = [&](int bb, uint64_t &lo, uint64_t &hi, uint64_t &valid) {
        const int64_t r0 = (int64_t)bb * 64;
        const int64_t p0 = dq > 0 ? qstart + r0 : qstart - r0 - 63;
        lo = lf_bits64(A.qlo, p0, A.q_words); hi = lf_bits64(A.qhi, p0, A.q_words); valid = lf_bits64(A.qvalid, p0, A.q_words);
        if (dq < 0) { lo = lf_brev64(lo); hi = lf_brev64(hi); valid = lf_brev64(valid); }
        if (cq) { lo = ~lo; hi = ~hi; }
        const int64_t rows = (int64_t)n - r0;
        const uint64_t rmask = rows <= 0 ? 0ull : rows >= 64 ? ~0ull : ((1ull << rows) - 1);
        valid &= rmask; lo &= valid; hi &= valid;
    };
    int b = -1, nbnext = gl, sc = 0;                       /* (skew(b) = b: L lanes, one "wavefront") */
    uint32_t jend = 0, jtop = 0, hout = LF_HIN_PLUS1;
    uint64_t Pv = ~0ull, Mv = 0, nlo = 0, nhi = 0, nvalid = 0;
    int score = 0, best = 0x7fffffff, best_c = 0;
    if (nbnext < nbA) planes(nbnext, nlo, nhi, nvalid);
    uint64_t *peq_l = peq + lane;
    auto finish = [&]() {
        if (blk_out && jend == (uint32_t)mm) {
            blk_out[3 * (size_t)b] = Pv; blk_out[3 * (size_t)b + 1] = Mv;
            blk_out[3 * (size_t)b + 2] = (uint64_t)(int64_t)(sc - (__popcll(Pv) - __popcll(Mv)));
        }
        if (TRACK && b == lastb) { s_shw[0] = best; s_shw[1] = best_c; }
        b = -1; jend = 0;
    };
    auto fetch = [&](int s, int sk) -> uint32_t { return lf_pac16(A.S.pac, tstart + (int64_t)dt * ((int64_t)s - sk), dt, ct, A.pac_syms); };
    uint32_t Vn;
    { const bool act0 = nbnext < nbA && 16 > jlo(nbnext) + nbnext; Vn = fetch(0, act0 ? nbnext : 0); }
    for (int G = 0; G < nG; G++) {
        const int s0 = 16 * G;
        const int sc_left = (int)lf_group_ror1<L>((uint32_t)sc); const uint32_t h_left = lf_group_ror1<L>(hout);
        if (b >= 0 && s0 > (int)jend - 1 + b) finish();
        if (b < 0 && nbnext < nbA && s0 + 16 > jlo(nbnext) + nbnext) {
            b = nbnext; nbnext += L;
            jend = (uint32_t)(jhi(b) + 1);
            { const int x = 64 * b + B.dhi; jtop = b == 0 ? 0u : (uint32_t)(x > mm ? mm : x); }
            Pv = ~0ull; Mv = 0;
            sc = s0 - b <= 0 ? 64 * (b + 1) : sc_left - ((int)(h_left & 1u) - (int)(h_left >> 1)) + 64;
            if (TRACK && b == lastb) { score = sc - (63 - lastbit); best = (n & 63) ? (int)n : 0x7fffffff; best_c = 0; }
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const uint64_t slo = 0ull - (uint64_t)(c & 1u), shi = 0ull - (uint64_t)(c >> 1);
                peq_l[c * 64] = ~((nlo ^ slo) | (nhi ^ shi)) & nvalid;
            }
            if (nbnext < nbA) planes(nbnext, nlo, nhi, nvalid);
        }
        const uint32_t V = Vn;
        const int skw = b >= 0 ? b : 0;
        {
            const int s1 = s0 + 16;
            const bool done1 = b >= 0 && s1 > (int)jend - 1 + b;
            const bool act1 = (b < 0 || done1) && nbnext < nbA && s1 + 16 > jlo(nbnext) + nbnext;
            Vn = fetch(s1, act1 ? nbnext : skw);
        }
        uint32_t acc = 0;
        if (lf_any(jend != 0u)) {
            const int p0 = s0 - skw;
            const bool is_last = TRACK && b == lastb;
            uint64_t EQ[16];
#pragma unroll
            for (int k = 0; k < 16; k++) EQ[k] = peq_l[((V >> (2 * k)) & 3u) * 64];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t from_left = lf_group_ror1<L>(hout);
                const uint32_t col0 = (uint32_t)(p0 + k);
                if (col0 < jend) {
                    const uint32_t hin = col0 < jtop ? from_left : LF_HIN_PLUS1;
                    uint64_t ph, mh;
                    hout = lf_myers_step(Pv, Mv, EQ[k], hin, ph, mh);
                    acc |= hout << (2 * k);
                    if (TRACK) {
                        score += is_last ? lf_delta_at(ph, mh, lastbit) : 0;
                        const bool upd = is_last && score < best; best = upd ? score : best; best_c = upd ? (int)col0 + 1 : best_c;
                    }
                }
            }
        }
        sc += __popc(acc & 0x55555555u) - __popc(acc & 0xAAAAAAAAu);
    }
    if (b >= 0) finish();
}
__device__ __forceinline__ int lf_hband_groupsL(uint32_t n, int mm, lf_hband B)
{
    if (mm <= 0) return 0;
    const int nbk = (int)((n + 63) >> 6);
    int nbA = (mm - 1 - B.dlo) / 64 + 1; if (nbA > nbk) nbA = nbk;
    const int bl = nbA - 1;
    int jh = 64 * bl + 63 + B.dhi; if (jh > mm - 1) jh = mm - 1;
    return (jh + bl) / 16 + 1;
}

/* one wavefront: SHW = false: the two halves of the nodes NPW blockIdx.x .. + NPW - 1 (L = 16: two nodes, L = 32: one); SHW = true: as many roots as the wavefront has groups */
template <int L, bool SHW>
__global__ void __launch_bounds__(64)
lf_hband_group_kernel(lf_hargs A)
{
    constexpr int NGRP = 64 / L, NPW = SHW ? NGRP : NGRP / 2;      /* groups of lanes, nodes per wavefront */
    __shared__ uint64_t s_peq[256];
    __shared__ int s_shw[NGRP][2];
    __shared__ unsigned long long s_base[NPW];
    const int lane = (int)threadIdx.x, grp = lane / L;
    const int slot = SHW ? grp : grp >> 1;                 /* the wavefront's node this lane works for */
    const uint32_t idx = (uint32_t)blockIdx.x * NPW + (uint32_t)slot;
    const bool live = idx < A.n_in;
    const lf_hnode P = A.q_in[live ? idx : A.n_in - 1];
    const uint32_t n = P.n, m = P.m, nbk = (n + 63) >> 6;
    const uint32_t lw = m / 2, rw = m - lw;
    if (!SHW && (lane & (2 * L - 1)) == 0 && live) s_base[slot] = atomicAdd(&A.ctl->aux_used, 6ull * nbk);
    if ((lane & (L - 1)) == 0) { s_shw[grp][0] = (n & 63) ? (int)n : 0x7fffffff; s_shw[grp][1] = 0; }
    __syncthreads();
    const bool aux_ok = SHW || !live || s_base[slot] + 6ull * nbk <= A.aux_cap;
    if (!SHW && live && !aux_ok && (lane & (2 * L - 1)) == 0) { atomicExch(&A.ctl->fail, 5u); A.out_ed[A.roots[P.root].desc] = -2; }
    uint64_t *Fb = SHW ? nullptr : A.aux + s_base[slot], *Rb = SHW ? nullptr : Fb + 3 * (size_t)nbk;
    const int64_t dq = (P.flags & LF_F_QREV) ? -1 : 1, dt = (P.flags & LF_F_TREV) ? -1 : 1;
    if constexpr (SHW) {
        const int k = (int)P.k0;
        const lf_hband B = lf_hband_shw(k);
        const int64_t mme64 = (int64_t)n + k; const int mme = (int)(mme64 < (int64_t)m ? mme64 : (int64_t)m);
        const int nG = lf_wave_max_i32(live ? lf_hband_groupsL(n, mme, B) : 0);
        lf_hband_sweepL<L, true>(A, P.qstart, P.tstart, P.flags, n, mme, B, !live, nG, s_peq, nullptr, s_shw[grp]);
        __syncthreads();
        for (int u = 0; u < NPW; u++) {                    /* the roots one after the other, with the whole wavefront */
            const uint32_t iu = (uint32_t)blockIdx.x * NPW + (uint32_t)u;
            if (iu >= A.n_in) break;
            const lf_hnode Pu = A.q_in[iu];
            const uint32_t desc = A.roots[Pu.root].desc;
            const int ed = s_shw[u][0], tl = s_shw[u][1];
            if (ed > (int)Pu.k0) { lf_hrequeue_bound(A, Pu, ed); continue; }
            if (lane == 0) { A.out_ed[desc] = ed; A.out_end[desc] = tl - 1; lf_hratio(A, 1, ed, Pu.n); }
            lf_hfinalize(A, Pu, 0, Pu.n, 0, (uint32_t)tl, ed, Pu.ops_off);
        }
        return;
    } else {
        const int k = P.best >= 0 ? P.best : (int)P.k0;
        const lf_hband B = lf_hband_nw(n, m, k);
        const bool rev = (grp & 1) != 0;                   /* the group's half: even groups forwards, odd groups backwards */
        const int mm = rev ? (int)rw : (int)lw;
        const bool ok = live && aux_ok;
        const int nG = lf_wave_max_i32(ok ? lf_hband_groupsL(n, mm, B) : 0);
        const unsigned fl = rev ? (P.flags ^ (LF_F_QREV | LF_F_TREV)) : P.flags;
        const int64_t qs = rev ? P.qstart + dq * (int64_t)(n - 1) : P.qstart, ts = rev ? P.tstart + dt * (int64_t)(m - 1) : P.tstart;
        lf_hband_sweepL<L, false>(A, qs, ts, fl, n, mm, B, !ok || mm == 0, nG, s_peq, rev ? Rb : Fb, s_shw[grp]);
        __threadfence_block();
        __syncthreads();
        for (int u = 0; u < NPW; u++) {                    /* split row and children of the two nodes, one after the other, with the whole wavefront */
            const uint32_t iu = (uint32_t)blockIdx.x * NPW + (uint32_t)u;
            if (iu >= A.n_in) break;
            const lf_hnode Pu = A.q_in[iu];
            const uint32_t nu = Pu.n, mu = Pu.m, nbku = (nu + 63) >> 6, lwu = mu / 2, rwu = mu - lwu;
            if (s_base[u] + 6ull * nbku > A.aux_cap) continue;
            const uint64_t *Fu = A.aux + s_base[u], *Ru = Fu + 3 * (size_t)nbku;
            const uint32_t desc = A.roots[Pu.root].desc;
            const int ku = Pu.best >= 0 ? Pu.best : (int)Pu.k0;
            const lf_hband Bu = lf_hband_nw(nu, mu, ku);
            auto F = [&](uint32_t x) -> int { return lwu ? lf_hcol_band(Fu, x, (int)lwu, (int)lwu, Bu, (int)nbku) : (int)x; };
            auto R = [&](uint32_t x) -> int { return lf_hcol_band(Ru, x, (int)rwu, (int)rwu, Bu, (int)nbku); };
            int bestu = Pu.best;
            if (bestu < 0) {
                int mn = 0x7fffffff;
                for (uint32_t base = 0; base <= nu; base += 64) { const uint32_t r = base + (uint32_t)lane; if (r <= nu) { const int v = F(r) + R(nu - r); mn = v < mn ? v : mn; } }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(mn, o); mn = v < mn ? v : mn; }
                if (mn > ku) { lf_hrequeue_bound(A, Pu, mn); continue; }
                bestu = mn;
                if (lane == 0 && Pu.is_root) { A.out_ed[desc] = bestu; A.out_end[desc] = (int)mu - 1; lf_hratio(A, 0, bestu, nu); }
            }
            int split = -2, ls = 0, rs = 0;
            for (uint32_t base = 0; base + 2 <= nu && split == -2; base += 64) {
                const uint32_t qi = base + (uint32_t)lane;
                const bool hit = qi + 2 <= nu && F(qi + 1) + R(nu - qi - 1) == bestu;
                const uint64_t bm = lf_ballot(hit);
                if (bm) split = (int)(base + (uint32_t)(__ffsll((long long)bm) - 1));
            }
            if (split >= 0) { ls = F((uint32_t)split + 1); rs = R(nu - (uint32_t)split - 1); }
            else if ((int)lwu + R(nu) == bestu) { split = -1; ls = (int)lwu; rs = R(nu); }
            else if (F(nu) + (int)rwu == bestu) { split = (int)nu - 1; ls = F(nu); rs = (int)rwu; }
            else { if (lane == 0) { atomicExch(&A.ctl->fail, 1u); A.out_ed[desc] = -2; } continue; }
            const uint32_t ul = (uint32_t)(split + 1);
            lf_hfinalize(A, Pu, 0, ul, 0, lwu, ls, Pu.ops_off);
            lf_hfinalize(A, Pu, ul, nu - ul, lwu, rwu, rs, Pu.ops_off + ul + lwu);
        }
    }
}

/* the problems above edlib's traceback switch become roots: a table entry + a node of level 0 */
__global__ void lf_hirsch_roots_kernel(const lf_aln_desc_t *__restrict__ d, const uint64_t *__restrict__ ops_off, int n, lf_hargs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const lf_aln_desc_t x = d[i];
    if (x.n == 0 || x.m == 0 || lf_leaf(x.n, x.m)) return;
    const uint32_t r = atomicAdd(&A.ctl->n_roots, 1u), cap = lf_hroot_cap(x.n, x.m), so = atomicAdd(&A.ctl->seg_used, cap);
    uint32_t k0;
    const int kbc = lf_hqueue_of(x.n, x.m, -1, x.mode ? 1 : 0, (x.pad[0] != 0 ? 1u : 0u) | A.no_band, A.trial16[x.mode ? 1 : 0], A.trial_min, &k0);
    const uint32_t idx = atomicAdd(&A.ctl->q_n[A.out_par][kbc], 1u);
    if (idx >= A.q_cap) { atomicExch(&A.ctl->fail, 4u); return; }
    if (k0 && k0 != x.n + x.m) atomicAdd(&A.ctl->n_trial, 1u);
    atomicMax(&A.ctl->max_root_n, x.n);
    lf_hroot R; R.ops_off = ops_off[i]; R.desc = (uint32_t)i; R.n = x.n; R.m = x.m; R.seg_off = so; R.seg_cap = cap; R.count = 0;
    A.roots[r] = R;
    lf_hnode c;
    c.qstart = x.qstart; c.tstart = x.tstart; c.ops_off = ops_off[i]; c.n = x.n; c.m = x.m; c.best = -1; c.root = r;
    c.flags = x.flags; c.kind = x.mode ? 1 : 0; c.is_root = 1; c.pad = x.pad[0] ? 1 : 0; c.k0 = k0;
    A.q_out[kbc][idx] = c;
}

/* after the leaves' tracebacks: a root's pieces, each end-aligned in its own part of the root's region, are moved together
 * (last piece first, towards higher addresses) so that the root's path is end-aligned in one piece like any other problem's */
__global__ void __launch_bounds__(64)
lf_hirsch_stitch_kernel(lf_hargs A, uint32_t n_roots)
{
    const uint32_t r = blockIdx.x; const int lane = threadIdx.x;
    if (r >= n_roots) return;
    const lf_hroot R = A.roots[r];
    const uint32_t c = R.count < R.seg_cap ? R.count : R.seg_cap;
    const lf_hseg *seg = A.segs + R.seg_off;
    const uint64_t end = R.ops_off + R.n + R.m;
    uint64_t w = end, cur = ~0ull;
    for (uint32_t it = 0; it < c; it++) {
        /* the piece with the largest offset below `cur` (offsets are distinct: empty pieces are never registered) */
        unsigned long long key = 0;
        for (uint32_t k = (uint32_t)lane; k < c; k += 64) { const uint64_t o = seg[k].off; if (o < cur) { const unsigned long long kk = ((unsigned long long)(o - R.ops_off + 1) << 24) | k; key = kk > key ? kk : key; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(key, o); key = v > key ? v : key; }
        if (key == 0) break;
        const lf_hseg e = seg[key & 0xffffffu];
        const uint32_t len = (e.len & 0x80000000u) ? A.out_len[A.n_desc + (e.len & 0x7fffffffu)] : e.len;
        const uint64_t src = e.off + e.cap - len, dst = w - len;
        if (src != dst) {
            for (int64_t hi = (int64_t)len; hi > 0; hi -= 256) {
                uint8_t v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int64_t i = hi - 1 - (u * 64 + lane); v[u] = i >= 0 ? A.ops[src + (uint64_t)i] : (uint8_t)0; }
#pragma unroll
                for (int u = 0; u < 4; u++) { const int64_t i = hi - 1 - (u * 64 + lane); if (i >= 0) A.ops[dst + (uint64_t)i] = v[u]; }
            }
        }
        w -= len; cur = e.off;
    }
    if (lane == 0) A.out_len[R.desc] = (uint32_t)(end - w);
}

void lf_hirsch_launch_roots(hipStream_t s, bool pac_targets, const lf_aln_desc_t *d_desc, const uint64_t *d_opsoff, int n, lf_hargs A)
{
    (void)pac_targets;
    hipLaunchKernelGGL(lf_hirsch_roots_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_desc, d_opsoff, n, A);
}
void lf_hirsch_launch_level(hipStream_t s, bool pac, int kbc, lf_hargs A)
{
    if (A.n_in == 0) return;
    /* thirty-two lanes per half cost two v_readlane per step on the dependent chain: worth it when the level has more nodes than the GPU holds wavefronts at once (a
     * level bound by VALU issue: half the wavefronts), not when all of them are resident anyway (a level bound by its longest chain: the one-wavefront kernel's
     * step is shorter) */
    if (kbc == LF_HQ_NW32 && A.n_in < 2500u) kbc = LF_HQ_NW0;
    if (kbc == LF_HQ_SHW32 && A.n_in < 5000u) kbc = LF_HQ_SHW0;
    if (kbc == LF_HQ_NW16) { hipLaunchKernelGGL((lf_hband_group_kernel<16, false>), dim3((A.n_in + 1) / 2), dim3(64), 0, s, A); return; }
    if (kbc == LF_HQ_SHW16) { hipLaunchKernelGGL((lf_hband_group_kernel<16, true>), dim3((A.n_in + 3) / 4), dim3(64), 0, s, A); return; }
    if (kbc == LF_HQ_NW32) { hipLaunchKernelGGL((lf_hband_group_kernel<32, false>), dim3(A.n_in), dim3(64), 0, s, A); return; }
    if (kbc == LF_HQ_SHW32) { hipLaunchKernelGGL((lf_hband_group_kernel<32, true>), dim3((A.n_in + 1) / 2), dim3(64), 0, s, A); return; }
    if (kbc >= LF_HQ_NW0) {      /* banded sweeps, by the wavefronts the band needs */
        const dim3 gb(A.n_in);
        switch (kbc) {
        case LF_HQ_NW0 + 0: hipLaunchKernelGGL((lf_hband_level_kernel<1, false>), gb, dim3(128), 0, s, A); break;
        case LF_HQ_NW0 + 1: hipLaunchKernelGGL((lf_hband_level_kernel<2, false>), gb, dim3(256), 0, s, A); break;
        case LF_HQ_NW0 + 2: hipLaunchKernelGGL((lf_hband_level_kernel<4, false>), gb, dim3(512), 0, s, A); break;
        case LF_HQ_NW0 + 3: hipLaunchKernelGGL((lf_hband_level_kernel<8, false>), gb, dim3(1024), 0, s, A); break;
        case LF_HQ_SHW0 + 0: hipLaunchKernelGGL((lf_hband_level_kernel<1, true>), gb, dim3(64), 0, s, A); break;
        case LF_HQ_SHW0 + 1: hipLaunchKernelGGL((lf_hband_level_kernel<2, true>), gb, dim3(128), 0, s, A); break;
        case LF_HQ_SHW0 + 2: hipLaunchKernelGGL((lf_hband_level_kernel<4, true>), gb, dim3(256), 0, s, A); break;
        case LF_HQ_SHW0 + 3: hipLaunchKernelGGL((lf_hband_level_kernel<8, true>), gb, dim3(512), 0, s, A); break;
        default: hipLaunchKernelGGL((lf_hband_level_kernel<16, true>), gb, dim3(1024), 0, s, A); break;
        }
        return;
    }
    /* blocks per lane x wavefronts per half: one block per lane, 1 / 4 / 8 wavefronts: queries of <= 4096 / 16384 / 32768 rows in one
     * super-band (more rows: several) */
    const dim3 g(A.n_in);
#define LV(KBV, WV) do { if (pac) hipLaunchKernelGGL((lf_hirsch_level_kernel<KBV, WV, true>), g, dim3(128 * WV), 0, s, A); else hipLaunchKernelGGL((lf_hirsch_level_kernel<KBV, WV, false>), g, dim3(128 * WV), 0, s, A); } while (0)
    if (kbc == 0) LV(1, 1);
    else if (kbc == 1) LV(1, 4); else LV(1, 8);
#undef LV
}
void lf_hirsch_launch_stitch(hipStream_t s, lf_hargs A, uint32_t n_roots)
{
    if (n_roots) hipLaunchKernelGGL(lf_hirsch_stitch_kernel, dim3(n_roots), dim3(64), 0, s, A, n_roots);
}
