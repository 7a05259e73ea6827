/* lf_tb_core.h -- one path per lane: the traceback of a problem of the one-block-per-lane forward kernel from its checkpoint rows
 * (lf_edlib_common.h: one row per 32 sweep steps).  Shared by lf_edlib_tb_kernel (lf_align.hip: rows and planes in HBM, 64 paths of 64
 * different problems).  gfx950 only.
 *
 * From (n, tl) the path is followed tile by tile: a tile = the 32 sweep steps of one checkpoint row, of the ONE block the cell is
 * in.  The lane takes that block's state in front of the row, the 32 carries the block received during it and the row's 32 target
 * symbols (four loads in flight together), steps once through the tile keeping the state at every part boundary in registers, and
 * then replays the parts that hold a path, right to left: (Pv, Ph) of a part's HK columns go to the lane's own LDS slots and the
 * path is walked through them one move per trip.  Same cells, same Up -> Left -> Diagonal priority
 * (lib/edlib/edlib.cpp:950,984,1015), same ops. */
#ifndef LF_TB_CORE_H
#define LF_TB_CORE_H
#include "lf_edlib_common.h"
#include <type_traits>

template <int I, int N, class F> __device__ __forceinline__ void lf_static_for(F &&f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); lf_static_for<I + 1, N>(f); } }
template <int I, class F> __device__ __forceinline__ void lf_static_rfor(F &&f) { if constexpr (I >= 0) { f(std::integral_constant<int, I>{}); lf_static_rfor<I - 1>(f); } }

/* `ck`, `planes`: this problem's checkpoint rows and bit planes in HBM; s_peq: the lane's private table of match masks (rebuilt when the path changes block). */
template <int HK>
__device__ __forceinline__ void lf_tb_core(const lf_aln_prob &pr, const bool want, const uint32_t tl, const int lane0, const int lane,
                                           const lf_hist_t *ck, const uint64_t *planes, uint64_t *s_peq, ulonglong2 *s_tile,
                                           const uint8_t *__restrict__ pac, const int64_t pac_syms, uint8_t *__restrict__ ops, uint32_t &path_len)
{
    constexpr int K = LF_RSTEPS, ROW = LF_RROW, NPART = K / HK;
    auto qget = [&](uint32_t) -> unsigned char { return 0; };          /* (lf_eq_tok's byte path is never taken with 2-bit targets) */
    const uint32_t n = pr.n, m = pr.m;
    const int dt = (pr.flags & LF_F_TREV) ? -1 : 1; const bool ct = (pr.flags & LF_F_TCOMP) != 0;
    lf_emitter em; em.init(ops + pr.ops_off, n + m, want);
    uint32_t r = want ? n : 0, c = want ? tl : 0;
    uint64_t lo = 0, hi = 0, valid = 0;
    const int pq = lane;                                                /* where the current block's match masks are: s_peq[code * 64 + pq] */
    auto load_planes = [&](uint32_t b) {
        const int ln = lane0 + (int)b; lo = planes[ln]; hi = planes[64 + ln]; valid = planes[128 + ln];
#pragma unroll
        for (uint32_t cde = 0; cde < 4; cde++) s_peq[cde * 64 + lane] = lf_eq_tok<true>(cde, lo, hi, valid, qget, n, b);      /* the lane's own slots: no barrier */
    };
    uint32_t cur_b = r > 0 ? (r - 1) >> 6 : 0;
    load_planes(cur_b);
    while (lf_any(r > 0 && c > 0)) {
        const bool act = r > 0 && c > 0;
        const uint32_t b = act ? (r - 1) >> 6 : 0;
        /* the tile: the K steps of row j for block b; step k works on column cbase + k */
        const uint32_t j = !act ? 0 : (c - 1 + b) / K;
        const int cbase = (int)(j * K) - (int)b + 1;
        /* the block's state in front of the row, the carries it received during the row and the row's 32 target symbols: four loads in
         * flight together, from clamped addresses */
        const lf_hist_t *row = ck + (size_t)j * ROW;
        const lf_hist_t est = row[lane0 + (int)b];
        const uint64_t craw = reinterpret_cast<const uint64_t *>(row + 64)[lane0 + (int)b];
        const uint32_t tokA = lf_pac16(pac, pr.tstart + (int64_t)dt * ((int64_t)cbase - 1), dt, ct, pac_syms);
        const uint32_t tokB = lf_pac16(pac, pr.tstart + (int64_t)dt * ((int64_t)cbase + 15), dt, ct, pac_syms);
        const uint64_t tok = (uint64_t)tokA | ((uint64_t)tokB << 32);
        const uint64_t cw = b > 0 ? craw : 0x5555555555555555ull;                /* block 0: +1 enters every column */
        auto in_part = [&](int p) -> bool { const int h0 = cbase + p * HK; return act && r > 0 && c > 0 && ((r - 1) >> 6) == b && (int)c >= h0 && (int)c < h0 + HK; };
        /* steps FROM .. TO - 1 of the tile; STORE: (Pv, Ph) of the columns go to the lane's own LDS slots (no barrier).  The test for columns in
         * front of the block's first one (col < 1: only in the first tile of a block below the first) is compiled in only when some lane needs it */
        auto replay = [&](auto from_c, auto to_c, auto store_c, auto guard_c, uint64_t &Pv, uint64_t &Mv) {
            constexpr int FROM = decltype(from_c)::value, TO = decltype(to_c)::value; constexpr bool STORE = decltype(store_c)::value, GUARD = decltype(guard_c)::value;
#pragma unroll
            for (int k = FROM; k < TO; k++) {
                const uint32_t tk = (uint32_t)(tok >> (2 * k)) & 3u;
                const uint64_t Eq = s_peq[tk * 64 + pq];
                uint64_t nPv = Pv, nMv = Mv, ph, mh;
                (void)lf_myers_step(nPv, nMv, Eq, (uint32_t)(cw >> (2 * k)) & 3u, ph, mh);
                if (GUARD) { const bool v = cbase + k >= 1; Pv = v ? nPv : Pv; Mv = v ? nMv : Mv; }       /* the block starts at column 1 */
                else { Pv = nPv; Mv = nMv; }
                if (STORE) s_tile[(k - FROM) * 64 + lane] = make_ulonglong2(Pv, ph);
            }
        };
        /* walk through part p: ONE MOVE per trip for every lane that is still inside the part (its column's (Pv, Ph) comes out of the
         * lane's LDS slots).  Unrolled over the columns, with the Up moves of a column in an inner loop, the wavefront paid every column's
         * longest Up run among its 64 paths: ~2.2 trips per column for ~1.05 moves per path. */
        auto walk = [&](int p) {
            const int h0 = cbase + p * HK;
            const int cmin = h0 < 1 ? 1 : h0;
            bool inh = in_part(p);
            while (lf_any(inh)) {
                if (inh) {
                    const int k = (int)c - h0;
                    const ulonglong2 pp = s_tile[k * 64 + lane];
                    const int bit = (int)((r - 1) & 63);
                    const uint32_t up = (uint32_t)(pp.x >> bit) & 1u, lf = ((uint32_t)(pp.y >> bit) & 1u) & ~up, dg = (up | lf) ^ 1u;
                    uint32_t op = up ? 1u : (lf ? 2u : 0u);      /* Up -> Left -> Diagonal (lib/edlib/edlib.cpp:950,984,1015) */
                    /* match or mismatch of a diagonal move: both bases are in registers -- the query's row in the block's bit planes, the
                     * target's column in the tile's symbol word */
                    const uint32_t tcode = (uint32_t)(tok >> (2 * (p * HK + k))) & 3u;
                    const uint32_t same = (uint32_t)(valid >> bit) & ~((uint32_t)(lo >> bit) ^ tcode) & ~((uint32_t)(hi >> bit) ^ (tcode >> 1)) & 1u;
                    op = (dg & (same ^ 1u)) ? 3u : op;
                    em.put(op);
                    r -= up | dg; c -= lf | dg;
                    inh = r > 0 && (int)c >= cmin && ((r - 1) >> 6) == b;      /* still in this block and this part */
                }
            }
        };
        /* The parts are walked right to left.  The state in front of part p is the tile's start state stepped through parts 0 .. p - 1:
         * those boundary states are made ONCE per tile (up to the rightmost part some path of the wavefront is in) and kept in registers,
         * then every part that holds a path is replayed from its own boundary. */
        auto tile = [&](auto guard_c) {
            int pmax = -1;
#pragma unroll
            for (int p = NPART - 1; p >= 0; p--) if (pmax < 0 && lf_any(in_part(p))) pmax = p;
            if (pmax < 0) return;
            uint64_t sPv[NPART], sMv[NPART];
            sPv[0] = est.pv; sMv[0] = est.ph;
            uint64_t Pv = est.pv, Mv = est.ph;
            lf_static_for<0, NPART - 1>([&](auto p_c) {
                constexpr int p = decltype(p_c)::value;
                if (p < pmax) { replay(std::integral_constant<int, p * HK>{}, std::integral_constant<int, (p + 1) * HK>{}, std::false_type{}, guard_c, Pv, Mv); sPv[p + 1] = Pv; sMv[p + 1] = Mv; }
            });
            lf_static_rfor<NPART - 1>([&](auto p_c) {
                constexpr int p = decltype(p_c)::value;
                if (p <= pmax && lf_any(in_part(p))) {
                    uint64_t qPv = sPv[p], qMv = sMv[p];
                    replay(std::integral_constant<int, p * HK>{}, std::integral_constant<int, (p + 1) * HK>{}, std::true_type{}, guard_c, qPv, qMv);
                    walk(p);
                }
            });
        };
        if (lf_any(act && cbase < 1)) tile(std::true_type{}); else tile(std::false_type{});
        {   /* the path climbed into the block above: its planes are requested now, used by the next tile */
            const uint32_t nb = r > 0 ? (r - 1) >> 6 : 0;
            if (nb != cur_b) { load_planes(nb); cur_b = nb; }
        }
    }
    if (want) {
        while (c > 0) { em.put(2); c--; }
        while (r > 0) { em.put(1); r--; }
        em.flush();
    }
    path_len = want ? n + m - em.w : 0;            /* ops are END-aligned: o[cap - len .. cap) */
}
#endif
