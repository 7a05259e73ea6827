/* lf_edlib_common.h -- device helpers shared by the edlib kernels (lf_align.hip: size-class forward / traceback kernels;
 * lf_hirsch.hip: the breadth-first Hirschberg levels).  gfx950 only. */
#ifndef LF_EDLIB_COMMON_H
#define LF_EDLIB_COMMON_H
#include "lf_gpu_common.h"

struct lf_aln_prob {
    int64_t  qstart, tstart; /* element 0 of query / target: byte index (ASCII buffers) or pac coordinate */
    uint64_t ops_off;        /* output ops region (capacity n + m) */
    uint64_t hist_base;      /* 16-byte entries; wave-transposed (template classes) or private (generic) */
    uint64_t aux_off;        /* generic kernel: private state words */
    uint32_t n, m;
    uint32_t id;             /* original problem index */
    uint8_t  mode, task, flags, pad;
};
/* flags: how element i of a sequence is fetched -- index start +/- i, optionally complemented.  Requests of the
 * mapping pipeline are DESCRIPTORS into the read batch and the 2-bit reference already resident in HBM
 * (no byte staging, no H2D of sequences); the stage API uploads byte strings and uses the same accessors. */

struct lf_hist_t { uint64_t pv, ph; };

/* one Myers block step. Pv/Mv in-out.  The horizontal delta entering / leaving the block travels as two bits:
 * bit 0 = +1, bit 1 = -1 (no compares, no sign handling on the per-step dependency chain).  ph_out / mh_out =
 * horizontal +1 / -1 bits of the block's rows (unshifted). */
__device__ __forceinline__ uint32_t lf_myers_step(uint64_t &Pv, uint64_t &Mv, uint64_t Eq, uint32_t hin, uint64_t &ph_out, uint64_t &mh_out)
{
    const uint64_t hpos = hin & 1u, hneg = hin >> 1;
    const uint64_t Xv = Eq | Mv;
    Eq |= hneg;
    const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    uint64_t Ph = Mv | ~(Xh | Pv);
    uint64_t Mh = Pv & Xh;
    ph_out = Ph; mh_out = Mh;
    const uint32_t hout = (uint32_t)(Ph >> 63) | ((uint32_t)(Mh >> 63) << 1);
    Ph = (Ph << 1) | hpos;
    Mh = (Mh << 1) | hneg;
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    return hout;
}
#define LF_HIN_PLUS1 1u          /* first block of a column: the row above the matrix grows by one per column */
/* +1 / 0 / -1 of a two-bit delta at bit `bit` of (ph, mh) */
__device__ __forceinline__ int lf_delta_at(uint64_t ph, uint64_t mh, int bit) { return (int)((ph >> bit) & 1) - (int)((mh >> bit) & 1); }

/* bit planes of 64 query bytes: uppercase A,C,G,T -> (lo,hi) code + valid; anything else never equals a
 * target base (edlib compares raw bytes, lib/edlib/edlib.cpp:1367-1384; the target comes from the 2-bit
 * reference and is upper case).  Branch-free (lf_code_upper). */
__device__ __forceinline__ void lf_plane_add(unsigned char ch, int bit, uint64_t &lo, uint64_t &hi, uint64_t &valid)
{
    bool ok;
    const uint32_t c = lf_code_upper(ch, ok);
    const uint64_t v = ok ? 1ull : 0ull;
    lo |= (v & (c & 1u)) << bit; hi |= (v & (c >> 1)) << bit; valid |= v << bit;
}

/* A target symbol inside the DP loops is a 32-bit token: bits 1:0 = code, bit 8 = "not one of ACGT", bits 23:16 = the raw
 * byte.  Targets of the mapping pipeline come from the 2-bit reference: always a bare code (template PAC). */
__device__ __forceinline__ uint32_t lf_tok_of_byte(unsigned char ch)
{
    bool ok;
    const uint32_t c = lf_code_upper(ch, ok);
    return c | (ok ? 0u : 0x100u) | ((uint32_t)ch << 16);
}
template <bool PAC>
__device__ __forceinline__ uint32_t lf_tok(const lf_tacc &T, uint32_t i) { return PAC ? T.pac_code(i) : lf_tok_of_byte(T.get(i)); }
/* Eq mask of a token against a block: three ops per word from the bit planes; a byte outside ACGT takes the exact
 * compare (general alphabets of the stage API; never on the pipeline path) */
template <bool PAC, class QG>
__device__ __forceinline__ uint64_t lf_eq_tok(uint32_t tok, uint64_t lo, uint64_t hi, uint64_t valid, const QG &qget, uint32_t n, uint32_t blk)
{
    const uint64_t slo = 0ull - (uint64_t)(tok & 1u), shi = 0ull - (uint64_t)((tok >> 1) & 1u);
    uint64_t e = ~((lo ^ slo) | (hi ^ shi)) & valid;
    if (!PAC) {
        if (tok & 0x100u) {
            const unsigned char tc = (unsigned char)(tok >> 16);
            e = 0;
            for (uint32_t i = 0; i < 64; i++) {
                const uint32_t r = blk * 64 + i;
                if (r < n && qget(r) == tc) e |= 1ull << i;
            }
        }
    }
    return e;
}
/* the generic kernel's form: raw byte */
__device__ __forceinline__ uint64_t lf_eq_mask(unsigned char tc, uint64_t lo, uint64_t hi, uint64_t valid,
                                               const lf_qacc &Q, uint32_t n, uint32_t blk)
{
    auto qg = [&](uint32_t r) -> unsigned char { return Q.get(r); };
    return lf_eq_tok<false>(lf_tok_of_byte(tc), lo, hi, valid, qg, n, blk);
}

/* edlib's own leaf / Hirschberg switch (lib/edlib/edlib.cpp:1117-1119), callable on the device */
__host__ __device__ __forceinline__ bool lf_leaf(int64_t n, int64_t m) { return 20LL * ((n + 63) / 64) * m + 8LL * m < 1024 * 1024; }
/* ================================================================================================
 * TRACEBACK WITHOUT A HISTORY STREAM
 *
 * Round 1 wrote two bits per DP cell (Pv, Ph: 16 B per column and 64-row block) to HBM and read them back:
 * 140 GB per 100 k reads for 4.4 GB of algorithmic bytes.  Now the forward pass keeps only CHECKPOINTS -- the
 * bit-vector state (Pv, Mv) of every block every K columns (lane classes) or every K sweep steps (group / wave
 * classes) -- and the traceback walks the matrix tile by tile from the end: the tile's K columns (steps) are
 * recomputed from the checkpoint in front of it, their (Pv, Ph) words go to LDS, the path is followed through
 * the tile, then the next tile to the left.  HBM sees 1/K of the old stream, the DP work doubles (integer ALU,
 * of which the old kernels used ~5 %), results are identical: the same cells are visited with the same
 * Up -> Left -> Diagonal priority (lib/edlib/edlib.cpp:950,984,1015).
 * ================================================================================================ */

#define LF_LANE_K   8        /* lane classes: columns per tile (tile in LDS: K x W blocks x 64 lanes x 16 B) */
#ifndef LF_LANE_W
#define LF_LANE_W   1        /* 8 KiB of LDS per wave: LDS, not registers, bounds the waves per SIMD of these kernels */
#endif

/* lf_wave_max_u32: lf_gpu_common.h (DPP row shifts + broadcasts; the same value in every lane, so loop bounds live in SGPRs) */
__device__ __forceinline__ int lf_wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(v, o); v = x > v ? x : v; }
    return __builtin_amdgcn_readfirstlane(v);
}

/* ops are produced back to front and leave in 8-byte words: bytes are shifted into `acc` and stored when the (descending)
 * address reaches an 8-byte boundary; only the bytes above the first boundary and below the last one are single-byte
 * stores (the neighbouring problems' regions start right there).  `store` = this lane owns the output. */
struct lf_emitter {
    uint8_t *o; uint32_t w, mis, cnt; uint64_t acc; bool store;
    __device__ __forceinline__ void init(uint8_t *base, uint32_t cap, bool st) { o = base; w = cap; mis = (uint32_t)((uintptr_t)base & 7); cnt = 0; acc = 0; store = st; }
    __device__ __forceinline__ void spill()
    {   /* the collected bytes start at o + w, lowest address in the lowest byte */
        if (store) {
            if (cnt == 8) *reinterpret_cast<uint64_t *>(o + w) = acc;
            else for (uint32_t i = 0; i < cnt; i++) o[w + i] = (uint8_t)(acc >> (8 * i));
        }
        cnt = 0;
    }
    __device__ __forceinline__ void put(uint32_t op)
    {
        --w;
        acc = (acc << 8) | op; cnt++;
        if (((mis + w) & 7) == 0) spill();
    }
    __device__ __forceinline__ void flush() { if (cnt) spill(); }
};


/* layout of a wave's checkpoint area (16-byte entries from the hist_base of the wave's first problem):
 *   lane classes   [0, 96 NB)  bit planes lo / hi / valid of every block: u64 [(b * 3 + x) * 64 + lane]
 *                  then        carries: the two-bit horizontal deltas ENTERING block b >= 1, 16 columns per u32,
 *                              u32 [((q * NB + b) * 64 + lane) * 4 + sub], column c in word (c - 1) / 16 = 4 q + sub
 *                  then        checkpoints (Pv, Mv) after every K-th column: [((j - 1) * NB + b) * 64 + lane]
 *   sweep classes  [0, 96)     bit planes of the lanes' blocks: u64 [x * 64 + lane]          (KB = 1 only)
 *                  then        rows of 64 KB + 16 entries (KB = 1; + 4 otherwise), one per K steps: (Pv, Mv) of every lane,
 *                              tail bytes [0, 64) pending carry of every lane, u16 [32 + lane] the carries the lane
 *                              RECEIVED during the row's K steps (KB = 1)
 * With the carries a block can be replayed on its own: lf_edlib_tb_kernel walks one path per lane and recomputes only
 * the block the path is in. */
#define LF_PLANE_ENTRIES 96
__host__ __device__ __forceinline__ int lf_sweep_row(int kb) { return 64 * kb + (kb == 1 ? 16 : 4); }
__host__ __device__ __forceinline__ uint64_t lf_lane_ck_off(int nb, uint32_t m_max) { return (uint64_t)nb * LF_PLANE_ENTRIES + (uint64_t)((m_max + 63) >> 6) * nb * 64; }


/* sixteen target symbols x0, x0 + dir, ... as one word, symbol k at bits [2k + 1 : 2k].  A lane that has not reached column 1
 * yet asks for positions in front of its target; where those lie outside the array (the first / last symbols of the
 * reference, the first problem of an uploaded buffer) the window is taken at the array's edge and shifted, so that the
 * symbols that do exist keep their places. */
__device__ __forceinline__ uint32_t lf_pac16(const uint8_t *__restrict__ pac, int64_t x0, int dir, bool comp, int64_t n_syms)
{
    int64_t xs = x0; uint32_t pre = 0;
    uint32_t v;
    if (dir > 0) {
        if (xs < 0) { pre = (uint32_t)(-xs); xs = 0; }
        if (xs > n_syms - 1) xs = n_syms - 1;                 /* past the end: every column of the window is past the target's end too */
        uint64_t raw; __builtin_memcpy(&raw, pac + (xs >> 2), 8);
        uint64_t B = __builtin_bswap64(raw);                  /* symbol order = bit order, first symbol on top */
        B <<= 2 * (uint32_t)(xs & 3);
        uint32_t r = __brev((uint32_t)(B >> 32));              /* first symbol at the bottom, the two bits of each symbol swapped */
        v = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    } else {
        if (xs > n_syms - 1) { pre = (uint32_t)(xs - (n_syms - 1)); xs = n_syms - 1; }
        if (xs < 0) xs = 0;
        int64_t xl = xs - 15; if (xl < 0) xl = 0;
        const int64_t byteL = xl >> 2;
        uint64_t raw; __builtin_memcpy(&raw, pac + byteL, 8);
        const uint64_t B = __builtin_bswap64(raw);
        v = (uint32_t)(B >> (62 - 2 * (uint32_t)(xs - 4 * byteL)));      /* symbol xs at the bottom, xs - 1 above it, ... */
    }
    v = pre >= 16 ? 0u : v << (2 * pre);
    return comp ? ~v : v;
}

/* checkpoint row of the one-block-per-lane forward kernel (lf_rsweep.hip): ONE ROW PER 32 SWEEP STEPS = 96 16-byte units: 64 x (Pv, Mv)
 * in FRONT of the row's first step, then 64 x u64 = the 32 two-bit carries every lane received during the row.  Both parts are written
 * coalesced (a wavefront stores 1 KiB + 2 x 256 B per 32 steps: 48 bytes per step instead of 84), and a path's tile is two 32-byte
 * sectors of ONE row: state and carries of its block.  (Rounds 2 - 4: a row per 16 steps with the state BEHIND it, so that a tile
 * needed two rows; a 32-byte entry per lane was tried in round 5 and cost the forward pass its coalesced stores.) */
#define LF_RROW 96           /* 16-byte units per row */
#define LF_RSTEPS 32

/* SMALL problems: at most LF_SMALL_NB query blocks and LF_SMALL_M target columns, i.e. at most 128 sweep steps = four checkpoint rows.  More
 * than half of a C2 step's problems (profiles/r05_search/alignment_problem_histogram_20k_reads.txt) -- their rows never leave the
 * CU: the forward pass keeps them in LDS and the same wavefront walks the paths (lf_edlib_small_kernel, lf_rsweep.hip).  In the
 * binning they form segments of their own: blocks + 64 in the key's block field. */
#define LF_SMALL_NB 2
#define LF_SMALL_M 127
#define LF_SMALL_ROWS 4
__host__ __device__ __forceinline__ bool lf_small_prob(uint32_t n, uint32_t m) { return n <= 64u * LF_SMALL_NB && m <= LF_SMALL_M; }
__host__ __device__ __forceinline__ int lf_seg_blocks(int nbf) { return nbf > 64 ? nbf - 64 : nbf; }      /* key's block field -> lanes per problem */
#define LF_SEG_NB_MAX (64 + LF_SMALL_NB)

/* 64 consecutive bits of a plane starting at (signed) bit position p0; positions outside [0, 64 n_words) read as garbage
 * inside the array (the caller masks them) */
__device__ __forceinline__ uint64_t lf_bits64(const uint64_t *__restrict__ a, int64_t p0, int64_t n_words)
{
    const int64_t w = p0 >> 6;                                      /* floor: -1 for a window that starts before the buffer */
    const uint32_t sh = (uint32_t)(p0 & 63);
    const int64_t i0 = w < 0 ? 0 : (w > n_words - 1 ? n_words - 1 : w), i1 = w + 1 < 0 ? 0 : (w + 1 > n_words - 1 ? n_words - 1 : w + 1);
    uint64_t a0 = a[i0], a1 = a[i1];                                /* unconditional loads from clamped indices */
    a0 = (w >= 0 && w <= n_words - 1) ? a0 : 0ull; a1 = (w + 1 >= 0 && w + 1 <= n_words - 1) ? a1 : 0ull;
    return sh ? (a0 >> sh) | (a1 << (64 - sh)) : a0;
}
__device__ __forceinline__ uint64_t lf_brev64(uint64_t x) { return ((uint64_t)__brev((uint32_t)x) << 32) | (uint64_t)__brev((uint32_t)(x >> 32)); }

/* lane l receives lane l-1's value (lane 0: 0): the horizontal carry of the anti-diagonal sweeps.  DPP wave_shr:1 is a
 * VALU move; __shfl_up goes through the LDS crossbar (ds_bpermute) and sits on the per-step dependency chain. */
__device__ __forceinline__ uint32_t lf_wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }

#endif
