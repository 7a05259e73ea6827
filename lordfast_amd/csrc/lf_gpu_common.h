/* lf_gpu_common.h -- shared by the HIP translation units of liblfgpu.so (gfx950 only). */
#ifndef LF_GPU_COMMON_H
#define LF_GPU_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "lf_internal.h"

/* wave64 vote: the comparison's own lane mask.  (HIP's __ballot / __any materialise the predicate as 0 / 1 in a VGPR and
 * compare it with 0 again: two 4-cycle VOP3 instructions per vote in loops that are bound by VALU issue.) */
#define lf_ballot(pred) ((uint64_t)__builtin_amdgcn_ballot_w64((bool)(pred)))
#define lf_any(pred) (__builtin_amdgcn_ballot_w64((bool)(pred)) != 0ull)

/* every checked HIP call leaves its source position in a per-lane slot: LF_WATCHDOG=<seconds> prints them when a batch
 * does not finish (lf_pipeline.c) */
extern "C" void lfg_phase(const char *file, int line);
extern "C" void lfg_phase_dump(void);
#define HIPCHK(expr) do { lfg_phase(__FILE__, __LINE__); hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    lf_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); return LF_ERR_HIP; } } while (0)

/* device view of the FM-index (all pointers in HBM) */
struct lf_dev_index {
    uint64_t primary, L2[5], seq_len;
    int64_t  l_pac;
    uint64_t n_sa;
    const uint32_t *bwt;        /* reference layout: per 128 symbols 4 x u64 Occ + 8 x u32 (lib/bwa/bwt.h:72-73) */
    const uint64_t *sa_sampled; /* every 32nd row (lib/bwa/bwt.c:62-84), sa[0] = -1 */
    const uint8_t  *sa_full;    /* seq_len + 1 rows of FIVE bytes (positions are below 2^33: 40 bits hold them; 31 GB instead of 49.6 GB for a 3.1 Gbp genome), or NULL */
    const uint64_t *cache;      /* 4^12 x (beg,end): SA interval of every 12-mer (src/BWT.cpp:60-115) */
    const uint64_t *cache14;    /* 4^14 x (beg,end), or NULL: two search steps saved per sample when -k >= 14 (4.3 GB; large genomes) */
    const uint64_t *cache16;    /* 4^16 x (beg,end), or NULL (68.7 GB; genomes >= 2^30 symbols when HBM allows): a sample whose 16-mer occurs starts
                                 * there -- four search steps saved; one whose 16-mer does not occur falls back to the narrower table */
    const uint8_t  *pac;
    const uint8_t  *occ2;       /* the BWT in the HBM layout of the mapping kernels (lf_occ2_* below): made IN PLACE out of `bwt` once the
                                 * tables and the full SA are built (bwt is NULL from then on); NULL while the index is being built */
};

/* full suffix array, five bytes per row (little endian); rows of one interval are contiguous: hits of a sample are 5 n consecutive bytes */
#define LF_SA_ROW_BYTES 5
__device__ __forceinline__ uint64_t lf_sa_full_get(const uint8_t *__restrict__ sa, uint64_t row)
{
    uint64_t w; __builtin_memcpy(&w, sa + row * LF_SA_ROW_BYTES, 8);      /* (the array has 8 bytes of slack behind its last row) */
    return w & 0xFFFFFFFFFFull;
}
__device__ __forceinline__ void lf_sa_full_put(uint8_t *__restrict__ sa, uint64_t row, uint64_t v)
{
    uint8_t *p = sa + row * LF_SA_ROW_BYTES;                               /* exactly five bytes: the neighbouring rows belong to other threads */
    const uint32_t lo = (uint32_t)v; __builtin_memcpy(p, &lo, 4); p[4] = (uint8_t)(v >> 32);
}

/* ---- the occurrence blocks as the mapping kernels read them (not bwa's file layout: nothing requires the resident form to equal
 * the one on disk).  64 bytes per 128 symbols, one HBM burst:
 *     u64 C[4]    CUMULATIVE counts in front of the block: C[j] = occurrences of symbols 0 .. j  (count of a = C[a] - C[a-1], of everything
 *                 above a = C[3] - C[a]: a lane fetches the two words its symbol needs instead of four and selecting)
 *     u64 lo[2]   bit i & 63 of word i >> 6 = low code bit of symbol i of the block
 *     u64 hi[2]   ... high code bit
 * "symbol == a" is (lo ^ ~A0) & (hi ^ ~A1), "symbol > a" is (hi & ~A1) | ((hi ^ ~A1) & lo & ~A0) with A0 / A1 = a's bits spread over a
 * word: two masks for 128 symbols in 12 word operations, against eight 16-symbol words with interleaved bits before; the symbols
 * in front of a row are the low bits of a 128-bit mask.  lib/bwa/bwt.h:72-73 (file layout), lib/bwa/bwt.c:98-163 (bwt_occ / bwt_2occ). ---- */
struct lf_occ2_masks { uint64_t e0, e1, g0, g1; };
__device__ __forceinline__ lf_occ2_masks lf_occ2_eq_gt(uint64_t lo0, uint64_t lo1, uint64_t hi0, uint64_t hi1, int a)
{
    const uint64_t nA0 = (a & 1) ? 0ull : ~0ull, nA1 = (a & 2) ? 0ull : ~0ull;
    lf_occ2_masks M;
    const uint64_t t0 = hi0 ^ nA1, t1 = hi1 ^ nA1;                 /* high bit equal */
    M.e0 = (lo0 ^ nA0) & t0; M.e1 = (lo1 ^ nA0) & t1;
    M.g0 = (hi0 & nA1) | (t0 & lo0 & nA0); M.g1 = (hi1 & nA1) | (t1 & lo1 & nA0);
    return M;
}
/* the first rem symbols of a block as a 128-bit mask, 1 <= rem <= 128 */
__device__ __forceinline__ void lf_occ2_first(uint32_t rem, uint64_t &m0, uint64_t &m1)
{
    m0 = rem >= 64u ? ~0ull : ((1ull << (rem & 63u)) - 1ull);
    m1 = rem <= 64u ? 0ull : (rem >= 128u ? ~0ull : ((1ull << ((rem - 64u) & 63u)) - 1ull));
}
/* bwt_occ on the resident layout (one symbol; the locate walk of the sampled-SA variant) */
__device__ __forceinline__ uint64_t lf_occ2(const lf_dev_index &ix, uint64_t k, int c)
{
    if (k == ix.seq_len) return ix.L2[c + 1] - ix.L2[c];
    if (k == ~0ull) return 0;
    k -= (k >= ix.primary);
    const uint64_t *blk = reinterpret_cast<const uint64_t *>(ix.occ2 + ((k >> 7) << 6));
    const uint64_t n = blk[c] - (c ? blk[c - 1] : 0ull);
    const lf_occ2_masks M = lf_occ2_eq_gt(blk[4], blk[5], blk[6], blk[7], c);
    uint64_t m0, m1; lf_occ2_first((uint32_t)(k & 127) + 1u, m0, m1);
    return n + (uint64_t)__popcll(M.e0 & m0) + (uint64_t)__popcll(M.e1 & m1);
}
__device__ __forceinline__ uint64_t lf_inv_psi2(const lf_dev_index &ix, uint64_t k)
{
    if (k == ix.primary) return 0;
    const uint64_t x = k - (k > ix.primary);
    const uint64_t *blk = reinterpret_cast<const uint64_t *>(ix.occ2 + ((x >> 7) << 6));
    const uint32_t i = (uint32_t)(x & 127);
    const int c = (int)(((blk[4 + (i >> 6)] >> (i & 63)) & 1ull) | (((blk[6 + (i >> 6)] >> (i & 63)) & 1ull) << 1));
    return ix.L2[c] + lf_occ2(ix, k, c);
}

struct lf_dev_state {           /* host-side owner of the device allocations */
    lf_dev_index view;
    void *bwt, *sa_sampled, *sa_full, *cache, *cache14, *cache16, *pac;
    void *ctg_names = nullptr, *ctg_name_off = nullptr;      /* contig names for lf_sam.hip (uploaded on first use) */
    hipStream_t stream;
};

int lfg_build_cache_table(const lf_dev_index *v, hipStream_t stream, int K, uint64_t **table);

/* ---- branch-free base classification.  A `switch` on a base compiles to a tree of exec-mask branches (dozens of scalar
 * instructions and s_cbranch per call); these sit inside the DP and search loops, so they are arithmetic instead:
 * for 'A' 'C' 'G' 'T' (0x41 0x43 0x47 0x54) bits 2:1 are 0 1 3 2, and x ^ (x >> 1) turns that into 0 1 2 3. ---- */
__device__ __forceinline__ uint32_t lf_code2(uint32_t ch) { const uint32_t x = (ch >> 1) & 3u; return x ^ (x >> 1); }
/* upper case only (edlib compares raw bytes): ok = ch is one of "ACGT" */
__device__ __forceinline__ uint32_t lf_code_upper(uint32_t ch, bool &ok)
{
    const uint32_t c = lf_code2(ch);
    ok = ch == ((0x54474341u >> (c << 3)) & 0xffu);
    return c;
}
__device__ __forceinline__ int lf_nt4(unsigned char ch)
{   /* nst_nt4_table (lib/bwa/bntseq.c:47-64): A/a C/c G/g T/t -> 0..3, everything else > 3 */
    bool ok;
    const uint32_t c = lf_code_upper((uint32_t)ch & 0xDFu, ok);
    return ok ? (int)c : 4;
}

struct lf_seqs { const unsigned char *q; const unsigned char *t; const uint8_t *pac; };

__device__ __forceinline__ unsigned char lf_rc_char(unsigned char ch)
{   /* tableRev (src/Common.cpp:31-40): complement with the case kept, anything else 'N' */
    bool ok;
    const uint32_t c = lf_code_upper((uint32_t)ch & 0xDFu, ok);
    const uint32_t r = ((0x41434754u >> (c << 3)) & 0xffu) | ((uint32_t)ch & 0x20u);      /* "TGCA"[c], case bit of ch */
    return (unsigned char)(ok ? r : (uint32_t)'N');
}
/* element i of a sequence = base[start +/- i], optionally complemented (flags LF_F_*): requests of the mapping pipeline
 * are DESCRIPTORS into the read batch and the 2-bit reference already resident in HBM */
struct lf_qacc {
    const unsigned char *b; int64_t start; int dir; bool comp;
    __device__ __forceinline__ lf_qacc(const unsigned char *base, int64_t st, unsigned flags) : b(base), start(st), dir((flags & LF_F_QREV) ? -1 : 1), comp(flags & LF_F_QCOMP) {}
    __device__ __forceinline__ unsigned char get(uint32_t i) const { const unsigned char c = b[start + (int64_t)dir * (int64_t)i]; return comp ? lf_rc_char(c) : c; }
};
struct lf_tacc {
    const unsigned char *b; const uint8_t *pac; int64_t start; int dir; bool comp, is_pac;
    __device__ __forceinline__ lf_tacc(const unsigned char *base, const uint8_t *pc, int64_t st, unsigned flags) : b(base), pac(pc), start(st), dir((flags & LF_F_TREV) ? -1 : 1), comp(flags & LF_F_TCOMP), is_pac(flags & LF_F_TPAC) {}
    /* 2-bit code straight from the packed reference (targets of the mapping pipeline) */
    __device__ __forceinline__ uint32_t pac_code(uint32_t i) const {
        const int64_t x = start + (int64_t)dir * (int64_t)i;
        const uint32_t c = ((uint32_t)pac[x >> 2] >> ((~(uint32_t)x & 3u) << 1)) & 3u;
        return comp ? 3u - c : c;
    }
    /* the 2-bit codes of elements i0 .. i0 + 7, two bits each (element i0 + k at bits 2k): ONE unaligned 4-byte load instead of
     * eight dependent byte loads.  Only elements in [0, count) are meaningful; the load address is clamped to the bytes that
     * hold those (the array has 16 bytes of padding behind its last symbol), the others come out as garbage. */
    __device__ __forceinline__ uint32_t pac_codes8(int64_t i0, uint32_t count) const {
        const int64_t xa = start + (int64_t)dir * i0, xb = xa + 7 * (int64_t)dir;
        const int64_t vlo = dir > 0 ? start : start - ((int64_t)count - 1), vhi = dir > 0 ? start + ((int64_t)count - 1) : start;
        int64_t xlo = xa < xb ? xa : xb;
        if (xlo > vhi) xlo = vhi;
        if (xlo < vlo - 7) xlo = vlo - 7;
        if (xlo < 0) xlo = 0;
        uint32_t w; __builtin_memcpy(&w, pac + (xlo >> 2), 4);
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int64_t x = xa + (int64_t)dir * k;
            const uint32_t sh = (uint32_t)(((x >> 2) - (xlo >> 2)) * 8) + ((~(uint32_t)x & 3u) << 1);
            const uint32_t c = (w >> (sh & 31u)) & 3u;
            out |= (comp ? 3u - c : c) << (2 * k);
        }
        return out;
    }
    __device__ __forceinline__ unsigned char get(uint32_t i) const {
        if (is_pac) return (unsigned char)(0x54474341u >> (pac_code(i) << 3));   /* "ACGT"[c] without a table load */
        const unsigned char c = b[start + (int64_t)dir * (int64_t)i];
        return comp ? lf_rc_char(c) : c;
    }
};

/* symbols equal to c among the first r symbols (r <= 0: none, r >= 32: all) of a 32-symbol chunk held
 * MSB-first in y */
__device__ __forceinline__ uint32_t lf_count_sym(uint32_t hi, uint32_t lo, int c, int r)
{
    const uint64_t y = ((uint64_t)hi << 32) | lo;
    const uint64_t rep = (c & 1 ? 0x5555555555555555ull : 0ull) | (c & 2 ? 0xAAAAAAAAAAAAAAAAull : 0ull);
    const uint64_t eq = ~(y ^ rep);
    uint64_t m = eq & (eq >> 1) & 0x5555555555555555ull;
    const uint64_t keep = r >= 32 ? ~0ull : (r <= 0 ? 0ull : (~0ull << (64 - 2 * r)));
    return (uint32_t)__popcll(m & keep);
}

/* Occ(k, c): occurrences of c in B[0..k] (bwt_occ, lib/bwa/bwt.c:107-127). One 64-byte block. */
__device__ __forceinline__ uint64_t lf_occ(const lf_dev_index &ix, uint64_t k, int c)
{
    if (k == ix.seq_len) return ix.L2[c + 1] - ix.L2[c];
    if (k == ~0ull) return 0;
    k -= (k >= ix.primary);
    const uint32_t *blk = ix.bwt + ((k >> 7) << 4);
    uint64_t n = reinterpret_cast<const uint64_t *>(blk)[c];
    const uint4 w0 = *reinterpret_cast<const uint4 *>(blk + 8);
    const uint4 w1 = *reinterpret_cast<const uint4 *>(blk + 12);
    const int rem = (int)(k & 127) + 1;
    n += lf_count_sym(w0.x, w0.y, c, rem) + lf_count_sym(w0.z, w0.w, c, rem - 32)
       + lf_count_sym(w1.x, w1.y, c, rem - 64) + lf_count_sym(w1.z, w1.w, c, rem - 96);
    return n;
}

/* one backward-search step: [k,l] -> interval of c.P (src/BWT.cpp:290-293). touches counted like
 * bwt_2occ (lib/bwa/bwt.c:132-139): 1 block if k-1 and l share one, else 2 */
__device__ __forceinline__ void lf_backward_step(const lf_dev_index &ix, uint64_t &k, uint64_t &l, int c, uint32_t &blk_touches)
{
    const uint64_t km = k - 1;
    const uint64_t _k = (km >= ix.primary) ? km - 1 : km, _l = (l >= ix.primary) ? l - 1 : l;
    blk_touches += (km == ~0ull || l == ~0ull || (_k >> 7) != (_l >> 7)) ? 2u : 1u;
    const uint64_t ok = lf_occ(ix, km, c), ol = lf_occ(ix, l, c);
    k = ix.L2[c] + ok + 1;
    l = ix.L2[c] + ol;
}

/* Occ(k, c) for all four symbols at once (bwt_occ4, lib/bwa/bwt.c:169-186): the same 64-byte block, four counts */
__device__ __forceinline__ void lf_occ4(const lf_dev_index &ix, uint64_t k, uint64_t cnt[4])
{
    if (k == ~0ull) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
    if (k == ix.seq_len) { for (int c = 0; c < 4; c++) cnt[c] = ix.L2[c + 1] - ix.L2[c]; return; }
    k -= (k >= ix.primary);
    const uint32_t *blk = ix.bwt + ((k >> 7) << 4);
    const uint4 w0 = *reinterpret_cast<const uint4 *>(blk + 8);
    const uint4 w1 = *reinterpret_cast<const uint4 *>(blk + 12);
    const ulonglong2 c01 = *reinterpret_cast<const ulonglong2 *>(blk), c23 = *reinterpret_cast<const ulonglong2 *>(blk + 4);
    const int rem = (int)(k & 127) + 1;
    cnt[0] = c01.x; cnt[1] = c01.y; cnt[2] = c23.x; cnt[3] = c23.y;
#pragma unroll
    for (int c = 0; c < 4; c++)
        cnt[c] += lf_count_sym(w0.x, w0.y, c, rem) + lf_count_sym(w0.z, w0.w, c, rem - 32)
                + lf_count_sym(w1.x, w1.y, c, rem - 64) + lf_count_sym(w1.z, w1.w, c, rem - 96);
}

/* Bidirectional step (bwt_extend with is_back = 0, lib/bwa/bwt.c:262-275).  The text is forward + reverse complement,
 * so a pattern P and revcomp(P) have intervals of the same size s; (x0, x1, s) = (first row of P, first row of
 * revcomp(P), size).  Appending base c to P = prepending (3 - c) to revcomp(P): one backward step on x1 with all four
 * counts, and x0 moves past the rows of the patterns P.d with d > c.  Returns false (state untouched) if P.c does not occur. */
__device__ __forceinline__ bool lf_extend_right(const lf_dev_index &ix, uint64_t &x0, uint64_t &x1, uint64_t &s, int c, uint32_t &blk_touches)
{
    const uint64_t km = x1 - 1, l = x1 - 1 + s;
    const uint64_t _k = (km >= ix.primary) ? km - 1 : km, _l = (l >= ix.primary) ? l - 1 : l;
    blk_touches += (km == ~0ull || l == ~0ull || (_k >> 7) != (_l >> 7)) ? 2u : 1u;
    uint64_t tk[4], tl[4];
    lf_occ4(ix, km, tk); lf_occ4(ix, l, tl);
    const int a = 3 - c;                                   /* symbol prepended on the complement strand */
    const uint64_t ns = tl[a] - tk[a];
    if (ns == 0) return false;
    /* rows of P.d for d = 0..3 follow each other inside P's interval in the order of d; on the complement strand d
     * appears as 3 - d, so P.c starts after the sentinel row (if P's complement interval spans it) and all a' > a */
    uint64_t nx0 = x0 + ((x1 <= ix.primary && x1 + s - 1 >= ix.primary) ? 1u : 0u);
    for (int j = 3; j > a; j--) nx0 += tl[j] - tk[j];
    x0 = nx0; x1 = ix.L2[a] + 1 + tk[a]; s = ns;
    return true;
}

/* bwt_B0 (lib/bwa/bwt.h:78) + bwt_invPsi (lib/bwa/bwt.c:53-59) */
__device__ __forceinline__ uint64_t lf_inv_psi(const lf_dev_index &ix, uint64_t k)
{
    if (k == ix.primary) return 0;
    const uint64_t x = k - (k > ix.primary);
    const uint32_t w = ix.bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)];
    const int c = (int)((w >> ((~x & 15) << 1)) & 3);
    return ix.L2[c] + lf_occ(ix, k, c);
}

/* bwt_sa (lib/bwa/bwt.c:86-96) by LF walk to the next sampled row */
__device__ __forceinline__ uint64_t lf_sa_walk(const lf_dev_index &ix, uint64_t k, uint32_t &steps)
{
    uint64_t off = 0;
    if (ix.occ2) while (k & 31) { ++off; k = lf_inv_psi2(ix, k); }      /* the resident layout (mapping); bwa's file layout while an index is built */
    else while (k & 31) { ++off; k = lf_inv_psi(ix, k); }
    steps += (uint32_t)off;
    return off + ix.sa_sampled[k >> 5];
}

/* wavefront maximum of an unsigned, in every lane: DPP row shifts + row broadcasts (no LDS), then one readlane.
 * PRECONDITION: all 64 lanes active (EXEC = ~0) -- a DPP move reads the registers of lanes that sit out a branch as they are (stale),
 * and lane 63 must hold the reduction.  Call it outside divergent control flow (every caller does: loop bounds at the top of a
 * kernel / of a wave-uniform loop); inside a branch use a ballot-based form instead. */
__device__ __forceinline__ uint32_t lf_wave_max_u32(uint32_t v)
{
    uint32_t x = v;
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));      /* row_shr:1 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));      /* row_shr:2 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));      /* row_shr:4 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));      /* row_shr:8 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));      /* row_bcast:15 -> rows 1, 3 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));      /* row_bcast:31 -> rows 2, 3 */
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

/* host side: every wait for a stream outside lf_mem.hip polls the calling thread's own event for LF_SPIN_US microseconds
 * (default 3000) and then SLEEPS on it (the event is created with hipEventBlockingSync, lf_mem.hip).  Measured on config C2,
 * HBM-resident step / host CPU per step: runtime's own spinning wait 69.2 ms / 0.57 core-s, sleep at once 77.5 ms / 0.18 -- a
 * sleeping thread is back ~0.2 ms after its stream is done, and a chunk waits ~40 times.  Most of those waits are for a few
 * microseconds of copy or a small kernel: polling briefly catches them, the long ones sleep (profiles/r04_waits/: 100 us -> 70.5 ms /
 * 0.20, 3000 us -> 67.7 ms / 0.34 with eight lanes).  Round 5, four lanes: 100 / 300 / 1000 us -> 64.1 / 63.7 - 66.0 / 65.0 ms at 0.145 / 0.15 - 0.165 /
 * 0.187 core-s (profiles/r05_lanes/): the default is 200 us.  LF_SPIN_WAIT=1: the runtime's wait. */
#ifndef LF_NO_SYNC_WRAP
#include <stdlib.h>
#include <time.h>
extern "C" void *lfg_thread_wait_event(int device);
extern "C" void lfg_count_wait(void);
extern "C" void lfg_count_wait_at(const char *file, int line);        /* LF_WAIT_TRACE=1: waits per call site, printed at exit */
static inline hipError_t lf_stream_wait(hipStream_t s, const char *file, int line)
{
    lfg_count_wait();
    static const bool trace = lf_env_set("LF_WAIT_TRACE") != 0;
    if (trace) lfg_count_wait_at(file, line);
    const bool spin = false;
    static const long spin_us = lf_env_long("LF_SPIN_US", 200);
    int dev = -1;
    if (spin || hipGetDevice(&dev) != hipSuccess) return hipStreamSynchronize(s);
    hipEvent_t e = (hipEvent_t)lfg_thread_wait_event(dev);
    if (!e || hipEventRecord(e, s) != hipSuccess) { (void)hipGetLastError(); return hipStreamSynchronize(s); }
    if (spin_us > 0) {
        struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000L >= spin_us) break;
            __builtin_ia32_pause();
        }
    }
    return hipEventSynchronize(e);
}
#define hipStreamSynchronize(s) lf_stream_wait(s, __FILE__, __LINE__)
#endif

#endif
