/* lf_hirsch.h -- breadth-first Hirschberg on the device (lf_hirsch.hip), driven by lf_align.hip. */
#ifndef LF_HIRSCH_H
#define LF_HIRSCH_H
#include "lf_edlib_common.h"

/* One node of edlib's recursion (obtainAlignmentHirschberg, lib/edlib/edlib.cpp:1161-1330): an NW problem on a query range x
 * target range of its ROOT problem, in the root's orientation (the root descriptor's direction / complement flags apply).
 * kind 1 = the distance pass of an SHW root (edit distance + end column first, lib/edlib/edlib.cpp:141-168), which turns
 * into an ordinary node on the truncated target. */
struct lf_hnode {
    int64_t  qstart, tstart;   /* element 0 of the node's query / target (descriptor coordinates) */
    uint64_t ops_off;          /* the node's part of the root's ops region: capacity n + m */
    uint32_t n, m;
    int32_t  best;             /* edit distance of the node, -1 = not known yet (NW roots: the first split finds it) */
    uint32_t root;             /* index into the root table */
    uint8_t  flags, kind, is_root, pad;      /* pad: bit 0 = the root's target holds bytes other than ACGT (stage API), bit 1 = LF_HN_NOBAND */
    uint32_t k0;               /* != 0: a TRIAL bound -- the node is swept inside the band of distance k0 and goes back to the queue unbanded if its distance is larger */
};
#define LF_HN_NOBAND 2u
#define LF_HQ 6
struct lf_hroot { uint64_t ops_off; uint32_t desc, n, m, seg_off, seg_cap, count; };      /* count: pieces registered so far (atomic) */
/* one finished piece of a root's path: its region inside the root's ops region and its length (bit 31 set: H-leaf j, whose
 * length the traceback kernels leave in out_len[n_desc + j]) */
struct lf_hseg { uint64_t off; uint32_t cap, len; };
struct lf_hctl {
    uint32_t n_roots, seg_used, n_hleaf, fail;
    uint32_t q_n[2][LF_HQ];              /* nodes queued for the next / current level: classes 0 .. 2 unbanded sweeps by query rows (1 / 4 / 8 wavefronts per half),
                                          * 3 .. 5 banded sweeps by band width (1 / 2 / 4 wavefronts per half) */
    uint32_t n_trial, n_trial_failed, pad_[2];
    unsigned long long aux_used, hcar_used;
};
struct lf_hargs {
    lf_seqs S; int64_t pac_syms;          /* symbols in S.pac (2-bit targets) */
    const lf_hnode *q_in; lf_hnode *q_out[LF_HQ];
    const uint64_t *qlo, *qhi, *qvalid; int64_t q_words;      /* bit planes of S.q (lf_pack_planes_kernel): the banded sweeps take a block's match masks from them */
    uint32_t n_in, q_cap, out_par;
    uint32_t no_band;                     /* A / B and test hook (LF_HIRSCH_BAND=0): every node takes the unbanded sweep of its size */
    lf_hctl *ctl; lf_hroot *roots; lf_hseg *segs;
    lf_aln_desc_t *hdesc; uint64_t *hopsoff; uint32_t hleaf_cap;
    uint64_t *aux; uint64_t aux_cap; uint8_t *hcar; uint64_t hcar_cap;
    uint8_t *ops; int32_t *out_ed, *out_end; uint32_t *out_len; uint32_t n_desc;
};
/* The carries that leave a super-band of a query above 32 768 rows wait in HBM for the super-band below.  2-bit targets, one block per lane: one 32-bit word per 16
 * steps of the super-band's last lane (the sweep's steps incl. the lag of up to eight wavefronts, read up to four words ahead); other instantiations: one byte
 * per column.  A node reserves room for four such buffers (two per half) in either form. */
__host__ __device__ __forceinline__ uint64_t lf_hband_bytes(uint32_t m) { return (((uint64_t)m + 63 + 7 * 96 + 15) / 16 + 8) * 4; }
__host__ __device__ __forceinline__ uint64_t lf_hband_reserve(uint32_t m) { const uint64_t a = 2ull * m + 64, b = 4 * lf_hband_bytes(m); return ((a > b ? a : b) + 3) & ~3ull; }
__host__ __device__ __forceinline__ int lf_hkb_class(uint32_t n) { return n <= 4096 ? 0 : n <= 16384 ? 1 : 2; }

/* ---- the band of a node (lib/edlib/edlib.cpp:134-153 finds k by doubling, :484-566 / :657-858 keep the band while sweeping; here the band is a
 * STATIC set of diagonals: a node below the root knows its distance exactly -- its side of `left + right == best`, :1263-1289 -- and a root tries a bound).
 * Cell (i, j), 0-based, lies on diagonal d = j - i.  A path of cost <= k from (0, 0) to (n, m) stays on diagonals [min(0, m - n) - e, max(0, m - n) + e],
 * e = floor((k - |m - n|) / 2); the reversed half of a Hirschberg node sees the same interval.  SHW (the target's end is free): [-k, k].  The sweep treats what
 * lies outside as edlib does: +1 enters a block whose upper neighbour is outside the band, a block that enters the band starts from a column of +1s -- every
 * computed value is >= the true one and every cell on an optimal path is exact, so `F + R == best` holds for exactly the rows it holds for in the full matrix. ---- */
struct lf_hband { int dlo, dhi; };
static inline __host__ __device__ lf_hband lf_hband_nw(uint32_t n, uint32_t m, int k)
{
    const int md = (int)m - (int)n, amd = md < 0 ? -md : md;
    int e = (k - amd) / 2; if (e < 0) e = 0;
    lf_hband B; B.dlo = (md < 0 ? md : 0) - e; B.dhi = (md > 0 ? md : 0) + e;
    if (B.dhi < 1) B.dhi = 1;                /* a block must still be inside the band at the column at which the block below it starts */
    return B;
}
static inline __host__ __device__ lf_hband lf_hband_shw(int k) { lf_hband B; B.dlo = -k; B.dhi = k < 1 ? 1 : k; return B; }
/* block b of a banded sweep lives on lane b mod 64 W of the half's W wavefronts and runs skew(b) steps behind block 0; a lane must be done with block b before
 * block b + 64 W enters the band (blocks change hands at the boundaries of 16-step groups): the widest band W wavefronts hold */
#define LF_HB_LAG(W) ((W) == 1 ? 64 : 96)
static inline __host__ __device__ bool lf_hband_fits(lf_hband B, int W) { return B.dhi - B.dlo <= 4096 * W + LF_HB_LAG(W) * W - 94; }
/* trial bounds of the roots: a quarter of the longer string on top of the length difference (15 % reads end up at 0.16 - 0.2 n) */
static inline __host__ __device__ uint32_t lf_htrial_nw(uint32_t n, uint32_t m) { const uint32_t d = n > m ? n - m : m - n, x = n > m ? n : m; return d + x / 4 + 1; }
static inline __host__ __device__ uint32_t lf_htrial_shw(uint32_t n) { return n / 4 + 1; }
/* the queue of a node: 3 + log2(W) when its band fits W <= 4 wavefronts (only queries above 4096 rows: a smaller one is ONE wavefront per half either way
 * and its sweep lasts m + blocks steps with or without a band), else by rows.  *k0: the trial bound when the distance is not known */
static inline __host__ __device__ int lf_hqueue_of(uint32_t n, uint32_t m, int best, int kind, unsigned pad, uint32_t *k0)
{
    *k0 = 0;
    if (n > 4096 && pad == 0) {
        uint32_t t = 0; lf_hband B;
        if (kind == 1) { t = lf_htrial_shw(n); B = lf_hband_shw((int)t); }
        else if (best >= 0) B = lf_hband_nw(n, m, best);
        else { t = lf_htrial_nw(n, m); B = lf_hband_nw(n, m, (int)t); }
        for (int c = 0, W = 1; c < 3; c++, W *= 2) if (lf_hband_fits(B, W)) { *k0 = t; return 3 + c; }
    }
    return lf_hkb_class(n);
}

void lf_hirsch_launch_roots(hipStream_t s, bool pac_targets, const lf_aln_desc_t *d_desc, const uint64_t *d_opsoff, int n, lf_hargs A);
void lf_hirsch_launch_level(hipStream_t s, bool pac_targets, int kbc, lf_hargs A);      /* kbc 0 .. 2: unbanded, 3 .. 5: banded */
void lf_hirsch_launch_stitch(hipStream_t s, lf_hargs A, uint32_t n_roots);
#endif
