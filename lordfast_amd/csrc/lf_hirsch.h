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
    uint8_t  flags, kind, is_root, pad;      /* pad: the root's target holds bytes other than ACGT (stage API) */
    uint32_t k0;               /* banded queues, distance not known: the node is swept inside the band of distance k0 (n + m: the whole matrix) and goes back to the
                                * queue with the bound that sweep found if its distance turns out larger (lf_hrequeue_bound) */
};
#define LF_HQ 16
struct lf_hroot { uint64_t ops_off; uint32_t desc, n, m, seg_off, seg_cap, count; };      /* count: pieces registered so far (atomic) */
/* one finished piece of a root's path: its region inside the root's ops region and its length (bit 31 set: H-leaf j, whose
 * length the traceback kernels leave in out_len[n_desc + j]) */
struct lf_hseg { uint64_t off; uint32_t cap, len; };
struct lf_hctl {
    uint32_t n_roots, seg_used, n_hleaf, fail;
    uint32_t q_n[2][LF_HQ];              /* nodes queued for the next / current level, by queue (lf_hqueue_of) */
    uint32_t n_trial, n_trial_failed, max_root_n, pad_;
    uint32_t ratio_hist[2][16];          /* the roots above 4096 rows by 16 distance / rows, NW and SHW: what the next calls' trial bounds are chosen from */
    unsigned long long aux_used, hcar_used;
};
struct lf_hargs {
    lf_seqs S; int64_t pac_syms;          /* symbols in S.pac (2-bit targets) */
    const lf_hnode *q_in; lf_hnode *q_out[LF_HQ];
    const uint64_t *qlo, *qhi, *qvalid; int64_t q_words;      /* bit planes of S.q (lf_pack_planes_kernel): the banded sweeps take a block's match masks from them */
    uint32_t n_in, q_cap, out_par;
    uint32_t no_band;                     /* A / B and test hook: 1 (LF_HIRSCH_BAND=0): every node takes the unbanded sweep of its size; 2 (LF_HIRSCH_BAND=64): no sixteen-lane queues */
    uint32_t trial16[2];                  /* trial bound of the NW / SHW roots in sixteenths of their rows, 0 = none */
    uint32_t trial_min;                   /* ... of the roots above this many rows (LF_HTRIAL_MIN_ROWS) */
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
 * block b + 64 W enters the band (blocks change hands at the boundaries of 16-step groups): the widest band W wavefronts hold.  What counts is the band INSIDE the
 * matrix of the half (n rows, mm columns): block b is in it for columns [max(0, 64 b + dlo), min(mm - 1, 64 b + 63 + dhi)]. */
#define LF_HB_LAG(W) ((W) == 1 ? 64 : 96)
static inline __host__ __device__ bool lf_hband_fits(lf_hband B, uint32_t n, int mm, int W)
{
    const int nbk = (int)((n + 63) >> 6);
    const int dhi = B.dhi < mm - 64 ? B.dhi : mm - 64, dlo = B.dlo > -64 * (nbk - 1) ? B.dlo : -64 * (nbk - 1);
    return dhi - dlo <= 4096 * W + LF_HB_LAG(W) * W - 94;
}
/* trial bounds of the roots, in sixteenths of the rows (trial16; 0 = none), chosen from the distances of the roots above 4096 rows -- the ones whose chain a failed trial
 * makes longer -- and applied to every root above LF_HTRIAL_MIN_ROWS: the many roots just above edlib's traceback switch (2 000 - 4 000 rows: C5's gaps between sparse
 * anchors) would otherwise be swept with the whole matrix as their band, forty blocks per column on two wavefronts per half where their distance needs seven on sixteen lanes.  A root does not know its distance: it is swept inside the band of a
 * bound k0 -- min (F + R) <= k0, resp. the SHW minimum <= k0, proves the result exact -- and k0 = n + m is the whole matrix.  A trial that fails still
 * says something: every value a banded sweep computes is the cost of a real path, so the minimum it found is an UPPER bound of the distance and the band of THAT bound
 * holds an optimal path -- the root goes back to the queue with it (for unrelated strings ~0.55 of the rows: a third of the whole matrix).  The levels are a chain of
 * dependent steps whose number a band does not change, so a trial that fails costs a whole level: lf_align.hip picks trial16 per mode from the distances of the roots the
 * process has seen so far (a batch is homogeneous: PacBio tails in the right place end at 0.16 - 0.2 of their rows, ONT ones at 0.1, tails in the wrong copy of a
 * duplication at 0.47).  Results do not depend on it. */
static inline __host__ __device__ uint32_t lf_htrial_nw(uint32_t n, uint32_t m, uint32_t trial16) { const uint32_t d = n > m ? n - m : m - n, x = n > m ? n : m; return d + (uint32_t)((uint64_t)x * trial16 / 16) + 1; }
static inline __host__ __device__ uint32_t lf_htrial_shw(uint32_t n, uint32_t trial16) { return (uint32_t)((uint64_t)n * trial16 / 16) + 1; }
/* the queue of a node.  Queues 3 .. 6: NW nodes on the banded sweep with 1 / 2 / 4 / 8 wavefronts per half; 7 .. 11: SHW roots (one half) on 1 .. 16 wavefronts;
 * 0 .. 2: what does not fit sixteen wavefronts, and targets with bytes other than ACGT (stage API), on the unbanded sweeps by rows (super-bands through HBM
 * above 32 768 rows).  *k0: the bound a node of unknown distance is swept with */
#define LF_HTRIAL_MIN_ROWS 512
#define LF_HQ_NW0 3
#define LF_HQ_SHW0 7
#define LF_HQ_NW16 12        /* NW nodes whose band fits SIXTEEN LANES: two nodes (four halves) per wavefront, lf_hband_group_kernel */
#define LF_HQ_SHW16 13       /* SHW roots ... : four per wavefront */
#define LF_HQ_NW32 14        /* ... THIRTY-TWO LANES: the two halves of a node in one wavefront */
#define LF_HQ_SHW32 15       /* ... two SHW roots per wavefront */
/* L lanes hold a band of 64 L + L - 94 diagonals (the same rule as lf_hband_fits with L lanes per "wavefront") */
static inline __host__ __device__ bool lf_hband_fits_lanes(lf_hband B, uint32_t n, int mm, int L)
{
    const int nbk = (int)((n + 63) >> 6);
    const int dhi = B.dhi < mm - 64 ? B.dhi : mm - 64, dlo = B.dlo > -64 * (nbk - 1) ? B.dlo : -64 * (nbk - 1);
    return dhi - dlo <= 64 * L + L - 94;
}
/* t: the bound a node of unknown distance (best < 0) is swept with */
static inline __host__ __device__ int lf_hqueue_of_bound(uint32_t n, uint32_t m, int best, int kind, unsigned pad, uint32_t t, uint32_t *k0)
{
    *k0 = 0;
    if ((pad & 1u) == 0 && (uint64_t)n + m < (1u << 30)) {      /* pad bit 0: no band; bit 1: no sixteen- / thirty-two-lane queues */
        if (kind == 1) {
            const lf_hband B = lf_hband_shw((int)t);
            const uint64_t me = (uint64_t)n + t; const int mm = (int)(me < m ? me : m);
            if (!(pad & 2u) && lf_hband_fits_lanes(B, n, mm, 16)) { *k0 = t; return LF_HQ_SHW16; }
            if (!(pad & 2u) && lf_hband_fits_lanes(B, n, mm, 32)) { *k0 = t; return LF_HQ_SHW32; }
            for (int c = 0, W = 1; c < 5; c++, W *= 2) if (lf_hband_fits(B, n, mm, W)) { *k0 = t; return LF_HQ_SHW0 + c; }
        } else {
            if (best >= 0) t = 0;
            const lf_hband B = lf_hband_nw(n, m, best >= 0 ? best : (int)t);
            if (!(pad & 2u) && lf_hband_fits_lanes(B, n, (int)(m - m / 2), 16)) { *k0 = t; return LF_HQ_NW16; }
            if (!(pad & 2u) && lf_hband_fits_lanes(B, n, (int)(m - m / 2), 32)) { *k0 = t; return LF_HQ_NW32; }
            for (int c = 0, W = 1; c < 4; c++, W *= 2) if (lf_hband_fits(B, n, (int)(m - m / 2), W)) { *k0 = t; return LF_HQ_NW0 + c; }
        }
    }
    return lf_hkb_class(n);
}
static inline __host__ __device__ int lf_hqueue_of(uint32_t n, uint32_t m, int best, int kind, unsigned pad, uint32_t trial16, uint32_t trial_min, uint32_t *k0)
{
    const bool trial = n > trial_min && trial16 != 0;
    const uint32_t t = !trial ? n + m : kind == 1 ? lf_htrial_shw(n, trial16) : lf_htrial_nw(n, m, trial16);
    return lf_hqueue_of_bound(n, m, best, kind, pad, t, k0);
}

void lf_hirsch_launch_roots(hipStream_t s, bool pac_targets, const lf_aln_desc_t *d_desc, const uint64_t *d_opsoff, int n, lf_hargs A);
void lf_hirsch_launch_level(hipStream_t s, bool pac_targets, int kbc, lf_hargs A);      /* kbc: the queue (lf_hqueue_of) */
void lf_hirsch_launch_stitch(hipStream_t s, lf_hargs A, uint32_t n_roots);
#endif
