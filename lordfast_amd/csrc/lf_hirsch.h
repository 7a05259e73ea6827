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
    uint8_t  flags, kind, is_root, pad;
    uint32_t pad2;
};
struct lf_hroot { uint64_t ops_off; uint32_t desc, n, m, seg_off, seg_cap, count; };      /* count: pieces registered so far (atomic) */
/* one finished piece of a root's path: its region inside the root's ops region and its length (bit 31 set: H-leaf j, whose
 * length the traceback kernels leave in out_len[n_desc + j]) */
struct lf_hseg { uint64_t off; uint32_t cap, len; };
struct lf_hctl {
    uint32_t n_roots, seg_used, n_hleaf, fail;
    uint32_t q_n[2][4];                  /* nodes queued for the next / current level, by blocks-per-lane class (KB 1 / 4 / 8) */
    unsigned long long aux_used, hcar_used;
};
struct lf_hargs {
    lf_seqs S; int64_t pac_syms;          /* symbols in S.pac (2-bit targets) */
    const lf_hnode *q_in; lf_hnode *q_out[3];
    uint32_t n_in, q_cap, out_par;
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

void lf_hirsch_launch_roots(hipStream_t s, bool pac_targets, const lf_aln_desc_t *d_desc, const uint64_t *d_opsoff, int n, lf_hargs A);
void lf_hirsch_launch_level(hipStream_t s, bool pac_targets, int kbc, lf_hargs A);
void lf_hirsch_launch_stitch(hipStream_t s, lf_hargs A, uint32_t n_roots);
#endif
