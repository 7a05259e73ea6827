/* lf_rsweep.h -- the forward pass of all problems with <= 64 query blocks (lf_rsweep.hip), driven by lf_align.hip. */
#ifndef LF_RSWEEP_H
#define LF_RSWEEP_H
#include "lf_edlib_common.h"

/* one wavefront of the forward kernel: `count` problems of G blocks each, probs[first ..], checkpoints from hist_base */
struct lf_rwave { uint32_t first; uint16_t count, G; uint64_t hist_base; };
struct lf_rsw_args {
    const lf_aln_prob *probs; const lf_rwave *waves; int wave0, n_waves, rev;      /* rev: workgroup b takes wave n_waves - 1 - b */
    const uint64_t *qlo, *qhi, *qvalid; int64_t q_words;      /* bit planes of the query buffer (lf_pack_planes_kernel) */
    const uint8_t *pac; int64_t pac_syms;                      /* 2-bit targets, four per byte, first symbol in the top bits */
    lf_hist_t *ckpt; int32_t *out_ed, *out_end;
    uint8_t *ops; uint32_t *out_len;                           /* the fused small-problem kernel writes the paths itself */
};
/* words of ONE plane for a buffer of n bytes (three planes follow each other) */
static inline uint64_t lf_plane_words(uint64_t n_bytes) { return (n_bytes + 63) / 64 + 2; }
/* lower_flag (optional): set to non-zero when the bytes hold a lower-case a / c / g / t (seeding accepts those, the planes do not) */
void lf_rsweep_pack_planes(hipStream_t s, const unsigned char *d_src, uint64_t n_bytes, uint64_t *d_planes, uint64_t n_words, unsigned long long *lower_flag = nullptr);
void lf_rsweep_pack_pac(hipStream_t s, const unsigned char *d_src, uint64_t n_bytes, uint8_t *d_pac);
void lf_rsweep_launch(hipStream_t s, bool track, lf_rsw_args A);
#endif
