/*
 * lf_stages.c -- stage-level batch entry points of the C ABI: chaining, edlib-equivalent alignment and ksw.
 * All DP runs in HIP kernels (lf_chain.hip, lf_align.hip, lf_hirsch.hip).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lf_internal.h"
#include "lf_stdsort.h"

/* compare_seed (src/Chain.cpp:227-230): by qPos only -> ties keep introsort's permutation */
#define SEED_QLESS(a, b) ((a)->qPos < (b)->qPos)
LF_DEFINE_STDSORT(seedq, Seed_t, SEED_QLESS)

void lf_sort_seeds_by_qpos(Seed_t *s, long n) { seedq_sort(s, n); }

int lf_chain_n2_batch(const lf_params_t *p, int n_windows, Seed_t *seeds, const uint64_t *off,
                      uint32_t *chain_idx, uint32_t *chain_len, float *score, int device)
{
    if (!p || n_windows < 0) { lf_set_error("lf_chain_n2_batch: bad argument"); return LF_ERR_ARG; }
    if (p->chain_alg != 0) { lf_set_error("lf_chain_n2_batch: only dp-n2 is implemented on this path"); return LF_ERR_ARG; }
    for (int w = 0; w < n_windows; w++) seedq_sort(seeds + off[w], (long)(off[w + 1] - off[w]));   /* std::sort, src/Chain.cpp:244 */
    return lfg_chain_n2(device, p, n_windows, seeds, off, chain_idx, chain_len, score, NULL);
}

int lf_chain_clasp_batch(int n_windows, const Seed_t *seeds, const uint64_t *off,
                         Seed_t *chain_out, uint32_t *chain_len, float *score, int device)
{
    if (n_windows < 0 || (n_windows && (!seeds || !off || !chain_out || !chain_len || !score))) { lf_set_error("lf_chain_clasp_batch: bad argument"); return LF_ERR_ARG; }
    return lfg_chain_clasp(device, n_windows, seeds, off, chain_out, chain_len, score, NULL);
}

int lf_ksw_extend2_batch(int n, const uint8_t *q, const uint64_t *qoff, const uint8_t *t, const uint64_t *toff,
                         const int32_t *prm, int32_t *score, int32_t *qle, int32_t *tle, int device)
{
    if (n < 0) { lf_set_error("lf_ksw_extend2_batch: bad argument"); return LF_ERR_ARG; }
    for (int i = 0; i < n; i++) if (prm[7 * i + 6] <= 0 || prm[7 * i + 1] <= 0 || prm[7 * i + 3] <= 0) {
        lf_set_error("lf_ksw_extend2_batch: h0, e_del, e_ins must be positive (lib/bwa/ksw.c:385)"); return LF_ERR_ARG;
    }
    return lfg_ksw(device, n, q, qoff, t, toff, prm, score, qle, tle, NULL);
}

/* ------------------------------------------------------------------------------------------------
 * edlib: the byte strings are uploaded and described like the pipeline's requests; binning, size-class kernels and -- for
 * problems above edlib's 1 MiB traceback switch -- the breadth-first Hirschberg levels all run on the device
 * (lf_align.hip, lf_hirsch.hip).  The host does no per-problem work.
 * ---------------------------------------------------------------------------------------------- */
int lf_edlib_batch(int n, const char *q, const uint64_t *qoff, const char *t, const uint64_t *toff,
                   const uint8_t *mode, int32_t *edit_distance, int32_t *end_location,
                   uint8_t *ops, uint32_t *ops_len, int device, float *kernel_ms)
{
    if (n < 0 || !qoff || !toff) { lf_set_error("lf_edlib_batch: bad argument"); return LF_ERR_ARG; }
    if (mode) for (int i = 0; i < n; i++) if (mode[i] > 1) { lf_set_error("lf_edlib_batch: only NW (0) and SHW (1) are implemented"); return LF_ERR_ARG; }
    int rc = lfg_edlib(device, n, q, qoff, t, toff, mode, edit_distance, end_location, ops, ops_len, kernel_ms);
    if (rc != LF_OK) return rc;
    /* the kernels leave each ops run END-aligned in its region; the public contract is start-aligned */
    for (int i = 0; i < n; i++) {
        uint8_t *o = ops + qoff[i] + toff[i];
        const uint64_t cap = (qoff[i + 1] - qoff[i]) + (toff[i + 1] - toff[i]);
        if (ops_len[i] && ops_len[i] < cap) memmove(o, o + cap - ops_len[i], ops_len[i]);
    }
    return LF_OK;
}
