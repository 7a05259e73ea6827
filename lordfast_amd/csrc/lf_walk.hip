/*
 * lf_walk.hip -- the common path of alignChain_edlib (src/LordFAST.cpp:1765-2258) on the device.
 *
 * The reference extends a chain piece by piece: the read prefix before the first anchor (SHW on reverse complements,
 * :1820-1899), every gap between adjacent anchors (NW, :1901-2137), the suffix after the last anchor (SHW, :2149-2230),
 * and only leaves that path when a piece is long and dissimilar enough to call ksw_extend (clip test :1840-1848,
 * :2172-2180; split test :1952-1983).  Round 1 walked that control flow on the host twice per chain -- once to learn
 * which alignments are needed, once with their results -- and shipped 32-byte descriptors down and 32-byte CIGAR
 * recipe items up for every piece (~1 GB per 100 k reads over PCIe, 1 core-second per step).
 *
 * Here a chain ("job" = one kept candidate window of a read) never leaves HBM on the common path:
 *
 *   lf_walk_plan_kernel   one wavefront per job, one lane per piece: which pieces need an alignment, their descriptors
 *                         (start / direction / complement into the resident read batch and 2-bit reference) -- a counting
 *                         pass, two scans, a writing pass
 *   (lf_align.hip)        the alignments, binned and laid out on the device as before
 *   lf_walk_emit_kernel   one wavefront per job: the clip / split triggers on the distances; if none fires, the record's
 *                         position, clip lengths and NM, and the ORDER of its pieces as recipe items for lf_render_kernel
 *
 * A job with a trigger (a few per cent), a query longer than the sweep kernels take, or anything else off the common
 * path is flagged and replayed by the host walk of lf_pipeline.c exactly as before.  lf_debug_crosscheck(4) sends every job there
 * (cross-check in the tests).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "lf_internal.h"
#include "lf_gpu_common.h"
#include "lf_scan.h"
#include <algorithm>

/* src/LordFAST.cpp:88-92 */
#define W_CLIP_LEN  500
#define W_CLIP_SIM  0.75
#define W_SPLIT_LEN 80
#define W_SPLIT_SIM 0.40

struct lf_walk_dev {
    const uint64_t *read_off;      /* start of each read in the resident batch (n_reads + 1) */
    const uint2 *chain_seeds;      /* (tPos, qPos | len << 20) */
    const uint64_t *chain_off;     /* per request */
    const int64_t *ctg_off, *ctg_len; int n_ctg; int64_t l_pac;
};

/* one piece of a chain.  slot 0: before the first anchor; slot k (1 <= k < chainLen): between anchors k-1 and k;
 * slot chainLen: after the last anchor */
struct lf_piece {
    int kind;                      /* 0 nothing, 1 alignment, 2 insertion run (no reference left / empty reference gap),
                                      3 deletion run, -1 off the common path */
    uint32_t qs, qn, ts, tn;       /* query [qs, qs + qn) of the walk's query string, reference [ts, ts + tn) */
    int mode, rc;                  /* SHW / NW; head: both strings reverse-complemented */
};

__device__ __forceinline__ void lf_chr_bounds(const lf_walk_dev &D, uint32_t beg, uint32_t end, uint32_t *cb, uint32_t *ce)
{   /* bwt_get_chr_boundaries (src/BWT.cpp:653-666): the contig of the MIDPOINT; beyond l_pac: the last one (lf_pipeline.c) */
    const int64_t mid = (int64_t)(((uint64_t)beg + (uint64_t)end) >> 1);
    int rid;
    if (mid >= D.l_pac) rid = D.n_ctg - 1;
    else { int lo = 0, hi = D.n_ctg - 1; while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (D.ctg_off[m] <= mid) lo = m; else hi = m - 1; } rid = lo; }
    *cb = (uint32_t)D.ctg_off[rid]; *ce = (uint32_t)(D.ctg_off[rid] + D.ctg_len[rid] - 1);
}

__device__ __forceinline__ lf_piece lf_piece_of(const uint2 *s, uint32_t chainLen, uint32_t k, uint32_t L, uint32_t chrBeg, uint32_t chrEnd)
{
    lf_piece P; P.kind = 0; P.qs = P.qn = P.ts = P.tn = 0; P.mode = 0; P.rc = 0;
    if (k == 0) {                                                            /* :1820-1899 */
        const uint32_t q0 = s[0].y & 0xFFFFFu, t0 = s[0].x;
        const int32_t readAlnLen = (int32_t)q0, refAlnLen = readAlnLen + 20;
        if (readAlnLen > 0) {
            if ((int64_t)t0 - refAlnLen >= (int64_t)chrBeg) { P.kind = 1; P.qs = 0; P.qn = (uint32_t)readAlnLen; P.ts = t0 - (uint32_t)refAlnLen; P.tn = (uint32_t)refAlnLen; P.mode = 1; P.rc = 1; }
            else { P.kind = 2; P.qn = (uint32_t)readAlnLen; }
        }
        return P;
    }
    const uint32_t qa = s[k - 1].y & 0xFFFFFu, la = s[k - 1].y >> 20, ta = s[k - 1].x;
    if (k < chainLen) {                                                      /* :1901-2137 */
        const uint32_t readAlnStart = qa + la, refAlnStart = ta + la;
        const int32_t readAlnLen = (int32_t)((s[k].y & 0xFFFFFu) - readAlnStart), refAlnLen = (int32_t)(s[k].x - refAlnStart);
        if (readAlnLen < 0 || refAlnLen < 0) { P.kind = -1; return P; }      /* overlapping fragments: the host walk keeps the reference's arithmetic */
        P.qs = readAlnStart; P.ts = refAlnStart; P.qn = (uint32_t)readAlnLen; P.tn = (uint32_t)refAlnLen;
        if (readAlnLen > 0 && refAlnLen > 0) P.kind = 1;
        else if (readAlnLen > 0) P.kind = 2;
        else if (refAlnLen > 0) P.kind = 3;
        return P;
    }
    {                                                                        /* :2149-2230 */
        const uint32_t readAlnStart = qa + la;
        const int32_t readAlnLen = (int32_t)L - (int32_t)readAlnStart, refAlnLen = readAlnLen + 20;
        if (readAlnLen > 0) {
            if (ta + la + (uint32_t)refAlnLen - 1 <= chrEnd) { P.kind = 1; P.qs = readAlnStart; P.qn = (uint32_t)readAlnLen; P.ts = ta + la; P.tn = (uint32_t)refAlnLen; P.mode = 1; }
            else { P.kind = 2; P.qs = readAlnStart; P.qn = (uint32_t)readAlnLen; }
        }
        return P;
    }
}

#define LF_WALK_TOT 256
/* ---- plan: count, then write descriptors ---- */
template <bool WRITE>
__global__ void __launch_bounds__(64)
lf_walk_plan_kernel(int n_jobs, const lf_wjob_t *__restrict__ jobs, lf_walk_dev D, int lazy,
                    uint32_t *__restrict__ job_ndesc, uint64_t *__restrict__ job_opsbytes, uint8_t *__restrict__ job_rare,
                    const uint64_t *__restrict__ desc_base, const uint64_t *__restrict__ ops_base, const uint64_t *__restrict__ slot_base,
                    lf_aln_desc_t *__restrict__ desc, uint64_t *__restrict__ ops_off, int32_t *__restrict__ slot_desc,
                    unsigned long long *__restrict__ totals /* ext_bytes, block_steps, then lf_hcount_t: roots, cap, sum_n, sum_m */)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= n_jobs) return;
    if (WRITE && job_rare[j]) return;                                         /* off the common path: the host walk plans it */
    const lf_wjob_t J = jobs[j];
    const uint2 *s = D.chain_seeds + D.chain_off[J.req];
    const uint32_t chainLen = J.chain_len;
    const uint64_t roff = D.read_off[J.read];
    const uint32_t L = (uint32_t)(D.read_off[J.read + 1] - roff);
    uint32_t chrBeg, chrEnd;
    lf_chr_bounds(D, s[0].x, s[chainLen - 1].x, &chrBeg, &chrEnd);           /* :1799 */
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t nd = 0; uint64_t ob = 0; bool rare = false;
    unsigned long long ext = 0, blk = 0, hr = 0, hcap = 0, hn = 0, hm = 0;      /* h*: pieces above edlib's traceback switch (roots of the Hirschberg levels) */
    const uint64_t dbase = WRITE ? desc_base[j] : 0, obase = WRITE ? ops_base[j] : 0, sbase = WRITE ? slot_base[j] : 0;
    for (uint32_t base = 0; base <= chainLen; base += 64) {
        const uint32_t k = base + lane;
        lf_piece P; P.kind = 0; P.qn = P.tn = 0;
        if (k <= chainLen) P = lf_piece_of(s, chainLen, k, L, chrBeg, chrEnd);
        const bool aln = P.kind == 1;
        const uint64_t am = lf_ballot(aln);
        rare |= lf_any(P.kind < 0) != 0;
        /* inclusive-exclusive prefix of the ops bytes of this tile's alignments (wave scan by shuffles: once per 64 pieces) */
        uint32_t bytes = aln ? P.qn + P.tn : 0, incl = bytes;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const uint32_t tile_bytes = __shfl(incl, 63);
        if (WRITE) {
            if (k <= chainLen) slot_desc[sbase + k] = aln ? (int32_t)(dbase + nd + (uint32_t)__popcll(am & below)) : -1;
            if (aln) {
                const uint64_t di = dbase + nd + (uint32_t)__popcll(am & below);
                lf_aln_desc_t d;
                /* the walk's query string is the read (forward chains) or its reverse complement (reverse chains); the
                 * head piece is itself aligned on reverse complements: compose into (start, direction, complement) */
                const int rev_q = (J.is_rev ? 1 : 0) ^ P.rc;
                int64_t qstart;
                if (!J.is_rev) qstart = P.rc ? (int64_t)roff + P.qs + P.qn - 1 : (int64_t)roff + P.qs;
                else qstart = P.rc ? (int64_t)roff + L - P.qs - P.qn : (int64_t)roff + L - 1 - P.qs;
                d.qstart = qstart;
                d.tstart = P.rc ? (int64_t)P.ts + P.tn - 1 : (int64_t)P.ts;
                d.n = P.qn; d.m = P.tn; d.mode = (uint8_t)P.mode;
                d.flags = (uint8_t)((rev_q ? (LF_F_QREV | LF_F_QCOMP) : 0) | (P.rc ? (LF_F_TREV | LF_F_TCOMP) : 0) | (lazy ? LF_F_LAZYX : 0));
                for (int z = 0; z < 6; z++) d.pad[z] = 0;
                desc[di] = d;
                ops_off[di] = obase + ob + (incl - bytes);
            }
        }
        if (aln) {
            ext += (uint64_t)P.qn + (P.tn + 3) / 4 + P.qn + P.tn; blk += (uint64_t)((P.qn + 63) / 64) * P.tn;
            if (P.qn && P.tn && !lf_is_leaf(P.qn, P.tn)) { hr++; hcap += lf_hroot_cap(P.qn, P.tn); hn += P.qn; hm += P.tn; }
        }
        nd += (uint32_t)__popcll(am); ob += tile_bytes;
    }
    if (!WRITE) {
        if (lane == 0) { job_ndesc[j] = rare ? 0 : nd; job_opsbytes[j] = rare ? 0 : ob; job_rare[j] = rare ? 1 : 0; }
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ext += __shfl_xor(ext, o); blk += __shfl_xor(blk, o); hr += __shfl_xor(hr, o); hcap += __shfl_xor(hcap, o); hn += __shfl_xor(hn, o); hm += __shfl_xor(hm, o); }
        /* LF_WALK_TOT sets of counters, a job adds to set j mod LF_WALK_TOT (the host sums the sets): 100 k wavefronts adding to ONE
         * set of addresses were served one after the other by the L2 -- 2.5 ms per 100 k reads for a kernel whose counting twin
         * takes 0.05 */
        if (lane == 0) { unsigned long long *tt = totals + (size_t)(j & (LF_WALK_TOT - 1)) * 8; atomicAdd(&tt[0], ext); atomicAdd(&tt[1], blk); if (hr) { atomicAdd(&tt[2], hr); atomicAdd(&tt[3], hcap); atomicAdd(&tt[4], hn); atomicAdd(&tt[5], hm); } }
    }
}

/* ---- emit: triggers, record fields, recipe items ---- */
__global__ void __launch_bounds__(64)
lf_walk_emit_kernel(int n_jobs, const lf_wjob_t *__restrict__ jobs, lf_walk_dev D, int lazy, const uint8_t *__restrict__ job_rare_in,
                    const uint64_t *__restrict__ slot_base, const int32_t *__restrict__ slot_desc, const uint64_t *__restrict__ item_base,
                    const int32_t *__restrict__ ed, const int32_t *__restrict__ endloc, const uint32_t *__restrict__ ops_len,
                    const uint64_t *__restrict__ ops_off,
                    lf_rrecord_t *__restrict__ recs, lf_ritem_t *__restrict__ items, lf_wrec_t *__restrict__ wrec)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= n_jobs) return;
    const lf_wjob_t J = jobs[j];
    const uint64_t ib = item_base[j];
    if (job_rare_in[j]) {
        if (lane == 0) { lf_wrec_t w; memset(&w, 0, sizeof w); w.rare = 1; wrec[j] = w; recs[j].item0 = (uint32_t)ib; recs[j].nitems = 0; }
        return;
    }
    const uint2 *s = D.chain_seeds + D.chain_off[J.req];
    const uint32_t chainLen = J.chain_len;
    const uint64_t roff = D.read_off[J.read];
    const uint32_t L = (uint32_t)(D.read_off[J.read + 1] - roff);
    uint32_t chrBeg, chrEnd;
    lf_chr_bounds(D, s[0].x, s[chainLen - 1].x, &chrBeg, &chrEnd);
    const uint64_t sb = slot_base[j];
    bool rare = false;
    int nm = 0;
    uint32_t pos = s[0].x, qStart = s[0].y & 0xFFFFFu;                                              /* :1092 */
    uint32_t posEnd = s[chainLen - 1].x + (s[chainLen - 1].y >> 20) - 1, qEnd = (s[chainLen - 1].y & 0xFFFFFu) + (s[chainLen - 1].y >> 20) - 1;   /* :2150 */
    for (uint32_t base = 0; base <= chainLen; base += 64) {
        const uint32_t k = base + lane;
        if (k <= chainLen) {
            const lf_piece P = lf_piece_of(s, chainLen, k, L, chrBeg, chrEnd);
            /* item layout of a job: [0] head piece, then per anchor a [2k-1] its match run and [2k] the piece after it */
            lf_ritem_t it; memset(&it, 0, sizeof it);
            if (P.kind == 1) {
                const int32_t di = slot_desc[sb + k];
                const int e = ed[di], en = endloc[di]; const uint32_t nops = ops_len[di];
                if (e < 0) rare = true;                                     /* the sweep kernel found no Hirschberg split (cannot happen): let the host replay report it */
                if (k == 0 || k == chainLen) { if ((int32_t)P.qn > W_CLIP_LEN && (double)(1 - ((float)e / (int32_t)P.qn)) < W_CLIP_SIM) rare = true; }             /* :1840, :2172 */
                else { const int32_t a = (int32_t)P.qn - (int32_t)P.tn; if ((a < 0 ? -a : a) >= W_SPLIT_LEN && (double)(1 - ((float)e / (int32_t)P.qn)) < W_SPLIT_SIM) rare = true; }   /* :1952 */
                nm -= e;
                const uint32_t tcons = P.mode == 0 ? P.tn : (uint32_t)(en + 1);
                it.n = nops; it.round = 0; it.ops_begin = ops_off[di] + ((uint64_t)P.qn + P.tn - nops); it.slot = (uint32_t)di; it.qn = P.qn; it.tcons = tcons; it.lazy = (uint8_t)lazy;
                if (k == 0) { it.kind = LF_RI_OPS_REV; it.tpos = P.ts + P.tn - tcons; pos = s[0].x - (uint32_t)en - 1; qStart = 0; }                           /* :1875-1887 */
                else { it.kind = LF_RI_OPS_FWD; it.tpos = P.ts; if (k == chainLen) { posEnd = P.ts + (uint32_t)en; qEnd = L; } }                               /* :2204-2207 */
            } else if (P.kind == 2) {
                it.kind = LF_RI_RUN_I; it.n = P.qn;
                if (k != 0 && k != chainLen) nm -= (int32_t)P.qn;                                      /* :2119 */
            } else if (P.kind == 3) { it.kind = LF_RI_DEL; it.n = P.tn; it.tpos = P.ts; nm -= (int32_t)P.tn; }                                                 /* :2126-2135 */
            items[ib + 2 * (uint64_t)k] = it;
            if (k >= 1) { lf_ritem_t m; memset(&m, 0, sizeof m); m.kind = LF_RI_RUN_M; m.n = s[k - 1].y >> 20; items[ib + 2 * (uint64_t)k - 1] = m; }
        }
    }
    rare = lf_any(rare) != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nm += __shfl_xor(nm, o);
    /* head values live in lane 0 of the first tile, tail values in the lane of slot chainLen of the last tile */
    const int tail_lane = (int)(chainLen & 63);
    const uint32_t pos0 = __shfl(pos, 0), qs0 = __shfl(qStart, 0);
    const uint32_t pe = __shfl(posEnd, tail_lane), qe = __shfl(qEnd, tail_lane);
    if (lane == 0) {
        lf_wrec_t w; memset(&w, 0, sizeof w);
        /* lane 0's pos / qStart are from the FIRST tile only if chainLen < 64 ... (they are set when k == 0, i.e. base == 0: kept) */
        w.pos = pos0; w.qStart = qs0; w.posEnd = pe; w.qEnd = qe; w.nm = nm; w.rare = rare ? 1u : 0u;
        wrec[j] = w;
        recs[j].item0 = (uint32_t)ib; recs[j].nitems = rare ? 0u : 2 * chainLen + 1;
    }
}

struct lf_slots_op { const lf_wjob_t *j; __device__ uint64_t operator()(uint32_t i) const { return (uint64_t)j[i].chain_len + 1; } };
struct lf_items_op { const lf_wjob_t *j; __device__ uint64_t operator()(uint32_t i) const { return 2ull * j[i].chain_len + 1; } };

#define WSLOT(T, k, bytes) (T *)lfg_dev_slot(dv, LF_DS_WALK0 + (k), (bytes))

/* plan: jobs (host, pinned) -> descriptors on the device.  out: counts and the device arrays the later steps use. */
extern "C" int lfg_walk_plan(const struct lf_index *ix, int n_jobs, const lf_wjob_t *jobs, const lfg_vc_t *vc, int lazy, lfg_walk_t *W)
{
    memset(W, 0, sizeof *W);
    if (n_jobs == 0) return LF_OK;
    const int dv = ix->device;
    HIPCHK(hipSetDevice(dv));
    hipStream_t s = (hipStream_t)lfg_lane_stream(dv, 1);
    if (!s) return LF_ERR_HIP;
    const size_t J = (size_t)n_jobs;
    lf_wjob_t *d_jobs = WSLOT(lf_wjob_t, 0, J * sizeof(lf_wjob_t));
    uint32_t *d_nd = WSLOT(uint32_t, 1, J * 4 + 16);
    uint64_t *d_ob = WSLOT(uint64_t, 2, J * 8 + 16);
    uint8_t *d_rare = WSLOT(uint8_t, 3, J + 16);
    uint64_t *d_dbase = WSLOT(uint64_t, 4, (J + 1) * 8), *d_obase = WSLOT(uint64_t, 5, (J + 1) * 8), *d_sbase = WSLOT(uint64_t, 6, (J + 1) * 8), *d_ibase = WSLOT(uint64_t, 7, (J + 1) * 8);
    unsigned long long *d_tot = WSLOT(unsigned long long, 8, LF_WALK_TOT * 64);
    uint64_t *h = (uint64_t *)lfg_pin_slot(LF_PS_WALK0 + 0, 256 + LF_WALK_TOT * 64);
    if (!d_jobs || !d_nd || !d_ob || !d_rare || !d_dbase || !d_obase || !d_sbase || !d_ibase || !d_tot || !h) return LF_ERR_NOMEM;
    lf_walk_dev D;
    D.read_off = (const uint64_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 1, 0);
    D.chain_seeds = (const uint2 *)vc->d_chain_seeds; D.chain_off = (const uint64_t *)vc->d_chain_off;
    D.ctg_off = (const int64_t *)vc->d_ctg; D.ctg_len = D.ctg_off + ix->n_seqs; D.n_ctg = ix->n_seqs; D.l_pac = ix->l_pac;
    if (!D.read_off || !D.chain_seeds || !D.chain_off || !D.ctg_off) { lf_set_error("lfg_walk_plan: no resident chains"); return LF_ERR_ARG; }
    HIPCHK(hipMemcpyAsync(d_jobs, jobs, J * sizeof(lf_wjob_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_tot, 0, LF_WALK_TOT * 64, s));
    hipLaunchKernelGGL(lf_walk_plan_kernel<false>, dim3((unsigned)n_jobs), dim3(64), 0, s, n_jobs, (const lf_wjob_t *)d_jobs, D, lazy, d_nd, d_ob, d_rare,
                       (const uint64_t *)nullptr, (const uint64_t *)nullptr, (const uint64_t *)nullptr, (lf_aln_desc_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr, d_tot);
    lf_slots_op so; so.j = d_jobs; lf_items_op io; io.j = d_jobs;
    /* scans over n_jobs + 1 inputs (the extra one is padding past the arrays' used part: zeroed) so that base[n_jobs] = total */
    HIPCHK(hipMemsetAsync(d_nd + J, 0, 4, s)); HIPCHK(hipMemsetAsync(d_ob + J, 0, 8, s));
    { lf_scan_u32 f; f.p = d_nd; const int src = lf_scan_excl(dv, 3, s, f, d_dbase, (size_t)n_jobs + 1); if (src != LF_OK) return src; }
    { lf_scan_u64 f; f.p = d_ob; const int src = lf_scan_excl(dv, 3, s, f, d_obase, (size_t)n_jobs + 1); if (src != LF_OK) return src; }
    { const int src = lf_scan_excl(dv, 3, s, so, d_sbase, (size_t)n_jobs); if (src != LF_OK) return src; }
    { const int src = lf_scan_excl(dv, 3, s, io, d_ibase, (size_t)n_jobs); if (src != LF_OK) return src; }
    /* slots and items follow from the chain lengths (host); a chain of L anchors has at most L + 1 pieces, so the slots bound the descriptors too:
     * the second pass is launched without waiting for the first one's totals (they come back with the statistics, one wait for both) */
    uint64_t n_slots = 0, n_items = 0;
    for (size_t k = 0; k < J; k++) { n_slots += (uint64_t)jobs[k].chain_len + 1; n_items += 2ull * jobs[k].chain_len + 1; }
    if (n_slots >= (1ull << 31) || n_items >= 0xffffffffull) { lf_set_error("lfg_walk_plan: too many alignment pieces in one chunk"); return LF_ERR_ARG; }
    /* the round's descriptors live next to its paths (slot of extension round 0), like the host-planned rounds' */
    lf_aln_desc_t *d_desc = (lf_aln_desc_t *)lfg_dev_slot(dv, LF_DS_RND0 + 1, (n_slots + 1) * sizeof(lf_aln_desc_t));
    uint64_t *d_opsoff = WSLOT(uint64_t, 10, (n_slots + 1) * 8);
    int32_t *d_slot_desc = WSLOT(int32_t, 11, (n_slots + 1) * 4);
    if (!d_desc || !d_opsoff || !d_slot_desc) return LF_ERR_NOMEM;
    hipLaunchKernelGGL(lf_walk_plan_kernel<true>, dim3((unsigned)n_jobs), dim3(64), 0, s, n_jobs, (const lf_wjob_t *)d_jobs, D, lazy, d_nd, d_ob, d_rare,
                       (const uint64_t *)d_dbase, (const uint64_t *)d_obase, (const uint64_t *)d_sbase, d_desc, d_opsoff, d_slot_desc, d_tot);
    HIPCHK(hipMemcpyAsync(h, d_dbase + J, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h + 1, d_obase + J, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h + 32, d_tot, LF_WALK_TOT * 64, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    const uint64_t n_desc = h[0], ops_total = h[1];
    if (n_desc > n_slots) { lf_set_error("lfg_walk_plan: %llu pieces for %llu slots", (unsigned long long)n_desc, (unsigned long long)n_slots); return LF_ERR_HIP; }
    for (int k = 0; k < 6; k++) { uint64_t t = 0; for (int u = 0; u < LF_WALK_TOT; u++) t += h[32 + 8 * u + k]; h[4 + k] = t; }
    W->n_jobs = n_jobs; W->n_desc = n_desc; W->ops_total = ops_total; W->n_items = n_items; W->ext_bytes = h[4]; W->block_steps = h[5];
    W->hc.roots = h[6]; W->hc.cap = h[7]; W->hc.sum_n = h[8]; W->hc.sum_m = h[9];
    W->d_jobs = d_jobs; W->d_rare = d_rare; W->d_sbase = d_sbase; W->d_ibase = d_ibase; W->d_desc = d_desc; W->d_opsoff = d_opsoff; W->d_slot_desc = d_slot_desc;
    return LF_OK;
}

/* emit: after the alignments.  Record fields come back in a pinned array (valid until this lane's next call); the recipe
 * (records + items) stays on the device for lfg_render. */
extern "C" int lfg_walk_emit(const struct lf_index *ix, const lfg_vc_t *vc, int lazy, lfg_walk_t *W, const void *d_ed, const void *d_end, const void *d_len, lf_wrec_t **wrec_out)
{
    *wrec_out = nullptr;
    if (W->n_jobs == 0) return LF_OK;
    const int dv = ix->device;
    HIPCHK(hipSetDevice(dv));
    hipStream_t s = (hipStream_t)lfg_lane_stream(dv, 1);
    if (!s) return LF_ERR_HIP;
    const size_t J = (size_t)W->n_jobs;
    lf_rrecord_t *d_recs = WSLOT(lf_rrecord_t, 12, (J + 1) * sizeof(lf_rrecord_t));
    lf_ritem_t *d_items = WSLOT(lf_ritem_t, 13, (W->n_items + 1) * sizeof(lf_ritem_t));
    lf_wrec_t *d_wrec = WSLOT(lf_wrec_t, 14, (J + 1) * sizeof(lf_wrec_t));
    lf_wrec_t *h_wrec = (lf_wrec_t *)lfg_pin_slot(LF_PS_WALK0 + 1, (J + 1) * sizeof(lf_wrec_t));
    if (!d_recs || !d_items || !d_wrec || !h_wrec) return LF_ERR_NOMEM;
    lf_walk_dev D;
    D.read_off = (const uint64_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 1, 0);
    D.chain_seeds = (const uint2 *)vc->d_chain_seeds; D.chain_off = (const uint64_t *)vc->d_chain_off;
    D.ctg_off = (const int64_t *)vc->d_ctg; D.ctg_len = D.ctg_off + ix->n_seqs; D.n_ctg = ix->n_seqs; D.l_pac = ix->l_pac;
    hipLaunchKernelGGL(lf_walk_emit_kernel, dim3((unsigned)W->n_jobs), dim3(64), 0, s, W->n_jobs, (const lf_wjob_t *)W->d_jobs, D, lazy, (const uint8_t *)W->d_rare,
                       (const uint64_t *)W->d_sbase, (const int32_t *)W->d_slot_desc, (const uint64_t *)W->d_ibase,
                       (const int32_t *)d_ed, (const int32_t *)d_end, (const uint32_t *)d_len, (const uint64_t *)W->d_opsoff, d_recs, d_items, d_wrec);
    HIPCHK(hipMemcpyAsync(h_wrec, d_wrec, J * sizeof(lf_wrec_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    W->d_recs = d_recs; W->d_items = d_items;
    *wrec_out = h_wrec;
    return LF_OK;
}
