/* lf_internal.h -- private to liblfgpu.so (host C glue <-> HIP translation units). */
#ifndef LF_INTERNAL_H
#define LF_INTERNAL_H

#include <stdint.h>
#include <stddef.h>
#include "lordfast_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int64_t offset; int32_t len; char *name; } lf_contig_t;

/* host copy of what the glue needs + opaque device handle */
struct lf_index {
    /* .bwt header */
    uint64_t primary, L2[5], seq_len, bwt_size;
    uint64_t sa_intv, n_sa;
    int64_t  l_pac;
    int32_t  n_seqs;
    lf_contig_t *contigs;
    uint8_t *pac;            /* host copy (glue: MD strings, SAM) */
    int      device;
    unsigned flags;
    void    *dev;            /* struct lf_dev_index* (lf_gpu.hip) */
};

void lf_set_error(const char *fmt, ...);

/* ---- lf_gpu.hip ---- */
int  lfg_device_count(void);
int  lfg_index_upload(struct lf_index *ix, const uint32_t *bwt, const uint64_t *sa_sampled);
void lfg_index_free(struct lf_index *ix);
int  lfg_index_describe(const struct lf_index *ix, char *buf, size_t cap);

typedef struct {        /* raw device results of the seed stage, copied to host */
    uint64_t n_hits;
    uint32_t *tpos;      /* n_hits */
    uint32_t *qpl;       /* n_hits: qPos | len << 20  (Seed_t bit layout) */
    uint8_t  *strand;    /* n_hits: 1 = reverse */
    uint64_t *read_off;  /* n_reads + 1: first hit of each read */
    uint64_t counters[4];
    float ms_search, ms_accept, ms_locate;
} lfg_hits_t;

/* want_hits == 0: the hits stay in HBM for lfg_vote_chain (out gets n_hits, read_off and the counters only) */
/* the read batch as three bit planes made on the host (bit i of word i / 64 of plane x describes base i: code low bit, code high
 * bit, "is one of ACGT", upper case; planes[x * qw + w]) plus the bytes that are NOT upper-case ACGT: 3 / 8 of the bytes on the link */
typedef struct {
    const uint64_t *planes; uint64_t qw; const uint64_t *exc_pos; const uint8_t *exc_byte; uint64_t n_exc;
    /* a chunk of a PREPACKED batch (lf_batch.h): its bases are bits [64 word0 + shift, ... + n_bases) of planes whose stride is src_qw words; the exception
     * positions count from exc_base.  src_qw == 0: the planes are this chunk's own (stride qw, first base at bit 0). */
    uint64_t src_qw, word0, exc_base; uint32_t shift;
} lf_packed_src_t;
int  lfg_seed_packed(const struct lf_index *ix, const lf_params_t *p, int n_reads, const lf_packed_src_t *pk, const uint64_t *off, int want_hits, lfg_hits_t *out);
int  lfg_seed(const struct lf_index *ix, const lf_params_t *p, int n_reads, const char *reads,
              const uint64_t *off, int want_hits, lfg_hits_t *out);

int  lfg_seed_src(const struct lf_index *ix, const lf_params_t *p, int n_reads, const char *reads, const void *d_src, const uint64_t *src_off,
                  const uint64_t *off, int want_hits, lfg_hits_t *out);
int  lfg_gather_reads(int device, void *stream, const void *d_src, const uint64_t *d_src_off, const uint64_t *d_off, int n_reads, void *d_dst);

/* ---- lf_vote.hip: votes -> candidate windows -> chains, on the hits that lfg_seed left in HBM ---- */
typedef struct {
    int n_reads, n_req;
    uint8_t *mode;            /* per read: 1 no window, 2 coarse (one request), 3 fine (nreq candidate requests) */
    uint64_t *req0;           /* n_reads + 1: first request of each read */
    uint32_t *nreq; float *vscore;
    uint32_t *req_win;        /* per request: window id | isReverse << 31 */
    uint32_t *chain_len; float *chain_score; uint64_t *chain_off; Seed_t *chain_seeds;      /* chain_seeds: NULL (in HBM only) until lfg_vote_fetch_chains */
    uint64_t n_req_seeds, n_chain_seeds, n_tie_req;
    float ms_vote, ms_chain;
    /* the same chains in HBM (valid until this lane's next lfg_vote_chain): what lf_walk.hip works on */
    const void *d_chain_seeds, *d_chain_off, *d_chain_len, *d_ctg;
} lfg_vc_t;
int lfg_vote_chain(const struct lf_index *ix, const lf_params_t *p, int n_reads, uint64_t n_hits, uint32_t max_read_len, lfg_vc_t *out);
int lfg_vote_fetch_chains(const struct lf_index *ix, lfg_vc_t *vc);      /* out->chain_seeds is NULL after lfg_vote_chain: the chains stay in HBM unless this is called */
void lfg_hits_free(lfg_hits_t *h);

int lfg_chain_n2(int device, const lf_params_t *p, int n_windows, const Seed_t *sorted_seeds,
                 const uint64_t *off, uint32_t *chain_idx, uint32_t *chain_len, float *score, float *ms);
/* clasp (lf_clasp_kernel.h): seeds in the caller's order; chain seeds land at chain_out[off[w] ..] */
int lfg_chain_clasp(int device, int n_windows, const Seed_t *seeds, const uint64_t *off,
                    Seed_t *chain_out, uint32_t *chain_len, float *score, float *ms);

/* edlib problems: sequences given as byte strings (host memory); ops end-aligned in [qoff[i] + toff[i], + n_i + m_i) */
int lfg_edlib(int device, int n, const char *q, const uint64_t *qoff, const char *t, const uint64_t *toff,
              const uint8_t *mode, int32_t *ed, int32_t *endloc, uint8_t *ops, uint32_t *ops_len, float *ms);
int lfg_ksw(int device, int n, const uint8_t *q, const uint64_t *qoff, const uint8_t *t, const uint64_t *toff,
            const int32_t *prm, int32_t *score, int32_t *qle, int32_t *tle, float *ms);

/* ---- lf_mem.hip: persistent grow-only buffers ---- */
void *lfg_dev_slot(int device, int slot, size_t bytes);
void *lfg_pin_slot(int slot, size_t bytes);
void  lfg_slots_release(void);
void *lfg_host_alloc(size_t bytes);    /* pinned host memory outside the slot system */
void  lfg_host_free(void *p);
void  lfg_lane_set_value(int device, int key, uint64_t v);   /* per-lane scratch numbers, key < 8; 0: words of one bit plane of the resident read batch (lf_seed.hip), 2 / 3: lines of the lane's two SAM buffers (lf_sam.hip), 4 .. 6: Hirschberg statistics (lf_align.hip) */
uint64_t lfg_lane_value(int device, int key);
void  lfg_set_lane(int lane);          /* calling thread drives lane 0 or 1 (own slots + streams) */
int   lfg_get_lane(void);
void  lfg_phase(const char *file, int line);      /* LF_WATCHDOG: where a lane is */
void  lfg_phase_dump(void);
uint64_t lfg_take_waits(void);                     /* host waits of the calling thread since the last call */
void  lfg_quiesce(int device);                     /* per-stream waits before the runtime's own device-wide ones */
void  lfg_drain_check(int device);                 /* LF_WATCHDOG: name the stream that never drains */
void *lfg_lane_stream(int device, int which);   /* persistent hipStream_t of the calling thread's lane */
void *lfg_lane_event(int device, int which);    /* persistent hipEvent_t (timing enabled) of the calling thread's lane; which < 48 */
/* the environment (lf_host.c): lf_env_set = the variable exists; lf_env_long = its value as a number, or the default */
const char *lf_env(const char *name);
long lf_env_long(const char *name, long dflt);
int  lf_env_set(const char *name);
/* slot ids */
enum { LF_DS_SEED0 = 0 /* ..15 */, LF_DS_CHAIN0 = 16 /* ..23 */, LF_DS_ALN0 = 24 /* ..47 */, LF_DS_KSW0 = 48 /* ..55 */,
       LF_DS_RND0 = 56 /* ..87: 2 per extension round (ops, spare) */, LF_DS_RENDER0 = 88 /* ..95 */, LF_DS_VOTE0 = 96 /* ..127 */,
       LF_DS_WALK0 = 128 /* ..147 */, LF_DS_SAM0 = 148 /* ..157 */, LF_DS_HIRSCH0 = 158 /* ..167 */, LF_DS_SCAN0 = 168 /* ..175: lf_scan.h workspaces */ };
#define LF_MAX_ED_ROUNDS 16
enum { LF_PS_READS = 0, LF_PS_READOFF = 1, LF_PS_HITS_T = 2, LF_PS_HITS_Q = 3, LF_PS_HITS_S = 4, LF_PS_HITS_OFF = 5,
       LF_PS_CHAIN_SEEDS = 6, LF_PS_CHAIN_IDX = 7, LF_PS_ALN_Q = 8, LF_PS_ALN_T = 9, LF_PS_ALN_PROB = 10 /* ..16 */,
       LF_PS_ROUND0 = 20 /* 4 per round: ed, end, len, ops ; up to 16 rounds */, LF_PS_RENDER0 = 84 /* ..91 */, LF_PS_VOTE0 = 92 /* ..107 */,
       LF_PS_WALK0 = 108 /* ..111 */, LF_PS_RENDER1 = 112 /* ..115 */, LF_PS_SAM0 = 116 /* ..123 */, LF_PS_HOSTBASES = 124, LF_PS_EXC_POS = 125, LF_PS_EXC_BYTE = 126, LF_PS_FETCH_META = 127 };

/* alignment request as a DESCRIPTOR into HBM-resident data: query = the read batch uploaded by the seed stage,
 * target = the 2-bit reference.  Element i = base[start +/- i], optionally complemented (flags LF_F_*). */
typedef struct { int64_t qstart, tstart; uint32_t n, m; uint8_t flags, mode, pad[6]; } lf_aln_desc_t;
/* ops == NULL: the edit paths stay in HBM, in the device slot `ops_slot` (address returned in *ops_dev) */
int lfg_edlib_desc(const struct lf_index *ix, int n, const lf_aln_desc_t *d, const uint64_t *ops_off, uint64_t ops_total,
                   int32_t *ed, int32_t *endloc, uint8_t *ops, uint32_t *ops_len, int ops_slot, void **ops_dev, void **desc_dev, float *ms);

/* the problems of a descriptor batch that lie above edlib's traceback switch (roots of the Hirschberg levels, lf_hirsch.hip):
 * their number, the sum of their piece bounds (lf_hroot_cap) and of their query / target lengths -- counted by whoever makes
 * the descriptors, so that the scratch of the levels can be sized without a round trip */
typedef struct { uint64_t roots, cap, sum_n, sum_m; } lf_hcount_t;
typedef struct { const void *d_text, *d_offs, *d_lens; } lfg_rtext_t;       /* rendered text on the device: offsets / lengths (incl. NUL) 2 per record */
/* ---- lf_render.hip: CIGAR / MD text from the paths in HBM ---- */
enum { LF_RI_RUN_M = 0, LF_RI_RUN_I = 1, LF_RI_OPS_FWD = 2, LF_RI_OPS_FWD_TRC = 3, LF_RI_OPS_REV = 4, LF_RI_DEL = 5 };
typedef struct {            /* one piece of a record, in output order (32 bytes) */
    uint64_t ops_begin;     /* OPS_*: first op of the path inside its round's ops buffer */
    uint32_t n;             /* elements: ops of the path / length of the run / deleted bases */
    uint32_t tpos;          /* reference position of the first non-insert element (OPS_FWD_TRC: walks downwards, complemented) */
    uint32_t slot;          /* OPS_*: the problem's descriptor in its round (lazy paths: where its query / target bases are) */
    uint32_t qn, tcons;     /* OPS_*: query bases / reference bases the path consumes */
    uint8_t kind, round, lazy, pad;
} lf_ritem_t;
typedef struct { uint32_t item0, nitems; } lf_rrecord_t;
/* records 0 .. n_dev_recs-1 and their items are already on the device (lfg_walk_emit: d_recs / d_items), the host's follow */
int lfg_render(const struct lf_index *ix, int n_dev_recs, const void *d_recs_dev, uint64_t n_dev_items, const void *d_items_dev,
               int n_recs, const lf_rrecord_t *recs, uint64_t n_items, const lf_ritem_t *items,
               const void *const *round_ops, const void *const *round_desc, char **text_out, uint64_t **offs_out, uint64_t *text_bytes, float *ms,
               lfg_rtext_t *dev_text /* != NULL: the text stays on the device (only offsets / lengths come back in *offs_out / *lens_out) */, uint32_t **lens_out);
/* ---- lf_walk.hip: the common path of alignChain_edlib on the device ---- */
typedef struct { uint32_t req, read, chain_len; uint8_t is_rev, pad[3]; } lf_wjob_t;      /* one kept candidate window: chain request, read (index in the seed batch) */
typedef struct { uint32_t pos, posEnd, qStart, qEnd; int32_t nm; uint32_t rare, pad[2]; } lf_wrec_t;   /* SAM record fields of a job; rare: replay it on the host */
typedef struct {
    int n_jobs; uint64_t n_desc, ops_total, n_items, ext_bytes, block_steps; lf_hcount_t hc;
    void *d_jobs, *d_rare, *d_sbase, *d_ibase, *d_desc, *d_opsoff, *d_slot_desc, *d_recs, *d_items;
} lfg_walk_t;
int lfg_walk_plan(const struct lf_index *ix, int n_jobs, const lf_wjob_t *jobs, const lfg_vc_t *vc, int lazy, lfg_walk_t *W);
int lfg_walk_emit(const struct lf_index *ix, const lfg_vc_t *vc, int lazy, lfg_walk_t *W, const void *d_ed, const void *d_end, const void *d_len, lf_wrec_t **wrec_out);
void lfg_edlib_breakdown(float *out4);      /* event brackets of the calling thread's last alignment batch: forward, traceback, Hirschberg levels, binning */
float lfg_edlib_round_ms(void);
/* alignments of device-resident descriptors (lfg_walk_plan); results stay on the device */
int lfg_edlib_desc_dev(const struct lf_index *ix, int n, const void *d_desc, const void *d_opsoff, uint64_t ops_total, const lf_hcount_t *hc, int ops_slot,
                       void **ops_dev, void **ed_dev, void **end_dev, void **len_dev, float *ms);
/* ---- lf_sam.hip: SAM lines assembled on the device ---- */
enum { LF_SL_MAPPED = 0, LF_SL_UNMAPPED = 1, LF_SL_LITERAL = 2 };
typedef struct {
    uint32_t name_off; uint16_t name_len, flag;     /* name in the chunk's name blob; SAM flag as printed */
    uint32_t read;                                  /* index in the resident read batch (SEQ / QUAL source) */
    int32_t  rname;                                 /* contig id */
    uint32_t pos1;                                  /* 1-based position */
    int32_t  mapq, as; uint32_t nm;                 /* as printed (MAPQ clamped at 0, NM absolute) */
    uint32_t rec;                                   /* record of the rendered CIGAR / MD text */
    uint32_t sa_off, sa_len;                        /* SA:Z value in the side blob (0: none); LITERAL: the whole line */
    uint8_t  kind, is_fq, pad[2];
} lf_samline_t;                                     /* 48 bytes */
int lfg_sam_build(const struct lf_index *ix, const lf_params_t *p, int n_lines, const lf_samline_t *lines,
                  const char *names, uint64_t names_bytes, const char *blob, uint64_t blob_bytes,
                  const char *quals, uint64_t quals_bytes, const void *d_quals_src, int n_batch_reads,
                  const lfg_rtext_t *rt, int parity, int holes, uint64_t *total_out, const uint64_t **h_offs_out, const uint32_t **h_hole_out);
int lfg_host_mapped(int device, const void *p, size_t bytes);      /* kernels of `device` can store into [p, p + bytes) (pinned host memory) */
int lfg_sam_fetch(const struct lf_index *ix, char *dst, uint64_t total, int parity);
int lfg_sam_wait(const struct lf_index *ix);
int lfg_sam_fetch_async(const struct lf_index *ix, char *dst, uint64_t total, int parity);
int lfg_sam_fetch_wait(const struct lf_index *ix);
int lfg_fetch(int device, void *dst, const void *src_dev, size_t bytes);
int lfg_fetch_gather(int device, int n, void *host_base, const uint64_t *host_off, const void *const *src_dev, const size_t *bytes, uint64_t total);
int lfg_upload(int device, void *dst_dev, const void *src, size_t bytes);
#define LF_F_QREV  1u
#define LF_F_QCOMP 2u
#define LF_F_TREV  4u
#define LF_F_TCOMP 8u
#define LF_F_TPAC  16u
#define LF_F_LAZYX 32u   /* diagonal moves are written as op 0 without comparing the bases; lf_render_kernel resolves match / mismatch */

#define LF_TASK_PATH 0
#define LF_TASK_DIST 1

/* longest query the sweep kernels of lf_align.hip take (64 lanes x 8 blocks x 64 rows).  Problems above edlib's traceback
 * switch never reach them whole: the Hirschberg levels (lf_hirsch.hip, any query length) cut them into leaves first. */
#define LF_SWEEP_MAX_N 32768
/* edlib's own leaf/Hirschberg switch (lib/edlib/edlib.cpp:1117-1119) */
#ifdef __HIPCC__
#define LF_HD __host__ __device__
#else
#define LF_HD
#endif
static inline LF_HD int lf_is_leaf(int64_t n, int64_t m) { return 20LL * ((n + 63) / 64) * m + 8LL * m < 1024 * 1024; }
/* upper bound of the pieces (leaves of edlib's recursion tree) of a problem above the switch: internal nodes only exist while
 * 20 ceil(n / 64) m + 8 m >= 2^20, and m halves per level */
static inline LF_HD uint32_t lf_hroot_cap(uint32_t n, uint32_t m)
{
    const uint64_t x = ((uint64_t)m * (20ull * ((n + 63) >> 6) + 8) + (1u << 20) - 1) >> 20;
    return (uint32_t)(2 * x + 4);
}

#ifdef __cplusplus
}
#endif
#endif
