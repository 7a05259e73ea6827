/*
 * lordfast_amd.h -- C ABI of liblfgpu.so: the MI355X (gfx950) implementation of lordFAST's per-read
 * seed -> vote -> chain -> extend -> SAM hot path (reference: vpc-ccg/lordfast v0.0.10).
 *
 * Plain C: pointers and sizes only.  Two layers:
 *
 *   1. Batch API (lf_*)  -- what a GPU wants: one call per read batch / problem batch.
 *   2. Drop-in entry points -- the reference's own internal function names and argument meaning
 *      (bwt_load, getLocs_extend_whole_step, chain_seeds_n2, edlibAlign, ksw_extend2, mapSeqMT ...),
 *      thin single-item wrappers over layer 1 that operate on one process-global index, exactly like
 *      the reference's globals.  Each cites the reference declaration it replaces.
 *
 * All compute goes through HIP kernels; there is no CPU fallback: every entry point that needs the
 * device returns LF_ERR_NO_DEVICE (or aborts with a message, for the void drop-in calls) when no
 * gfx950 device / code object is available.
 */
#ifndef LORDFAST_AMD_H
#define LORDFAST_AMD_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_OK             0
#define LF_ERR_IO         1
#define LF_ERR_NO_DEVICE  2
#define LF_ERR_ARG        3
#define LF_ERR_HIP        4
#define LF_ERR_NOMEM      5

/* ------------------------------------------------------------------------------------------------
 * Data model (reference: src/LordFAST.h:30-75)
 * ---------------------------------------------------------------------------------------------- */

/* Seed_t (src/LordFAST.h:30-35): same 8-byte layout, qPos 20 bits / len 12 bits. */
typedef struct {
    uint32_t tPos;
    uint32_t qPos : 20;
    uint32_t len  : 12;
} Seed_t;

typedef struct { Seed_t *list; uint32_t num; } SeedList;                 /* src/LordFAST.h:37-41 */
typedef struct { Seed_t *seeds; uint32_t chainLen; float score; } Chain_t; /* src/LordFAST.h:65-70 */

/* option globals of src/CommandLineParser.cpp:41-55 gathered in one struct */
typedef struct {
    int    min_anchor_len;   /* -k MIN_ANCHOR_LEN   14   [12..20] */
    int    sampling_count;   /* -c SAMPLING_COUNT   1000 */
    int    max_map;          /* -n MAX_MAP          10   */
    int    min_read_len;     /* -l MIN_READ_LEN     1000 (>= 100) */
    int    max_ref_hits;     /* -m MAX_REF_HITS     1000 */
    int    chain_alg;        /* -a 0 = dp-n2, 1 = clasp */
    double chain_reward;     /* --chainReward  9.3  */
    double chain_penalty;    /* --chainPenalty 11.4 */
    double gap_penalty;      /* --gapPenalty   0.15 */
    int    threads;          /* -t host glue threads (0 = all online CPUs, max 255) */
    char   read_group_id[256]; /* -R ... ID, empty = none */
    char   read_group[1000];   /* -R: the whole @RG header line (escapes resolved), printed by lf_sam_header */
} lf_params_t;

void lf_params_default(lf_params_t *p);
/* set_read_group (src/CommandLineParser.cpp:85-124): `rg_line` must start with "@RG", contain no literal tab and an
 * "ID:" field; \t \n \r \\ escapes are resolved. Fills read_group and read_group_id. Non-zero + lf_last_error on error. */
int  lf_params_set_read_group(lf_params_t *p, const char *rg_line);

/* ------------------------------------------------------------------------------------------------
 * Device / index
 * ---------------------------------------------------------------------------------------------- */
typedef struct lf_index lf_index_t;      /* FM-index + pac + contigs, resident in HBM */

/* number of usable gfx950 devices (0 = none; never throws) */
int lf_device_count(void);
const char *lf_last_error(void);

/* Load <prefix>.bwt/.sa/.pac/.ann/.amb (formats: SURVEY App. A; lib/bwa/bwt.c:421-462,
 * lib/bwa/bntseq.c:100-160) and put them in HBM of `device`.  The 256 MiB <prefix>.cache of
 * src/BWT.cpp:159-187 is NOT read: the 12-mer table is rebuilt on the GPU (src/BWT.cpp:60-115).
 * flags: LF_IDX_FULL_SA materialises the whole suffix array in HBM so that locate is one load
 * instead of a ~31-step LF walk (results identical). */
#define LF_IDX_FULL_SA 1u
int  lf_index_load(const char *prefix, int device, unsigned flags, lf_index_t **out);
void lf_index_free(lf_index_t *idx);
/* bwt_index (src/BWT.cpp:140-157 over lib/bwa/bwtindex.c:187-293): build <fasta>.pac/.ann/.amb/.bwt/.sa/.cache on
 * the GPU, byte-identical to the reference's indexer (plain or gzip FASTA). */
int  lf_index_build(const char *fasta_path, int device);
uint32_t lf_index_genome_len(const lf_index_t *idx);          /* bwt_get_refGenLen, src/BWT.cpp:305 */
int  lf_index_n_contigs(const lf_index_t *idx);
const char *lf_index_contig(const lf_index_t *idx, int i, int64_t *offset, int32_t *len);
/* one line naming the index structures that are resident in HBM and their sizes (full SA; which of the 12- / 14- / 16-mer
 * tables were built -- the wide ones are dropped when HBM is short, which changes speed, never results) */
int  lf_index_describe(const lf_index_t *idx, char *buf, size_t cap);

/* ------------------------------------------------------------------------------------------------
 * Stage 1: seeding -- getLocs_extend_whole_step for a batch of reads (src/BWT.cpp:312-394)
 *   reads: concatenated bases (ASCII, any case), read i = reads[off[i] .. off[i+1])
 *   result: for read i the forward seeds F[offF[i]..offF[i+1]) and reverse seeds R[offR[i]..offR[i+1]),
 *   same content and order as the reference's two SeedLists.  Buffers are owned by the result.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int       n_reads;
    uint64_t *offF, *offR;        /* n_reads + 1 each */
    Seed_t   *F, *R;
    /* algorithmic-byte counters of SURVEY 8(d), summed over the batch */
    uint64_t  n_cache, n_occblk, n_sa, n_readbytes;
    /* kernel time (ms, HIP events on the launch stream): search, accept+scan, locate */
    float     ms_search, ms_accept, ms_locate;
} lf_seeds_t;

int  lf_seed_batch(const lf_index_t *idx, const lf_params_t *p, int n_reads,
                   const char *reads, const uint64_t *off, lf_seeds_t **out);
void lf_seeds_free(lf_seeds_t *s);

/* ------------------------------------------------------------------------------------------------
 * Stage 3: dp-n2 chaining for a batch of windows (src/Chain.cpp:232-310)
 *   seeds of window w: seeds[off[w]..off[w+1]) -- REORDERED in place exactly as the reference's
 *   std::sort by qPos does; chain indices (into the reordered window) are written to
 *   chain_idx[off[w] .. off[w]+chain_len[w]) ; score[w] is the float the reference stores.
 * ---------------------------------------------------------------------------------------------- */
int lf_chain_n2_batch(const lf_params_t *p, int n_windows, Seed_t *seeds, const uint64_t *off,
                      uint32_t *chain_idx, uint32_t *chain_len, float *score, int device);

/* ------------------------------------------------------------------------------------------------
 * Stage 3b: clasp chaining (`-a clasp`) for a batch of windows: chain_seeds_clasp (src/Chain.cpp:39-209) over
 *   lib/clasp's SOP-gap-cost chaining (lib/clasp/slchain.c:568-974), lambda 0.15, eps 0.
 *   seeds of window w: seeds[off[w]..off[w+1]) in the caller's order (not modified); the chain, in target
 *   order, is written to chain_out[off[w] .. off[w]+chain_len[w]); score[w] is the float the reference stores
 *   (-1 and chain_len 0 for an empty window).  Like clasp itself, positions must fit `int` (the reference's
 *   caller subtracts 2e9 first, src/LordFAST.cpp:684-692) and a window must be narrower than 2^28.
 * ---------------------------------------------------------------------------------------------- */
int lf_chain_clasp_batch(int n_windows, const Seed_t *seeds, const uint64_t *off,
                         Seed_t *chain_out, uint32_t *chain_len, float *score, int device);

/* ------------------------------------------------------------------------------------------------
 * Stage 4: edlib-equivalent alignment for a batch of problems (lib/edlib/edlib.cpp:101-221;
 * output function: SURVEY App. F).  Raw-byte comparison, k = -1, task = PATH.
 *   problem i: query q[qoff[i]..qoff[i+1]) vs target t[toff[i]..toff[i+1]), mode[i] 0 = NW, 1 = SHW
 *   out: edit_distance[i], end_location[i] (may be -1 in SHW), ops of problem i at
 *        ops[ops_off[i] .. ops_off[i] + ops_len[i]) with ops_off[i] = qoff[i] + toff[i]
 *        (0 match, 1 insertion/query base, 2 deletion/target base, 3 mismatch).
 * ---------------------------------------------------------------------------------------------- */
int lf_edlib_batch(int n, const char *q, const uint64_t *qoff, const char *t, const uint64_t *toff,
                   const uint8_t *mode, int32_t *edit_distance, int32_t *end_location,
                   uint8_t *ops, uint32_t *ops_len, int device, float *kernel_ms);

/* ------------------------------------------------------------------------------------------------
 * ksw_extend2 for a batch (lib/bwa/ksw.c:380-478), lordFAST's 5x5 clip matrix (+2 / -16 / N 0),
 * end_bonus 0.  q/t hold codes 0..4.  prm[i] = {o_del,e_del,o_ins,e_ins,w,zdrop,h0}.
 * ---------------------------------------------------------------------------------------------- */
int lf_ksw_extend2_batch(int n, const uint8_t *q, const uint64_t *qoff, const uint8_t *t,
                         const uint64_t *toff, const int32_t *prm, int32_t *score, int32_t *qle,
                         int32_t *tle, int device);

/* ------------------------------------------------------------------------------------------------
 * Whole path: mapSeq over a batch (src/LordFAST.cpp:461-580) -> SAM records in read order.
 *   names/seqs/quals: n NUL-terminated strings; quals may be NULL or quals[i] "" (FASTA -> "*").
 *   *sam is malloc'd (free with lf_free), *sam_len bytes, no header.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    double ms_total, ms_seed, ms_vote, ms_chain, ms_extend, ms_sam;   /* wall, host clock */
    float  ms_k_search, ms_k_accept, ms_k_locate, ms_k_chain, ms_k_edlib, ms_k_ksw; /* HIP events */
    uint64_t n_reads, n_bases, n_seeds, n_chain_problems, n_edlib_problems, n_ksw_problems;
    uint64_t n_cache, n_occblk, n_sa, n_readbytes;     /* SURVEY 8(d) seed counters */
    uint64_t ext_bytes;                                /* sum over problems (q + ceil(t/4) + q + t) */
    uint64_t edlib_launches, search_launches, locate_launches;
    /* CIGAR / MD rendering (lf_render_kernel): wall, HIP events, text bytes written, launches */
    double ms_render; float ms_k_render, ms_k_vote; uint64_t render_bytes, render_launches;
    uint64_t ops_bytes;                                 /* edit-path bytes left in HBM for the renderer */
    uint64_t n_req_seeds, n_tie_requests;              /* seeds gathered into chain requests; requests whose equal qPos needed the std::sort replay */
    uint64_t dp_block_steps;                            /* sum over alignment problems of ceil(q / 64) * t: Myers block steps of one forward pass */
    uint64_t ksw_bytes;                                 /* sum over ksw problems of qlen + tlen + 12: the two sequences read once, three results written */
    /* the alignment group's kernels one by one (HIP events): lf_edlib_rsweep_kernel (both modes), lf_edlib_tb_kernel, the
     * Hirschberg levels (incl. their readbacks), request binning */
    float ms_k_rsweep, ms_k_tb, ms_k_hirsch, ms_k_bin;
    uint64_t hirsch_bytes;                              /* sum over the problems above edlib's traceback switch of (q + ceil(t/4) + q + t): what the Hirschberg levels read and write */
    uint64_t n_host_waits, n_chunks;                    /* host waits for a stream / event on the chunk drivers' threads, and the chunks they drove */
    uint64_t hirsch_max_rows;                           /* the longest query among the problems above edlib's traceback switch */
    uint64_t hirsch_banded_nodes, hirsch_unbanded_nodes;        /* nodes of edlib's recursion swept inside a band (lf_hband_level_kernel) / by the unbanded sweeps (queries with more
                                                         * diagonals than sixteen wavefronts hold, targets with bytes other than ACGT) */
    uint64_t n_stale_first_windows;                     /* -a clasp, fine mode: candidate windows without seeds that had NO earlier chain of the same read to fall back on -- where the
                                                         * reference extends whatever chain its thread mapped last (src/Chain.cpp:68,92: scheduling-dependent) and this library none */
} lf_stats_t;

int  lf_map_batch(const lf_index_t *idx, const lf_params_t *p, int n, const char *const *names,
                  const char *const *seqs, const char *const *quals, char **sam, size_t *sam_len,
                  lf_stats_t *stats);
/* ------------------------------------------------------------------------------------------------
 * Either side of the path: FASTA / FASTQ (plain or gzip) reader with the reference's record grammar
 * (src/Reads.cpp:43-131, kseq over gzFile) and the `--search` loop of src/baseFAST.cpp:56-81 as one call.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lf_reads lf_reads_t;
typedef struct lf_read_batch lf_read_batch_t;
int  lf_reads_open(const char *path, lf_reads_t **out);
/* up to max_reads records / max_bases sequence bytes (0 = no limit); *out = NULL at end of input */
int  lf_reads_next(lf_reads_t *r, int max_reads, uint64_t max_bases, lf_read_batch_t **out);
void lf_reads_close(lf_reads_t *r);
int  lf_read_batch_size(const lf_read_batch_t *b);
const char *const *lf_read_batch_names(const lf_read_batch_t *b);
const char *const *lf_read_batch_seqs(const lf_read_batch_t *b);
const char *const *lf_read_batch_quals(const lf_read_batch_t *b);     /* "" for FASTA records (printed as "*") */
void lf_read_batch_free(lf_read_batch_t *b);
/* reads_path -> SAM at out_path (NULL or "-": stdout); the next batch is read while the current one is on the GPU */
int  lf_map_file(const lf_index_t *idx, const lf_params_t *p, const char *reads_path, const char *out_path, int no_header,
                 const char *cmdline, int batch_reads, lf_stats_t *total);

/* same, SAM text written into a caller-owned buffer (reusable); LF_ERR_NOMEM when out_cap is too small.
 * If the buffer is PINNED host memory of the HIP runtime (hipHostMalloc / hipHostRegister; lf_map_file allocates its own that
 * way), the SEQ / QUAL columns -- 60 % of the text, bytes the caller handed in -- never cross the link: a kernel stores the
 * rest of every line straight into the buffer and host threads copy SEQ / QUAL in from `seqs` / `quals` (byte-identical output). */
int  lf_map_batch_into(const lf_index_t *idx, const lf_params_t *p, int n, const char *const *names,
                       const char *const *seqs, const char *const *quals, char *out, size_t out_cap, size_t *sam_len,
                       lf_stats_t *stats);
/* same with seq_lens[i] == strlen(seqs[i]) supplied by the caller (the reference's Read records carry `length`,
 * src/Reads.h): saves the library one pass over the bases.  The strings must still be NUL-terminated. */
int  lf_map_batch_into_lens(const lf_index_t *idx, const lf_params_t *p, int n, const char *const *names,
                            const char *const *seqs, const char *const *quals, const uint32_t *seq_lens,
                            char *out, size_t out_cap, size_t *sam_len, lf_stats_t *stats);
/* ---- a mapper-ready batch (the reference's readChunk leaves one outside its mapping timer: src/Reads.cpp:84-104, src/baseFAST.cpp:59-75).
 * lf_batch_create takes the reads as the reader has them (pointers; the strings must outlive the batch) and makes, ONCE, what every mapping call of
 * the other entry points makes per call: the lengths, and the bit planes the read batch crosses the host link as (code low bit, code high bit,
 * "is upper-case ACGT": 3 / 8 of the bytes; bytes outside ACGT as an exception list) in pinned memory.  Reads shorter than min_read_len (-l) are not mapped
 * and not packed: lf_map_batch_from uses the planes when its -l equals the batch's and packs per call otherwise.  Same records as lf_map_batch.  A batch
 * can be mapped any number of times, from several threads at once; lf_batch_free releases the planes, not the caller's strings (for a batch of lf_reads_next it is
 * lf_read_batch_free).  threads <= 0: all CPUs.  The object is the reader's lf_read_batch_t. ---- */
lf_read_batch_t *lf_batch_create(int n, const char *const *names, const char *const *seqs, const char *const *quals,
                                 const uint32_t *seq_lens /* NULL: strlen */, int min_read_len, int threads);
int  lf_read_batch_prepack(lf_read_batch_t *b, int min_read_len, int threads);      /* the same for a batch of the library's reader (lf_reads_next); idempotent */
void lf_batch_free(lf_read_batch_t *b);
int  lf_batch_size(const lf_read_batch_t *b);
/* output: a caller-owned buffer as in lf_map_batch_into (pinned: the SEQ-less egress) */
int  lf_map_batch_from(const lf_index_t *idx, const lf_params_t *p, const lf_read_batch_t *b, char *out, size_t out_cap, size_t *sam_len, lf_stats_t *stats);
/* Device-resident form of lf_map_batch_into_lens: the bases (and qualities) are already in HBM of idx's device and the
 * SAM text is left there (out_is_device) -- the bulk data never crosses PCIe.  This is what one rank of the N-GPU
 * deployment calls on the shard it received over xGMI (lordfast_amd/dist.py); the reference's equivalent is the chunk
 * hand-over initFASTChunk(Read *seqList, int n) + mapSeqMT() (src/LordFAST.h:124-125), whose buffers live in host memory.
 *   read i = d_seqs[seq_off[i] .. seq_off[i] + seq_lens[i])   (anything may sit between two reads);
 *   d_quals: NULL (FASTA, QUAL printed as "*") or a device blob in the same layout; names / seq_off / seq_lens: host arrays.
 *   out: out_cap bytes of device memory (out_is_device != 0) or host memory; a device buffer gets no terminating NUL.
 *   Stream ordering: the library reads d_seqs / d_quals and writes `out` on streams of its own -- whatever produced the blobs
 *   (a copy, a collective, a kernel of the caller) must be COMPLETE before the call, and the text in `out` is complete when the
 *   call returns.
 * All lf_map_batch* calls are re-entrant: threads may map several batches at once; they share the device's eight lanes (the
 * per-lane working memory in HBM) through one allocator, so small batches overlap and large ones queue for lanes. */
int  lf_map_batch_dev(const lf_index_t *idx, const lf_params_t *p, int n, const char *const *names, const void *d_seqs,
                      const uint64_t *seq_off, const uint32_t *seq_lens, const void *d_quals, void *out, size_t out_cap,
                      int out_is_device, size_t *sam_len, lf_stats_t *stats);
/* Stage view of mapSeq (src/LordFAST.cpp:507-561) for a batch of reads: the coarse / fine decision of
 * findTopWins_coarse / _fine (:582-657, :819-904), the windows alignWin (:995-1189) is called with and its result per window
 * -- totalScore and the records' positions / alnScore / nmCount -- before the sort and printSamEntry.
 *   mode[i]: 0 shorter than -l, 1 no window, 2 coarse (one window: the best), 3 fine (the top-N heap array, in array order)
 *   windows of read i: wins[win0[i] .. win0[i + 1]); records of a window: recs[rec0 .. rec0 + n_records) */
typedef struct { uint32_t tStart, tEnd, isReverse; float score; int32_t totalScore; uint32_t n_records, rec0; } lf_stage_win_t;
typedef struct { uint32_t pos, posEnd, qStart, qEnd; int32_t flag, alnScore, nmCount; } lf_stage_rec_t;
typedef struct { int n_reads; uint8_t *mode; uint32_t *win0; uint32_t n_wins; lf_stage_win_t *wins; uint32_t n_recs; lf_stage_rec_t *recs; } lf_stages_t;
int  lf_map_stages_batch(const lf_index_t *idx, const lf_params_t *p, int n, const char *const *seqs, lf_stages_t **out);
void lf_stages_free(lf_stages_t *s);
/* Test hook, not a production switch: the host-side cross-check implementations of four device stages (bit 0 vote / selection /
 * sort, bit 1 CIGAR / MD strings, bit 2 chain walk, bit 3 SAM line assembly) can only be selected through this call; no
 * environment variable reaches them.  Process-wide; returns the previous mask.  0 = the product path. */
unsigned lf_debug_crosscheck(unsigned mask);
/* device memory for callers that do not link HIP themselves (a caller that does may pass any pointer hipMalloc gave it):
 * lf_device_copy takes host or device pointers on either side and is synchronous */
void *lf_device_alloc(int device, size_t bytes);
void  lf_device_free(int device, void *ptr);
int   lf_device_copy(int device, void *dst, const void *src, size_t bytes);
/* ------------------------------------------------------------------------------------------------
 * Several GPUs in one process: idx[d] = the same index loaded on device d (lf_index_load(prefix, d, ...)).
 * Reads are independent, so the batch is cut into chunks that the devices pull from one counter -- the reference's
 * pthread pool takes reads from a shared cursor the same way (src/LordFAST.cpp:295-303) -- and the SAM records are
 * written in input order: the output is byte-identical to the one-device output.  seq_lens may be NULL.
 * out == NULL: *sam is malloc'd (lf_free); otherwise the text goes to out[0 .. out_cap).
 * (One process per GPU with torch.distributed / RCCL is the other deployment: lordfast_amd/dist.py, bench.py.)
 * ---------------------------------------------------------------------------------------------- */
int  lf_map_batch_multi(const lf_index_t *const *idx, int n_idx, const lf_params_t *p, int n, const char *const *names,
                        const char *const *seqs, const char *const *quals, const uint32_t *seq_lens,
                        char *out, size_t out_cap, char **sam, size_t *sam_len, lf_stats_t *stats);
int  lf_map_file_multi(const lf_index_t *const *idx, int n_idx, const lf_params_t *p, const char *reads_path, const char *out_path,
                       int no_header, const char *cmdline, int batch_reads, lf_stats_t *total);
char *lf_sam_header(const lf_index_t *idx, const lf_params_t *p, const char *cmdline); /* src/BWT.cpp:668-681 */
void lf_free(void *ptr);

/* ------------------------------------------------------------------------------------------------
 * Drop-in entry points: the reference's internal C/C++ functions with C linkage.
 * They use one process-global index (like _fmd_index, src/BWT.cpp:32) and option set.
 * ---------------------------------------------------------------------------------------------- */
extern lf_params_t lf_global_params;      /* MIN_ANCHOR_LEN, SAMPLING_COUNT, ... (src/Common.h:76-81) */

int      bwt_index(char *ref_path);                                       /* src/BWT.h:28  */
int      bwt_load(char *ref_path);                                        /* src/BWT.h:29  */
uint32_t bwt_get_refGenLen(void);                                         /* src/BWT.h:31  */
void     getLocs_extend_whole_step(char *qSeq, uint32_t qLen, uint32_t hash_count,
                                   SeedList *seedForward, SeedList *seedReverse);   /* src/BWT.h:32 */
void     bwt_get_intv_info(uint64_t beg, uint64_t end, char **chr_name, int32_t *chr_len,
                           uint32_t *chr_beg, uint32_t *chr_end);         /* src/BWT.h:35 */
void     bwt_get_chr_boundaries(uint64_t beg, uint64_t end, uint32_t *chr_beg, uint32_t *chr_end); /* src/BWT.h:36 */
void     bwt_str_pac2int(uint32_t beg, uint32_t len, uint8_t *seq);       /* src/BWT.h:37 */
void     bwt_str_pac2char(uint32_t beg, uint32_t len, char *seq);         /* src/BWT.h:38 */
void     printSamHeader(FILE *fp);                                        /* src/BWT.h:39 */
/* src/Chain.h:51 -- `Chain_t &bestChain` becomes a pointer in the C ABI */
void     chain_seeds_n2(Seed_t *fragment_list, uint32_t nFragment, Chain_t *bestChain);
int      chain_seeds_clasp(Seed_t *fragment_list, uint32_t nFragment, Chain_t *bestChain);   /* src/Chain.h:51 */

/* lib/edlib/edlib.h:21-56,79-135,172,190 (same enum values and struct layouts) */
typedef enum { EDLIB_MODE_NW, EDLIB_MODE_SHW, EDLIB_MODE_HW } EdlibAlignMode;
typedef enum { EDLIB_TASK_DISTANCE, EDLIB_TASK_LOC, EDLIB_TASK_PATH } EdlibAlignTask;
typedef struct { int k; EdlibAlignMode mode; EdlibAlignTask task; } EdlibAlignConfig;
typedef struct {
    int editDistance;
    int *endLocations;
    int *startLocations;
    int numLocations;
    unsigned char *alignment;
    int alignmentLength;
    int alphabetLength;
} EdlibAlignResult;
EdlibAlignConfig edlibNewAlignConfig(int k, EdlibAlignMode mode, EdlibAlignTask task);
/* supports what lordFAST calls: k = -1, mode NW or SHW, task PATH (src/LordFAST.cpp:1833,1941,2168) */
EdlibAlignResult edlibAlign(const char *query, int queryLength, const char *target, int targetLength,
                            EdlibAlignConfig config);
void edlibFreeAlignResult(EdlibAlignResult result);

/* lib/bwa/ksw.h:108 -- m must be 5 and mat the clip matrix of src/LordFAST.cpp:178-187; end_bonus 0;
 * _gtle/_gscore/_max_off must be NULL (lordFAST passes 0 for them) */
int ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                int *qle, int *tle, int *gtle, int *gscore, int *max_off);

/* lib/bwa/ksw.h:107 -- ksw_extend = ksw_extend2 with equal insertion / deletion costs (lib/bwa/ksw.c:480-483) */
int ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape,
               int w, int end_bonus, int zdrop, int h0, int *qle, int *tle, int *gtle, int *gscore, int *max_off);

/* src/LordFAST.h:122-126 -- the chunk driver. Read layout as src/Reads.h:28-35. */
typedef struct { uint32_t *length; char *seq; char *qual; char *name; uint8_t *isFq; } Read;
void initializeFAST(void);             /* opens lf_global_output (stdout if NULL), prints the header */
void finalizeFAST(void);
void initFASTChunk(Read *seqList, int seqListSize);
void mapSeqMT(void);                   /* maps the chunk on the GPU, writes SAM records in read order */
extern FILE *lf_global_output;
extern int   lf_global_no_header;
extern char  lf_global_cmdline[2000];

#ifdef __cplusplus
}
#endif
#endif /* LORDFAST_AMD_H */
