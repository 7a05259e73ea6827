/*
 * lf_oracle.h -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Plain-C CPU restatement of lordFAST's per-read seed -> vote -> chain -> extend -> SAM path
 * (reference: vpc-ccg/lordfast v0.0.10).  Every function cites the reference file:line it follows.
 * It is the checker the GPU path is compared with; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (lordfast_amd/) never links or calls it.
 *
 * Parity status: PINNED -- checked against the real reference compiled from /root/reference
 * (oracle/_ref/liblfref.so, see oracle/Makefile + tests/test_oracle_vs_ref.py) and against the
 * committed golden vectors under tests/golden/ that were generated from that build.
 */
#ifndef LF_ORACLE_H
#define LF_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Seed_t (src/LordFAST.h:30-35): qPos is a 20-bit and len a 12-bit field; we keep u32 fields and
 * apply the same wrap (qPos & 0xFFFFF, len & 0xFFF) when a seed is stored. */
typedef struct { uint32_t tPos, qPos, len; } lfo_seed_t;

typedef struct {
    int64_t offset; int32_t len; char *name;
} lfo_contig_t;

typedef struct {
    /* .bwt (lib/bwa/bwt.c:443-462) */
    uint64_t primary, L2[5], seq_len, bwt_size;
    uint32_t *bwt;
    /* .sa (lib/bwa/bwt.c:421-441) */
    uint64_t sa_intv, n_sa, *sa;
    /* .ann / .pac (lib/bwa/bntseq.c:100-160, lib/bwa/bwa.c:270-274) */
    int64_t l_pac; int32_t n_seqs; lfo_contig_t *contigs; uint8_t *pac;
    /* .cache (src/BWT.cpp:159-187) */
    int32_t kcache; uint64_t *cache; /* pairs (beg,end) */
} lfo_index_t;

/* option globals of src/CommandLineParser.cpp:41-55 */
typedef struct {
    int min_anchor_len;   /* -k  MIN_ANCHOR_LEN 14  */
    int sampling_count;   /* -c  SAMPLING_COUNT 1000 */
    int max_map;          /* -n  MAX_MAP 10 */
    int min_read_len;     /* -l  MIN_READ_LEN 1000 */
    int max_ref_hits;     /* -m  MAX_REF_HITS 1000 */
    int chain_alg;        /* 0 dp-n2, 1 clasp */
    double chain_reward, chain_penalty, gap_penalty; /* 9.3, 11.4, 0.15 */
    int threads;
    char read_group_id[256];
    char read_group[1000];   /* the @RG header line (readGroup, src/CommandLineParser.cpp:42) */
} lfo_params_t;

void lfo_params_default(lfo_params_t *p);

lfo_index_t *lfo_index_load(const char *prefix);
void lfo_index_free(lfo_index_t *idx);

/* FM-index primitives (lib/bwa/bwt.c:86-163, src/BWT.cpp:265-298) */
uint64_t lfo_occ(const lfo_index_t *idx, uint64_t k, int c);
uint64_t lfo_sa(const lfo_index_t *idx, uint64_t k, uint32_t *steps);
int64_t  lfo_count_exact_cached(const lfo_index_t *idx, const char *str, int len, int avail,
                                uint64_t *sp, uint64_t *ep);

/* getLocs_extend_whole_step (src/BWT.cpp:312-394). F/R need sampling_count*max_ref_hits capacity.
 * stats (optional, 4 u64): N_cache, N_occblk, N_sa, N_readbytes of SURVEY 8(d). */
void lfo_seed(const lfo_index_t *idx, const lfo_params_t *p, const char *seq, uint32_t len,
              lfo_seed_t *F, uint32_t *nF, lfo_seed_t *R, uint32_t *nR, uint64_t *stats);

/* chain_seeds_n2 (src/Chain.cpp:232-310): reorders `seeds`, writes chain (capacity n). */
void lfo_chain_n2(const lfo_params_t *p, lfo_seed_t *seeds, uint32_t n,
                  lfo_seed_t *chain, uint32_t *chainLen, float *score);

/* chain_seeds_clasp (src/Chain.cpp:39-209; lib/clasp SOP chaining, lambda 0.15): `seeds` untouched, chain in
 * target order (capacity n). n == 0: score -1, chainLen 0 (the reference leaves chainLen stale). */
void lfo_chain_clasp(const lfo_seed_t *seeds, uint32_t n, lfo_seed_t *chain, uint32_t *chainLen, float *score);

/* edlibAlign(q,t,{k=-1,mode,PATH}) as a pure function (lib/edlib/edlib.cpp:101-221).
 * mode 0 = NW, 1 = SHW. ops: capacity n+m. Returns edit distance; *endLoc may be -1 in SHW. */
int lfo_edlib(const char *q, int n, const char *t, int m, int mode, int *endLoc, uint8_t *ops, int *nops);

/* ksw_extend2 (lib/bwa/ksw.c:380-478) with lordFAST's clip matrix 2/-16/N=0 (src/LordFAST.cpp:178-187) */
int lfo_ksw_extend2(int qlen, const uint8_t *q, int tlen, const uint8_t *t, int o_del, int e_del,
                    int o_ins, int e_ins, int w, int zdrop, int h0, int *qle, int *tle);

/* reference fetch (src/BWT.cpp:593-666) */
void lfo_pac2char(const lfo_index_t *idx, uint32_t beg, uint32_t len, char *out);
void lfo_chr_boundaries(const lfo_index_t *idx, uint64_t beg, uint64_t end, uint32_t *cb, uint32_t *ce);

/* libstdc++ std::sort / heap restatements (bits/stl_algo.h:1855-1960, bits/stl_heap.h) -- exposed for fuzzing */
void lfo_sort_seeds_by_qpos(lfo_seed_t *a, size_t n);

/* SAM header (src/BWT.cpp:668-681); returns malloc'd string */
char *lfo_sam_header(const lfo_index_t *idx, const lfo_params_t *p, const char *cmdline);

/* mapSeq over a batch (src/LordFAST.cpp:461-580): returns malloc'd SAM records, in read order.
 * quals[i] may be NULL/"" (FASTA input -> QUAL "*"). */
char *lfo_map_batch(const lfo_index_t *idx, const lfo_params_t *p, int n,
                    const char **names, const char **seqs, const char **quals, size_t *out_len);

void lfo_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
