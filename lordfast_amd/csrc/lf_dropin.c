/*
 * lf_dropin.c -- the reference's own internal entry points, with C linkage, over the GPU batch API.
 *
 * lordFAST has no plugin interface: its driver (src/baseFAST.cpp:32-84) calls bwt_load / initializeFAST /
 * initFASTChunk / mapSeqMT / finalizeFAST, and its mapper calls getLocs_extend_whole_step, chain_seeds_n2,
 * edlibAlign, ksw_extend2 and the bwt_* fetch helpers on process-global state.  These wrappers keep those
 * names, argument meanings and error behaviour ("[ERROR] ..." on stderr + exit(EXIT_FAILURE), like
 * src/LordFAST.cpp:199-200) so a maintainer can link liblfgpu.so in place of BWT.o / Chain.o / edlib.o /
 * ksw.o.  One item per call means one tiny GPU launch per call: correct, but the
 * batch API (lf_map_batch) is what performance comes from.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lf_internal.h"

lf_params_t lf_global_params = { 14, 1000, 10, 1000, 1000, 0, 9.3, 11.4, 0.15, 0, "" };   /* src/CommandLineParser.cpp:41-55 */
FILE *lf_global_output = NULL;
int   lf_global_no_header = 0;
char  lf_global_cmdline[2000] = "";

static lf_index_t *g_ix = NULL;
static Read *g_chunk = NULL;
static int g_chunk_n = 0;

static void die(const char *where)
{
    fprintf(stderr, "[ERROR] (%s) %s\n", where, lf_last_error());
    exit(EXIT_FAILURE);
}
static void need_index(const char *where)
{
    if (!g_ix) { lf_set_error("no index loaded: call bwt_load() first"); die(where); }
}

int bwt_index(char *ref_path)                                   /* src/BWT.cpp:140-157 */
{
    fprintf(stderr, "[NOTE] (bwt_index) building the index on the GPU...\n");
    if (lf_index_build(ref_path, 0) != LF_OK) { fprintf(stderr, "[ERROR] (bwt_index) %s\n", lf_last_error()); return 1; }
    return 0;
}

int bwt_load(char *ref_path)                                    /* src/BWT.cpp:189-242: builds the index when .bwt is missing */
{
    char path[4096];
    snprintf(path, sizeof path, "%s.bwt", ref_path);
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        fprintf(stderr, "[WARNING] (bwt_load) could not locate index file: %s\n", path);
        if (bwt_index(ref_path)) return 1;
    } else fclose(fp);
    if (g_ix) { lf_index_free(g_ix); g_ix = NULL; }
    if (lf_index_load(ref_path, 0, LF_IDX_FULL_SA, &g_ix) != LF_OK) { fprintf(stderr, "[ERROR] (bwt_load) %s\n", lf_last_error()); return 1; }
    return 0;
}

uint32_t bwt_get_refGenLen(void) { need_index("bwt_get_refGenLen"); return lf_index_genome_len(g_ix); }

void getLocs_extend_whole_step(char *qSeq, uint32_t qLen, uint32_t hash_count, SeedList *seedForward, SeedList *seedReverse)
{
    need_index("getLocs_extend_whole_step");
    lf_params_t p = lf_global_params;
    p.sampling_count = (int)hash_count;
    uint64_t off[2] = { 0, qLen };
    lf_seeds_t *s = NULL;
    if (lf_seed_batch(g_ix, &p, 1, qSeq, off, &s) != LF_OK) die("getLocs_extend_whole_step");
    memcpy(seedForward->list, s->F, s->offF[1] * sizeof(Seed_t)); seedForward->num = (uint32_t)s->offF[1];
    memcpy(seedReverse->list, s->R, s->offR[1] * sizeof(Seed_t)); seedReverse->num = (uint32_t)s->offR[1];
    lf_seeds_free(s);
}

static int rid_of(uint64_t pos)
{   /* bns_pos2rid (lib/bwa/bntseq.c:349-363); pos >= l_pac: the reference reads anns[-1], we clamp */
    const struct lf_index *ix = g_ix;
    if ((int64_t)pos >= ix->l_pac) return ix->n_seqs - 1;
    int lo = 0, hi = ix->n_seqs - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (ix->contigs[mid].offset <= (int64_t)pos) lo = mid; else hi = mid - 1; }
    return lo;
}

void bwt_get_intv_info(uint64_t beg, uint64_t end, char **chr_name, int32_t *chr_len, uint32_t *chr_beg, uint32_t *chr_end)
{   /* src/BWT.cpp:636-651 */
    need_index("bwt_get_intv_info");
    const int rid = rid_of((beg + end) >> 1);
    *chr_beg = (uint32_t)(beg - (uint64_t)g_ix->contigs[rid].offset);
    *chr_end = (uint32_t)(end - (uint64_t)g_ix->contigs[rid].offset);
    *chr_name = g_ix->contigs[rid].name;
    *chr_len = g_ix->contigs[rid].len;
}

void bwt_get_chr_boundaries(uint64_t beg, uint64_t end, uint32_t *chr_beg, uint32_t *chr_end)
{   /* src/BWT.cpp:653-666 */
    need_index("bwt_get_chr_boundaries");
    const int rid = rid_of((beg + end) >> 1);
    *chr_beg = (uint32_t)g_ix->contigs[rid].offset;
    *chr_end = (uint32_t)(g_ix->contigs[rid].offset + g_ix->contigs[rid].len - 1);
}

void bwt_str_pac2int(uint32_t beg, uint32_t len, uint8_t *seq)
{   /* src/BWT.cpp:593-599 */
    need_index("bwt_str_pac2int");
    for (uint32_t i = 0; i < len; i++) { uint32_t l = beg + i; seq[i] = (g_ix->pac[l >> 2] >> ((~l & 3) << 1)) & 3; }
}

void bwt_str_pac2char(uint32_t beg, uint32_t len, char *seq)
{   /* src/BWT.cpp:601-607 */
    need_index("bwt_str_pac2char");
    for (uint32_t i = 0; i < len; i++) { uint32_t l = beg + i; seq[i] = "ACGT"[(g_ix->pac[l >> 2] >> ((~l & 3) << 1)) & 3]; }
}

void printSamHeader(FILE *fp)
{   /* src/BWT.cpp:668-681 */
    need_index("printSamHeader");
    char *h = lf_sam_header(g_ix, &lf_global_params, lf_global_cmdline);
    fputs(h, fp);
    free(h);
}

void chain_seeds_n2(Seed_t *fragment_list, uint32_t nFragment, Chain_t *bestChain)
{   /* src/Chain.cpp:232-310: reorders fragment_list, fills bestChain->seeds (caller-allocated) */
    uint64_t off[2] = { 0, nFragment };
    uint32_t *idx = (uint32_t *)malloc(((size_t)nFragment + 1) * 4), len = 0;
    float score = 0;
    if (nFragment == 0) { bestChain->chainLen = 0; bestChain->score = -1; free(idx); return; }
    if (lf_chain_n2_batch(&lf_global_params, 1, fragment_list, off, idx, &len, &score, g_ix ? g_ix->device : 0) != LF_OK) die("chain_seeds_n2");
    for (uint32_t i = 0; i < len; i++) bestChain->seeds[i] = fragment_list[idx[i]];
    bestChain->chainLen = len; bestChain->score = score;
    free(idx);
}

int chain_seeds_clasp(Seed_t *fragment_list, uint32_t nFragment, Chain_t *bestChain)
{   /* src/Chain.cpp:39-209: fragment_list is left untouched, bestChain->seeds (caller-allocated) gets the chain in
     * target order, returns 1.  nFragment == 0: score -1; the reference leaves chainLen stale, we set it to 0. */
    uint64_t off[2] = { 0, nFragment };
    uint32_t len = 0; float score = -1;
    if (nFragment == 0) { bestChain->chainLen = 0; bestChain->score = -1; return 1; }
    Seed_t *out = (Seed_t *)malloc((size_t)nFragment * sizeof(Seed_t));
    if (lf_chain_clasp_batch(1, fragment_list, off, out, &len, &score, g_ix ? g_ix->device : 0) != LF_OK) die("chain_seeds_clasp");
    memcpy(bestChain->seeds, out, (size_t)len * sizeof(Seed_t));
    bestChain->chainLen = len; bestChain->score = score;
    free(out);
    return 1;
}

EdlibAlignConfig edlibNewAlignConfig(int k, EdlibAlignMode mode, EdlibAlignTask task)
{
    EdlibAlignConfig c; c.k = k; c.mode = mode; c.task = task; return c;
}

EdlibAlignResult edlibAlign(const char *query, int queryLength, const char *target, int targetLength, EdlibAlignConfig config)
{   /* lib/edlib/edlib.cpp:101-221 */
    EdlibAlignResult r; memset(&r, 0, sizeof r);
    r.editDistance = -1;
    if (config.mode == EDLIB_MODE_HW) { lf_set_error("EDLIB_MODE_HW is not on lordFAST's path and is not implemented"); die("edlibAlign"); }
    uint8_t seen[256]; memset(seen, 0, sizeof seen);
    for (int i = 0; i < queryLength; i++) seen[(unsigned char)query[i]] = 1;
    for (int i = 0; i < targetLength; i++) seen[(unsigned char)target[i]] = 1;
    for (int i = 0; i < 256; i++) r.alphabetLength += seen[i];
    uint64_t qoff[2] = { 0, (uint64_t)queryLength }, toff[2] = { 0, (uint64_t)targetLength };
    uint8_t mode = config.mode == EDLIB_MODE_SHW;
    int32_t ed = 0, end = 0; uint32_t nops = 0;
    uint8_t *ops = (uint8_t *)malloc((size_t)queryLength + (size_t)targetLength + 1);
    if (lf_edlib_batch(1, query, qoff, target, toff, &mode, &ed, &end, ops, &nops, g_ix ? g_ix->device : 0, NULL) != LF_OK) die("edlibAlign");
    if (config.k >= 0 && ed > config.k) { free(ops); return r; }          /* no solution within k */
    r.editDistance = ed;
    r.numLocations = 1;
    r.endLocations = (int *)malloc(sizeof(int)); r.endLocations[0] = end;
    if (config.task != EDLIB_TASK_DISTANCE) { r.startLocations = (int *)malloc(sizeof(int)); r.startLocations[0] = 0; }
    if (config.task == EDLIB_TASK_PATH) { r.alignment = (unsigned char *)realloc(ops, nops ? nops : 1); r.alignmentLength = (int)nops; }
    else free(ops);
    return r;
}

void edlibFreeAlignResult(EdlibAlignResult result)
{
    free(result.endLocations); free(result.startLocations); free(result.alignment);
}

int ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                int *qle, int *tle, int *gtle, int *gscore, int *max_off)
{   /* lib/bwa/ksw.c:380-478, with the only matrix lordFAST passes (src/LordFAST.cpp:178-187) */
    int ok = (m == 5 && end_bonus == 0 && !gtle && !gscore && !max_off);
    for (int i = 0; ok && i < 4; i++) for (int j = 0; j < 5; j++) ok = ok && mat[i * 5 + j] == (j == 4 ? 0 : (i == j ? 2 : -16));
    for (int j = 0; ok && j < 5; j++) ok = ok && mat[20 + j] == 0;
    if (!ok) { lf_set_error("only lordFAST's clip matrix (m=5, +2/-16/N 0, end_bonus 0, no gtle/gscore/max_off) is implemented"); die("ksw_extend2"); }
    uint64_t qoff[2] = { 0, (uint64_t)qlen }, toff[2] = { 0, (uint64_t)tlen };
    int32_t prm[7] = { o_del, e_del, o_ins, e_ins, w, zdrop, h0 }, sc = 0, ql = 0, tl = 0;
    if (lf_ksw_extend2_batch(1, query, qoff, target, toff, prm, &sc, &ql, &tl, g_ix ? g_ix->device : 0) != LF_OK) die("ksw_extend2");
    if (qle) *qle = ql;
    if (tle) *tle = tl;
    return sc;
}

int ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape,
               int w, int end_bonus, int zdrop, int h0, int *qle, int *tle, int *gtle, int *gscore, int *max_off)
{   /* lib/bwa/ksw.c:480-483 */
    return ksw_extend2(qlen, query, tlen, target, m, mat, gapo, gape, gapo, gape, w, end_bonus, zdrop, h0, qle, tle, gtle, gscore, max_off);
}

/* ---- the chunk driver of src/LordFAST.h:122-126 ---- */
void initializeFAST(void)
{   /* src/LordFAST.cpp:110-215: output + header */
    need_index("initializeFAST");
    if (!lf_global_output) lf_global_output = stdout;
    if (!lf_global_no_header) printSamHeader(lf_global_output);
}
void finalizeFAST(void) { if (lf_global_output && lf_global_output != stdout) fflush(lf_global_output); }
void initFASTChunk(Read *seqList, int seqListSize) { g_chunk = seqList; g_chunk_n = seqListSize; }

void mapSeqMT(void)
{   /* src/LordFAST.cpp:305-316: here one GPU batch instead of THREAD_COUNT pthreads; records come out in read order */
    need_index("mapSeqMT");
    const char **names = (const char **)malloc((size_t)g_chunk_n * sizeof(char *));
    const char **seqs = (const char **)malloc((size_t)g_chunk_n * sizeof(char *));
    const char **quals = (const char **)malloc((size_t)g_chunk_n * sizeof(char *));
    for (int i = 0; i < g_chunk_n; i++) { names[i] = g_chunk[i].name; seqs[i] = g_chunk[i].seq; quals[i] = *g_chunk[i].isFq ? g_chunk[i].qual : ""; }
    char *sam = NULL; size_t len = 0;
    if (lf_map_batch(g_ix, &lf_global_params, g_chunk_n, names, seqs, quals, &sam, &len, NULL) != LF_OK) die("mapSeqMT");
    fwrite(sam, 1, len, lf_global_output ? lf_global_output : stdout);
    free(sam); free(names); free(seqs); free(quals);
}
