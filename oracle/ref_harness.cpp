/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * A thin C-linkage harness around the *real* lordFAST reference, which oracle/Makefile compiles
 * from the sources where they lie under /root/reference (nothing is copied into this repo).
 * The result, oracle/_ref/liblfref.so, is used
 *   - to pin oracle/lf_oracle.c (our restatement) at stage and SAM level,
 *   - to generate the committed golden vectors (tests/golden/make_golden.py),
 *   - as bench.py's cpu_baseline (kind "reference").
 *
 * The reference's CLI (src/CommandLineParser.cpp) and main (src/baseFAST.cpp) are not compiled;
 * this file defines the option globals the CLI would define (same names, same defaults,
 * src/CommandLineParser.cpp:32-55) and drives the same calls main() makes
 * (src/baseFAST.cpp:32-84).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <string>
#include <vector>

#include "Common.h"
#include "Reads.h"
#include "LordFAST.h"
#include "BWT.h"
#include "Chain.h"
#include "edlib.h"
extern "C" {
#include "ksw.h"
}

/* ---- option globals normally owned by src/CommandLineParser.cpp:32-55 ---- */
int         indexingMode = 0;
int         searchingMode = 1;
int         noSamHeader = 0;
char       *seqFile = NULL;
char       *refFile = NULL;
char        outputMap[1000];
char        opt_commandAll[2000];
int         opt_outputBufferSize = 2000000;
chainAlg_t  chainAlg = CHAIN_ALG_DPN2;
char        readGroup[1000];
char        readGroupId[1000];
double      chainReward = 9.3;
double      chainPenalty = 11.4;
double      gapPenalty = 0.15;
long long   memUsage = 0;
int         THREAD_COUNT = 1;
int         THREAD_ID[255];
int         MIN_ANCHOR_LEN = 14;
int         SAMPLING_COUNT = 1000;
int         MAX_MAP = 10;
int         MIN_READ_LEN = 1000;
int         MAX_REF_HITS = 1000;

/* internal (non-static) reference symbols used for stage-level checks */
extern int8_t _pf_kswMatrix_clip[25];

extern WinCount_t **_pf_refWin_cnt;
extern uint32_t _pf_refWin_num;
/* initializeFAST takes the vote array from malloc (getMem, src/Common.cpp:68-75) and never clears it: votes are told apart by
 * a read-number tag (src/LordFAST.cpp:596-610).  A fresh process gets zeroed pages; a process that runs several sessions
 * (this harness, the tests) can get a previous session's array back, tags and counts included.  Every session of the
 * harness therefore starts from what a fresh process sees. */
static void fresh_vote_array(void)
{
    for (int i = 0; i < THREAD_COUNT; i++) memset(_pf_refWin_cnt[i], 0, (size_t)_pf_refWin_num * sizeof(WinCount_t));
}

static double now_s() { return getRealTime(); }

extern "C" {

int ref_index_build(const char *fasta) { return bwt_index((char *)fasta); }

int ref_load(const char *fasta) { return bwt_load((char *)fasta); }

uint32_t ref_genome_len(void) { return bwt_get_refGenLen(); }

void ref_set_params(int k, int c, int n, int l, int m, int chain_alg_clasp,
                    double reward, double penalty, double gap, int threads)
{
    MIN_ANCHOR_LEN = k; SAMPLING_COUNT = c; MAX_MAP = n; MIN_READ_LEN = l; MAX_REF_HITS = m;
    chainAlg = chain_alg_clasp ? CHAIN_ALG_CLASP : CHAIN_ALG_DPN2;
    chainReward = reward; chainPenalty = penalty; gapPenalty = gap;
    if (threads <= 0 || threads > sysconf(_SC_NPROCESSORS_ONLN)) threads = sysconf(_SC_NPROCESSORS_ONLN);
    if (threads > 255) threads = 255;
    THREAD_COUNT = threads;
    for (int i = 0; i < 255; i++) THREAD_ID[i] = i;
}

int ref_threads(void) { return THREAD_COUNT; }

void ref_set_cmdline(const char *s) { strncpy(opt_commandAll, s, sizeof(opt_commandAll) - 1); }

/* --search ref --seq reads [-o out]: the loop of src/baseFAST.cpp:44-81. Returns mapSeqMT seconds. */
double ref_map_file(const char *reads_path, const char *out_path, int no_header)
{
    Read *seqList; unsigned int seqListSize; double t = 0;
    noSamHeader = no_header;
    strncpy(outputMap, out_path, sizeof(outputMap) - 1);
    if (!initRead((char *)reads_path, 100000000)) return -1;
    initializeFAST();
    fresh_vote_array();
    while (readChunk(&seqList, &seqListSize) > 0) {
        initFASTChunk(seqList, seqListSize);
        double t0 = now_s();
        mapSeqMT();
        t += now_s() - t0;
        releaseChunk();
    }
    finalizeFAST();
    finalizeReads();
    return t;
}

/* Same, on an in-memory batch (one chunk). Read blocks laid out as src/Reads.cpp:84-104. */
double ref_map_mem(int n, const char **names, const char **seqs, const char **quals,
                   const char *out_path, int no_header)
{
    std::vector<Read> reads(n);
    std::vector<char *> blocks(n);
    for (int i = 0; i < n; i++) {
        int readLen = strlen(seqs[i]);
        int isFq = (quals && quals[i] && quals[i][0]);
        int qualLen = isFq ? (int)strlen(quals[i]) : 1;
        int nameLen = strlen(names[i]);
        size_t size = sizeof(uint32_t) + (readLen + 1) + (qualLen + 1) + (nameLen + 1) + sizeof(uint8_t);
        char *b = (char *)malloc(size);
        blocks[i] = b;
        reads[i].length = (uint32_t *)b;
        reads[i].seq = (char *)(reads[i].length + 1);
        reads[i].qual = reads[i].seq + readLen + 1;
        reads[i].name = reads[i].qual + qualLen + 1;
        reads[i].isFq = (uint8_t *)(reads[i].name + nameLen + 1);
        *reads[i].length = readLen;
        strcpy(reads[i].seq, seqs[i]);
        strcpy(reads[i].qual, isFq ? quals[i] : "*");
        strcpy(reads[i].name, names[i]);
        *reads[i].isFq = isFq;
    }
    noSamHeader = no_header;
    strncpy(outputMap, out_path, sizeof(outputMap) - 1);
    initializeFAST();
    fresh_vote_array();
    initFASTChunk(reads.data(), n);
    double t0 = now_s();
    mapSeqMT();
    double t = now_s() - t0;
    finalizeFAST();
    for (int i = 0; i < n; i++) free(blocks[i]);
    return t;
}

/* ---- stage level ---- */

/* getLocs_extend_whole_step (src/BWT.cpp:312). Seeds returned as (tPos,qPos,len) u32 triples. */
void ref_seed(const char *seq, uint32_t len, uint32_t hash_count,
              uint32_t *F, uint32_t *nF, uint32_t *R, uint32_t *nR)
{
    SeedList f, r;
    size_t cap = (size_t)hash_count * MAX_REF_HITS;
    f.list = (Seed_t *)malloc(cap * sizeof(Seed_t)); r.list = (Seed_t *)malloc(cap * sizeof(Seed_t));
    f.num = r.num = 0;
    getLocs_extend_whole_step((char *)seq, len, hash_count, &f, &r);
    for (uint32_t i = 0; i < f.num; i++) { F[3*i] = f.list[i].tPos; F[3*i+1] = f.list[i].qPos; F[3*i+2] = f.list[i].len; }
    for (uint32_t i = 0; i < r.num; i++) { R[3*i] = r.list[i].tPos; R[3*i+1] = r.list[i].qPos; R[3*i+2] = r.list[i].len; }
    *nF = f.num; *nR = r.num;
    free(f.list); free(r.list);
}

/* chain_seeds_n2 (src/Chain.cpp:232). io: n triples in (reordered like the reference does), chain triples out. */
void ref_chain_n2(uint32_t *seeds, uint32_t n, uint32_t *chain, uint32_t *chainLen, float *score)
{
    std::vector<Seed_t> in(n ? n : 1), out(n ? n : 1);
    for (uint32_t i = 0; i < n; i++) { in[i].tPos = seeds[3*i]; in[i].qPos = seeds[3*i+1]; in[i].len = seeds[3*i+2]; }
    Chain_t c; c.seeds = out.data(); c.chainLen = 0; c.score = 0;
    chain_seeds_n2(in.data(), n, c);
    for (uint32_t i = 0; i < n; i++) { seeds[3*i] = in[i].tPos; seeds[3*i+1] = in[i].qPos; seeds[3*i+2] = in[i].len; }
    for (uint32_t i = 0; i < c.chainLen; i++) { chain[3*i] = out[i].tPos; chain[3*i+1] = out[i].qPos; chain[3*i+2] = out[i].len; }
    *chainLen = c.chainLen; *score = c.score;
}

int ref_chain_clasp(uint32_t *seeds, uint32_t n, uint32_t *chain, uint32_t *chainLen, float *score)
{
    std::vector<Seed_t> in(n ? n : 1), out(n ? n : 1);
    for (uint32_t i = 0; i < n; i++) { in[i].tPos = seeds[3*i]; in[i].qPos = seeds[3*i+1]; in[i].len = seeds[3*i+2]; }
    Chain_t c; c.seeds = out.data(); c.chainLen = 0; c.score = 0;
    int rc = chain_seeds_clasp(in.data(), n, c);
    for (uint32_t i = 0; i < c.chainLen; i++) { chain[3*i] = out[i].tPos; chain[3*i+1] = out[i].qPos; chain[3*i+2] = out[i].len; }
    *chainLen = c.chainLen; *score = c.score;
    return rc;
}

/* edlibAlign(q,t,{k=-1,mode,PATH}) (lib/edlib/edlib.cpp:101). mode: 0 = NW, 1 = SHW. ops must hold n+m bytes. */
int ref_edlib(const char *q, int n, const char *t, int m, int mode, int *endLoc, uint8_t *ops, int *nops)
{
    EdlibAlignResult r = edlibAlign(q, n, t, m, edlibNewAlignConfig(-1, mode ? EDLIB_MODE_SHW : EDLIB_MODE_NW, EDLIB_TASK_PATH));
    int ed = r.editDistance;
    *endLoc = (r.endLocations ? r.endLocations[0] : -2);
    *nops = r.alignmentLength;
    if (r.alignment) memcpy(ops, r.alignment, r.alignmentLength);
    edlibFreeAlignResult(r);
    return ed;
}

/* ksw_extend2 with the 5x5 clip matrix lordFAST uses (src/LordFAST.cpp:178-187). */
int ref_ksw_extend2(int qlen, const uint8_t *q, int tlen, const uint8_t *t, int o_del, int e_del,
                    int o_ins, int e_ins, int w, int zdrop, int h0, int *qle, int *tle)
{
    int8_t mat[25]; int k = 0;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[k++] = (i == j ? 2 : -16); mat[k++] = 0; }
    for (int j = 0; j < 5; ++j) mat[k++] = 0;
    return ksw_extend2(qlen, q, tlen, t, 5, mat, o_del, e_del, o_ins, e_ins, w, 0, zdrop, h0, qle, tle, 0, 0, 0);
}

/* ---- the window stage and alignWin as mapSeq drives them (src/LordFAST.cpp:507-561): the reference's own non-static
 * functions on its own per-thread globals (thread 0).  ref_stage_begin / _end bracket a session (initializeFAST allocates
 * the per-thread arrays; its output goes to /dev/null). ---- */
} /* extern "C" */
extern SeedList *_pf_seedsForward, *_pf_seedsReverse;
extern WinList *_pf_topWins;
extern MapInfo *_pf_topMappings;
extern int _pf_seqListSize;
void findTopWins_coarse(uint32_t read_len, SeedList *seeds, int isRev, int readIdx, int id);
void findTopWins_fine(uint32_t read_len, SeedList *seeds, int isRev, int readIdx, float minScore, int id);
void alignWin(Win_t &win, char *query, char *query_rev, uint32_t rLen, char *qual, char *qual_rev, SamList_t &map, int id);
bool compareWin(const Win_t &w1, const Win_t &w2);
#include <algorithm>
extern "C" {

static int g_stage_t = 0;
void ref_stage_begin(void)
{
    noSamHeader = 1; strcpy(outputMap, "/dev/null");
    THREAD_COUNT = 1;
    initializeFAST();
    fresh_vote_array();
    _pf_seqListSize = 1 << 20;            /* the tag of fine-mode votes is t + _pf_seqListSize + 1 (src/LordFAST.cpp:554) */
    /* g_stage_t keeps counting across sessions: the vote array is tagged with the read number (:596-610) and a later
     * initializeFAST may get the previous session's memory back un-zeroed -- a repeated tag would add to stale counts */
}
void ref_stage_end(void) { finalizeFAST(); }

static void put_win(uint32_t *o, const Win_t &w) { o[0] = w.tStart; o[1] = w.tEnd; o[2] = w.isReverse; memcpy(&o[3], &w.score, 4); }

/* one read.  mode: 1 no window, 2 coarse, 3 fine.  coarse[4 * k]: the windows after std::sort_heap (:528); wins[4 * k]: the
 * windows alignWin is called with (coarse: the first one; fine: the heap array of :553-562, in array order); maps: per such
 * window { totalScore, n_records, then per record pos, posEnd, qStart, qEnd, flag, alnScore, nmCount }.  Returns words in maps. */
int ref_stage_windows(const char *seq, uint32_t len, int *mode, uint32_t *coarse, int *n_coarse, uint32_t *wins, int *n_wins, int32_t *maps, int maps_cap)
{
    const int id = 0, t = g_stage_t++;
    *mode = 0; *n_coarse = 0; *n_wins = 0;
    if ((int)len < MIN_READ_LEN) return 0;
    std::vector<char> fwd(seq, seq + len + 1), rev(len + 1), qrev(2);
    fwd[len] = 0;
    reverseComplement(fwd.data(), rev.data(), len);
    char qual[2] = "*"; qrev[0] = '*'; qrev[1] = 0;
    getLocs_extend_whole_step(fwd.data(), len, SAMPLING_COUNT, _pf_seedsForward + id, _pf_seedsReverse + id);
    _pf_topWins[id].num = 0;
    findTopWins_coarse(len, _pf_seedsForward + id, 0, t + 1, id);
    findTopWins_coarse(len, _pf_seedsReverse + id, 1, -(t + 1), id);
    if (_pf_topWins[id].num == 0) { *mode = 1; return 0; }
    std::sort_heap(_pf_topWins[id].list, _pf_topWins[id].list + _pf_topWins[id].num, compareWin);
    *n_coarse = (int)_pf_topWins[id].num;
    for (uint32_t k = 0; k < _pf_topWins[id].num; k++) put_win(coarse + 4 * k, _pf_topWins[id].list[k]);
    const float scoreRatio = 4;
    int nw = 0, words = 0;
    auto dump = [&](SamList_t &m) {
        if (words + 2 + 7 * (int)m.samList.size() > maps_cap) return;
        maps[words++] = m.totalScore; maps[words++] = (int)m.samList.size();
        for (const Sam_t &r : m.samList) { maps[words++] = (int32_t)r.pos; maps[words++] = (int32_t)r.posEnd; maps[words++] = (int32_t)r.qStart; maps[words++] = (int32_t)r.qEnd; maps[words++] = r.flag; maps[words++] = r.alnScore; maps[words++] = r.nmCount; }
    };
    /* a single candidate is compared with a stale slot in the reference (list[1] of an earlier read); both branches align the
     * same window then, so the harness reports the coarse branch for num == 1 */
    if (_pf_topWins[id].num == 1 || _pf_topWins[id].list[0].score >= scoreRatio * _pf_topWins[id].list[1].score) {
        *mode = 2;
        _pf_topMappings[id].mappings[0].samList.clear();
        alignWin(_pf_topWins[id].list[0], fwd.data(), rev.data(), len, qual, qrev.data(), _pf_topMappings[id].mappings[0], id);
        put_win(wins, _pf_topWins[id].list[0]); nw = 1;
        dump(_pf_topMappings[id].mappings[0]);
    } else {
        *mode = 3;
        _pf_topWins[id].num = 0;
        const float tmp_minScore = (float)_pf_topWins[id].list[0].score / scoreRatio;
        findTopWins_fine(len, _pf_seedsForward + id, 0, t + _pf_seqListSize + 1, tmp_minScore, id);
        findTopWins_fine(len, _pf_seedsReverse + id, 1, -(t + _pf_seqListSize + 1), tmp_minScore, id);
        for (uint32_t i = 0; i < _pf_topWins[id].num; i++) {
            _pf_topMappings[id].mappings[i].samList.clear();
            alignWin(_pf_topWins[id].list[i], fwd.data(), rev.data(), len, qual, qrev.data(), _pf_topMappings[id].mappings[i], id);
            put_win(wins + 4 * i, _pf_topWins[id].list[i]);
            dump(_pf_topMappings[id].mappings[i]);
        }
        nw = (int)_pf_topWins[id].num;
    }
    *n_wins = nw;
    return words;
}

void ref_pac2char(uint32_t beg, uint32_t len, char *out) { bwt_str_pac2char(beg, len, out); }

void ref_chr_boundaries(uint64_t beg, uint64_t end, uint32_t *cb, uint32_t *ce) { bwt_get_chr_boundaries(beg, end, cb, ce); }

} /* extern "C" */
