/*
 * lf_shim.cpp -- the ONE file a lordFAST maintainer adds on the reference side to run the hot path on an MI355X.
 *
 * The reference's driver (src/baseFAST.cpp:32-84) is C++ and calls its mapper through C++-linkage functions on
 * process-global state:  bwt_index / bwt_load (src/BWT.h:28-29), initializeFAST / initFASTChunk / mapSeqMT /
 * finalizeFAST (src/LordFAST.h:122-126).  liblfgpu.so exports the same functions with C linkage
 * (include/lordfast_amd.h, "Drop-in entry points").  This shim is compiled with the reference's own headers and
 * flags, defines the C++-linkage functions the driver objects reference, forwards the option globals of
 * src/Common.h:58-81 into the library before the index is touched, and hands every call to the library.  With it,
 *
 *     g++ baseFAST.o Common.o CommandLineParser.o Reads.o HELP.o lf_shim.o -llfgpu -lz -lpthread -o lordfast
 *
 * replaces BWT.o LordFAST.o Chain.o edlib.o lib/bwa/*.o lib/clasp/*.o (Makefile:26-41) -- no call site changes.
 * oracle/Makefile target `shim` performs exactly this link in the build container (tests/test_shim.py).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "Common.h"            /* MIN_ANCHOR_LEN, SAMPLING_COUNT, MAX_MAP, ... opt_commandAll, outputMap, noSamHeader */
#include "Reads.h"             /* Read (src/Reads.h:28-35) */

/* the library's C ABI lives in its own namespace here: the driver's C++ names stay free for the forwarders below */
namespace lfgpu {
extern "C" {
#include "lordfast_amd.h"
}
}

static void push_options()
{
    lfgpu::lf_params_t &p = lfgpu::lf_global_params;
    p.min_anchor_len = MIN_ANCHOR_LEN;   p.sampling_count = SAMPLING_COUNT;
    p.max_map        = MAX_MAP;          p.min_read_len   = MIN_READ_LEN;
    p.max_ref_hits   = MAX_REF_HITS;     p.chain_alg      = (chainAlg == CHAIN_ALG_CLASP);
    p.chain_reward   = chainReward;      p.chain_penalty  = chainPenalty;
    p.gap_penalty    = gapPenalty;       p.threads        = THREAD_COUNT;
    strncpy(p.read_group_id, readGroupId, sizeof p.read_group_id - 1);
    strncpy(p.read_group, readGroup, sizeof p.read_group - 1);
    strncpy(lfgpu::lf_global_cmdline, opt_commandAll, sizeof lfgpu::lf_global_cmdline - 1);
    lfgpu::lf_global_no_header = noSamHeader;
    if (outputMap[0] && !lfgpu::lf_global_output) {
        lfgpu::lf_global_output = fopen(outputMap, "w");
        if (!lfgpu::lf_global_output) { fprintf(stderr, "[ERROR] (lf_shim) could not open %s for writing\n", outputMap); lfgpu::lf_global_output = stdout; }
    }
}

/* src/baseFAST.cpp:40,56 -- the first calls after parseCommandLine(): the options are final here */
int bwt_index(char *ref_path) { push_options(); return lfgpu::bwt_index(ref_path); }
int bwt_load(char *ref_path)  { push_options(); return lfgpu::bwt_load(ref_path); }
/* src/baseFAST.cpp:62,66,73,80 */
void initializeFAST() { lfgpu::initializeFAST(); }
void finalizeFAST()   { lfgpu::finalizeFAST(); if (lfgpu::lf_global_output && lfgpu::lf_global_output != stdout) fclose(lfgpu::lf_global_output); }
void initFASTChunk(Read *seqList, int seqListSize)
{
    static_assert(sizeof(Read) == sizeof(lfgpu::Read), "Read layout (src/Reads.h:28-35)");
    lfgpu::initFASTChunk(reinterpret_cast<lfgpu::Read *>(seqList), seqListSize);
}
void mapSeqMT() { lfgpu::mapSeqMT(); }
