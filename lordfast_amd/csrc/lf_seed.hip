/*
 * lf_seed.hip -- FM-index residency in HBM and the seeding kernels (gfx950).
 *
 * Reference path replaced: getLocs_extend_whole_step (src/BWT.cpp:312-394) over
 * bwt_count_exact_cached (src/BWT.cpp:265-298), bwt_2occ/bwt_occ (lib/bwa/bwt.c:107-163) and bwt_sa
 * (lib/bwa/bwt.c:86-96).
 *
 * Formulation (results identical, see DESIGN.md "seed search"):
 *   The reference restarts a whole backward search for every candidate length m+1.  Because the text is
 *   forward + reverse complement, count(P) == count(revcomp(P)), and revcomp(q[pos..pos+m)) grows to the
 *   LEFT as m grows -- so ONE backward search over the complemented read finds the maximal m, and one
 *   more backward search over q[pos..pos+m) yields the exact [sp,ep] rows whose order the seed lists keep.
 *   HBM-bound on random 64-byte Occ blocks: one lane per sample position, thousands of independent
 *   dependent-load chains per CU in flight.
 */
#include <mutex>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include "lf_gpu_common.h"
#include "lf_rsweep.h"
#include "lf_scan.h"

/* ---------------------------------------------------------------- index residency */
__global__ void lf_occ2_convert_kernel(uint32_t *__restrict__ bwt, uint64_t n_blocks);

/* src/BWT.cpp:60-115 -- one level of the 12-mer table: child (4i+j) = j prepended to k-mer i */
__global__ void lf_cache_level_kernel(lf_dev_index ix, const uint64_t *__restrict__ parent, uint64_t *__restrict__ child, uint32_t n_parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parent) return;
    const uint64_t bk = parent[2 * (size_t)i], bl = parent[2 * (size_t)i + 1];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const size_t ni = (size_t)i * 4 + j;
        uint64_t k = bk, l = bl;
        if (bk <= bl) { uint32_t dummy = 0; lf_backward_step(ix, k, l, j, dummy); }
        child[2 * ni] = k; child[2 * ni + 1] = l;
    }
}

/* Full suffix array from the sampled one.  bwt_sa (lib/bwa/bwt.c:86-96) walks every row to the next sampled row:
 * SA[k] = steps + SA[row reached], ~31 LF steps per row.  Read the other way the same identity, SA[invPsi(k)] = SA[k] - 1,
 * lets the rows be filled FROM the sampled ones: a chain starts at a sampled row (value known) and every LF step reaches
 * the row of the text position just before, until the next sampled row.  Every row is written exactly once: n LF steps
 * in total instead of ~31 n (4.7 s and 11.5 TB of index reads for a 3.1 Gbp genome before; the values are the same).
 * Chains have geometric lengths (mean 32, max in the hundreds), so a lane is not tied to one chain: each loop iteration
 * is ONE LF step of whatever chain the lane is on, and a lane that reaches a sampled row takes its next chain.
 * Row 0 (the sentinel suffix) is stored as -1 like the reference does (sa[0] = -1) but precedes text position seq_len - 1. */
__global__ void lf_full_sa_kernel(lf_dev_index ix, uint8_t *__restrict__ sa_full)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ix.n_sa) return;
    uint64_t k = i << 5, v = ix.sa_sampled[i];
    lf_sa_full_put(sa_full, k, v);
    if (i == 0) v = ix.seq_len;
    for (;;) {
        k = lf_inv_psi(ix, k);
        if ((k & 31) == 0) {                       /* the next sampled row: this chain is complete */
            i += stride;
            if (i >= ix.n_sa) break;
            k = i << 5; v = ix.sa_sampled[i];
            lf_sa_full_put(sa_full, k, v);
        } else { --v; lf_sa_full_put(sa_full, k, v); }
    }
}

/* A search starts from TWO table entries: the interval of revcomp(P_W) (rows x1, size) and the first row of P_W itself (x0).
 * The two indices are digit-reversed complements of each other, i.e. two random lines of a table of up to 68.7 GB per sample.
 * Packed form, same 16 bytes per k-mer: entry i = (first row of k-mer i : 33 bits, interval size : 33 bits, first row of the
 * reverse-complement k-mer : 33 bits) -- ONE line per start.  In place: a thread owns the pair (i, rc(i)). */
#define LF_M33 ((1ull << 33) - 1)
__device__ __forceinline__ uint32_t lf_kmer_rc_index(uint32_t i, int K)
{
    uint32_t x = __brev(i);
    x = ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);        /* 2-bit digits reversed, bits of a digit in place */
    x >>= (32 - 2 * K);
    return x ^ (K == 16 ? 0xffffffffu : ((1u << (2 * K)) - 1u));
}
__global__ void lf_cache_pack_kernel(uint64_t *__restrict__ tab, int K, uint64_t n)
{
    /* grid-stride: 4^16 entries are more work-items than one dispatch can have (2^32 - 1) */
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t j = lf_kmer_rc_index((uint32_t)i, K);
        if ((uint64_t)j < i) continue;
        const uint64_t ab = tab[2 * i], ae = tab[2 * i + 1], bb = tab[2 * (size_t)j], be = tab[2 * (size_t)j + 1];
        const uint64_t as = ab <= ae ? ae - ab + 1 : 0, bs = bb <= be ? be - bb + 1 : 0;
        tab[2 * i] = (ab & LF_M33) | ((as & 0x7fffffffull) << 33); tab[2 * i + 1] = (bb & LF_M33) | ((as >> 31) << 33);
        if ((uint64_t)j != i) { tab[2 * (size_t)j] = (bb & LF_M33) | ((bs & 0x7fffffffull) << 33); tab[2 * (size_t)j + 1] = (ab & LF_M33) | ((bs >> 31) << 33); }
    }
}
static int lfg_pack_cache_table(hipStream_t stream, int K, uint64_t *tab)
{
    const uint64_t n = 1ull << (2 * K);
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(lf_cache_pack_kernel, dim3((unsigned)(blocks < (1u << 20) ? blocks : (1u << 20))), dim3(256), 0, stream, tab, K, n);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    return LF_OK;
}

/* 12-mer table (src/BWT.cpp:60-115), level by level, ping-pong between two buffers; *table = 4^12 pairs */
int lfg_build_cache_table(const lf_dev_index *v, hipStream_t stream, int K, uint64_t **table)
{
    uint64_t *bufA, *bufB;
    if (hipMalloc(&bufA, ((size_t)1 << (2 * K)) * 16) != hipSuccess) { (void)hipGetLastError(); lf_set_error("no memory for the %d-mer table", K); return LF_ERR_NOMEM; }
    if (hipMalloc(&bufB, ((size_t)1 << (2 * (K - 1))) * 16) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(bufA); lf_set_error("no memory for the %d-mer table", K); return LF_ERR_NOMEM; }
    uint64_t *cur = (K % 2 == 0) ? bufA : bufB, *nxt = (K % 2 == 0) ? bufB : bufA;   /* level K lands in bufA */
    const uint64_t root[2] = { 0, v->seq_len };
    HIPCHK(hipMemcpy(cur, root, 16, hipMemcpyHostToDevice));
    for (int k = 0; k < K; k++) {
        const uint32_t np = 1u << (2 * k);
        hipLaunchKernelGGL(lf_cache_level_kernel, dim3((unsigned)(((size_t)np + 255) / 256)), dim3(256), 0, stream, *v, cur, nxt, np);
        uint64_t *t = cur; cur = nxt; nxt = t;
    }
    HIPCHK(hipStreamSynchronize(stream));
    if (cur != bufA) { lf_set_error("cache table ended in the wrong buffer"); return LF_ERR_HIP; }
    HIPCHK(hipFree(bufB));
    *table = bufA;
    return LF_OK;
}

extern "C" int lfg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int lfg_index_upload(struct lf_index *ix, const uint32_t *bwt, const uint64_t *sa_sampled)
{
    /* The chunk drivers wait for the GPU many times per chunk.  HIP's default wait spins on the CPU; with eight drivers
     * that burns half of a 16-CPU quota.  Blocking waits give those cores to the worker pool (LF_SPIN_WAIT=1 keeps
     * the default). */
    { (void)hipSetDevice(ix->device); (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); }

    HIPCHK(hipSetDevice(ix->device));
    lfg_quiesce(ix->device);                 /* the hipMallocs below synchronise the device: see lf_mem.hip */
    lf_dev_state *st = new lf_dev_state();
    memset(st, 0, sizeof(*st));
    ix->dev = st;
    HIPCHK(hipStreamCreate(&st->stream));
    lf_dev_index &v = st->view;
    v.primary = ix->primary; v.seq_len = ix->seq_len; v.l_pac = ix->l_pac; v.n_sa = ix->n_sa;
    for (int i = 0; i < 5; i++) v.L2[i] = ix->L2[i];

    const size_t bwt_bytes = ix->bwt_size * 4;
    HIPCHK(hipMalloc(&st->bwt, bwt_bytes + 256));
    HIPCHK(hipMemset(st->bwt, 0, bwt_bytes + 256));
    HIPCHK(hipMemcpy(st->bwt, bwt, bwt_bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&st->sa_sampled, ix->n_sa * 8));
    HIPCHK(hipMemcpy(st->sa_sampled, sa_sampled, ix->n_sa * 8, hipMemcpyHostToDevice));
    const size_t pac_bytes = (size_t)(ix->l_pac / 4 + 1);
    HIPCHK(hipMalloc(&st->pac, pac_bytes + 16));
    HIPCHK(hipMemcpy(st->pac, ix->pac, pac_bytes, hipMemcpyHostToDevice));
    v.bwt = (const uint32_t *)st->bwt; v.sa_sampled = (const uint64_t *)st->sa_sampled; v.pac = (const uint8_t *)st->pac;

    {
        uint64_t *tab = nullptr;
        int rc = lfg_build_cache_table(&v, st->stream, 12, &tab);
        if (rc == LF_OK) rc = lfg_pack_cache_table(st->stream, 12, tab);
        if (rc != LF_OK) return rc;
        st->cache = tab;
        v.cache = tab;
        /* a 14-mer table pays when it is not larger than the text itself (LF_WIDE_TABLE=0/1 overrides) */
        const char *wt = lf_env("LF_WIDE_TABLE");
        if (wt ? atoi(wt) != 0 : ix->seq_len >= (1ull << 28)) {
            rc = lfg_build_cache_table(&v, st->stream, 14, &tab);
            if (rc == LF_OK) rc = lfg_pack_cache_table(st->stream, 14, tab);
            if (rc != LF_OK) return rc;
            st->cache14 = tab; v.cache14 = tab;
        }
        /* the 16-mer table (68.7 GB + 17 GB while it is built): only where it leaves the batch buffers plenty of room.  On a
         * human-size text three 16-mers in four occur somewhere, so most samples start their search four steps later.
         * LF_TABLE16=0/1 overrides. */
        const char *t16 = lf_env("LF_TABLE16");
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const size_t need16 = (((size_t)1 << 32) + ((size_t)1 << 30)) * 16 + ((size_t)8 << 30);
        if (t16 ? atoi(t16) != 0 : (ix->seq_len >= (1ull << 30) && free_b > need16 + ((size_t)100 << 30) + ((ix->flags & LF_IDX_FULL_SA) ? (ix->seq_len + 1) * LF_SA_ROW_BYTES : 0))) {
            rc = lfg_build_cache_table(&v, st->stream, 16, &tab);
            if (rc == LF_OK) { rc = lfg_pack_cache_table(st->stream, 16, tab); if (rc != LF_OK) (void)hipFree(tab); }
            if (rc == LF_OK) { st->cache16 = tab; v.cache16 = tab; }
            else if (t16) return rc;                    /* asked for explicitly */
        }
    }

    if (ix->flags & LF_IDX_FULL_SA) {
        const uint64_t rows = ix->seq_len + 1;
        HIPCHK(hipMalloc(&st->sa_full, rows * LF_SA_ROW_BYTES + 16));
        hipLaunchKernelGGL(lf_full_sa_kernel, dim3(256 * 16), dim3(256), 0, st->stream, v, (uint8_t *)st->sa_full);
        HIPCHK(hipStreamSynchronize(st->stream));
        v.sa_full = (const uint8_t *)st->sa_full;
    }
    {   /* everything that reads bwa's file layout is built: the blocks take the form the mapping kernels read (in place) */
        const uint64_t n_blocks = (ix->seq_len + 127) >> 7;
        if ((n_blocks << 4) * 4 > bwt_bytes + 256) { lf_set_error("BWT array shorter than its %llu occurrence blocks", (unsigned long long)n_blocks); return LF_ERR_ARG; }
        hipLaunchKernelGGL(lf_occ2_convert_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, st->stream, (uint32_t *)st->bwt, n_blocks);
        HIPCHK(hipStreamSynchronize(st->stream));
        v.occ2 = (const uint8_t *)st->bwt; v.bwt = nullptr;
    }
    HIPCHK(hipGetLastError());
    return LF_OK;
}

extern "C" void lfg_index_free(struct lf_index *ix)
{
    lf_dev_state *st = (lf_dev_state *)ix->dev;
    if (!st) return;
    (void)hipSetDevice(ix->device);
    lfg_drain_check(ix->device);
    lfg_quiesce(ix->device);
    if (st->stream) (void)hipStreamSynchronize(st->stream);
    if (st->bwt) (void)hipFree(st->bwt);
    if (st->sa_sampled) (void)hipFree(st->sa_sampled);
    if (st->sa_full) (void)hipFree(st->sa_full);
    if (st->cache) (void)hipFree(st->cache);
    if (st->cache14) (void)hipFree(st->cache14);
    if (st->cache16) (void)hipFree(st->cache16);
    if (st->pac) (void)hipFree(st->pac);
    if (st->ctg_names) (void)hipFree(st->ctg_names);
    if (st->ctg_name_off) (void)hipFree(st->ctg_name_off);
    if (st->stream) (void)hipStreamDestroy(st->stream);
    delete st;
    ix->dev = NULL;
}

/* which index structures are resident (bench.py's config.index; a table that did not fit changes speed, never results) */
extern "C" int lfg_index_describe(const struct lf_index *ix, char *buf, size_t cap)
{
    const lf_dev_state *st = (const lf_dev_state *)ix->dev;
    if (!st || !buf || cap < 2) return LF_ERR_ARG;
    const double GB = 1e9;
    int o = snprintf(buf, cap, "BWT + Occ %.2f GB (64-byte blocks: cumulative counts + two bit planes of 128 symbols); ", ix->bwt_size * 4 / GB);
    if (st->sa_full) o += snprintf(buf + o, o < (int)cap ? cap - o : 0, "full suffix array (5 bytes per row) %.1f GB; ", (ix->seq_len + 1) * 5 / GB);
    o += snprintf(buf + o, o < (int)cap ? cap - o : 0, "sampled suffix array (every 32nd row) %.2f GB; k-mer tables: 12 (%.2f GB)", ix->n_sa * 8 / GB, (double)(1ull << 24) * 16 / GB);
    if (st->cache14) o += snprintf(buf + o, o < (int)cap ? cap - o : 0, ", 14 (%.1f GB)", (double)(1ull << 28) * 16 / GB);
    if (st->cache16) o += snprintf(buf + o, o < (int)cap ? cap - o : 0, ", 16 (%.1f GB)", (double)(1ull << 32) * 16 / GB);
    o += snprintf(buf + o, o < (int)cap ? cap - o : 0, "; 2-bit reference %.2f GB", (double)(ix->l_pac / 4 + 1) / GB);
    return LF_OK;
}

/* ---------------------------------------------------------------- seeding kernels */

/* sample positions: seed_pos accumulates `step` in FP64 exactly like src/BWT.cpp:320-321,388-389
 * (sequential adds, truncation), one lane per read. Layout pos[i * n_reads + r] (coalesced). */
__global__ void lf_seed_pos_kernel(int n_reads, const uint64_t *__restrict__ off, uint32_t hash_count,
                                   uint32_t *__restrict__ pos_by_sample /* [r * hash_count + i]: coalesced for the per-sample kernels */)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t qLen = (uint32_t)(off[r + 1] - off[r]);
    const double step = (double)qLen / hash_count;
    double sp = 0;
    uint32_t p = 0;
    for (uint32_t i = 0; i < hash_count; i++) {
        pos_by_sample[(size_t)r * hash_count + i] = p;
        sp += step;
        p = (uint32_t)sp;
    }
}

struct lf_sample_t { uint64_t sp; uint32_t occ; uint32_t m; };   /* occ saturates at 2^32-1; m = 0: no seed */

/* the resident form of the occurrence blocks (lf_gpu_common.h: lf_occ2_*), made in place out of bwa's file layout once the
 * k-mer tables and the full suffix array are built: a thread owns one 64-byte block */
__global__ void lf_occ2_convert_kernel(uint32_t *__restrict__ bwt, uint64_t n_blocks)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t *blk = bwt + (b << 4);
    uint64_t c[4]; uint32_t w[8];
    for (int j = 0; j < 4; j++) c[j] = reinterpret_cast<const uint64_t *>(blk)[j];
    for (int j = 0; j < 8; j++) w[j] = blk[8 + j];
    uint64_t lo[2] = { 0, 0 }, hi[2] = { 0, 0 };
    for (int i = 0; i < 128; i++) {
        const uint32_t sym = (w[i >> 4] >> ((~(uint32_t)i & 15u) << 1)) & 3u;       /* 16 symbols per word, first one in the top bits (lib/bwa/bwt.h:78) */
        lo[i >> 6] |= (uint64_t)(sym & 1u) << (i & 63); hi[i >> 6] |= (uint64_t)(sym >> 1) << (i & 63);
    }
    uint64_t *o = reinterpret_cast<uint64_t *>(blk);
    o[0] = c[0]; o[1] = c[0] + c[1]; o[2] = c[0] + c[1] + c[2]; o[3] = c[0] + c[1] + c[2] + c[3];
    o[4] = lo[0]; o[5] = lo[1]; o[6] = hi[0]; o[7] = hi[1];
}

/* bit j of x -> bit 2 j (16 digits of a table index) */
__device__ __forceinline__ uint32_t lf_spread16(uint32_t x)
{
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x;
}

/* Maximal exact match of every (read, sample) starting at its sample position, >= k long (bwt_count_exact_cached restarted per
 * length in the reference, src/BWT.cpp:330-342; here ONE bidirectional pass, see DESIGN.md "seed search formulation").
 *
 * The kernel is a stream of DEPENDENT random 64-byte reads: table entry -> occurrence block -> occurrence block ..., two to
 * four per sample, each a row activation somewhere in 70 GB.  What it can reach is set by how many of them are in flight, so
 *   - a lane runs S searches at once (S slots; their loads are issued together, then awaited together), and a slot whose search
 *     ends takes the wavefront's next sample (match lengths are very uneven: ballot + prefix count on a wave-uniform cursor);
 *   - a trip costs every slot exactly ONE block: a step whose two rows (k - 1 and l of lib/bwa/bwt.c:132-139) lie in different
 *     blocks -- a few per cent of the steps -- is done in two trips (first row's counts parked in the slot), so that no wavefront
 *     ever executes a second, mostly idle, instruction stream for them;
 *   - the bases come from the batch's bit planes (three unaligned 8-byte windows = 56 bases: table index by bit tricks, the next
 *     32 codes in two registers) instead of 24 byte loads and 24 classifications; batches with lower-case bases (which the
 *     reference's nst_nt4_table accepts and the planes do not) take the byte variant (PLANES = false);
 *   - a block holds cumulative counts and two bit planes (lf_occ2_*): counts of "== a" and "> a" in front of both rows are three
 *     masked popcounts of 128-bit masks.
 * A wavefront owns the samples of `rpw` whole reads (no division per sample when rpw <= 2). */
#define LF_SEARCH_MIN_SPAN 1024
template <int S, bool PLANES>
__global__ void __launch_bounds__(256)
lf_seed_search_kernel(lf_dev_index ix, int n_reads, const char *__restrict__ reads, const uint64_t *__restrict__ planes, int64_t q_words,
                      const uint64_t *__restrict__ off, uint32_t hash_count, uint32_t rpw, int kmin, const uint32_t *__restrict__ pos,
                      lf_sample_t *__restrict__ out, unsigned long long *__restrict__ counters)
{
    const uint32_t wave = (uint32_t)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t r0 = wave * rpw;
    uint32_t n_cache = 0, n_blk = 0;
    const bool lower = counters[4] != 0ull;              /* the batch holds lower-case acgt (lf_pack_planes_kernel / the host's exception list) */
    if (r0 < (uint32_t)n_reads && lower != PLANES) {
    const uint32_t r1 = r0 + rpw < (uint32_t)n_reads ? r0 + rpw : (uint32_t)n_reads;
    uint32_t nxt = r0 * hash_count; const uint32_t first = nxt, end = r1 * hash_count;
    /* table width W: 14 when the wide table exists and -k >= 14 (a match shorter than k is dropped anyway), else the reference's 12 */
    const bool wide = ix.cache14 != nullptr && kmin >= 14;
    const uint32_t W = wide ? 14u : 12u;
    const uint64_t *__restrict__ tabW = wide ? ix.cache14 : ix.cache;
    const uint64_t *__restrict__ plo = planes, *__restrict__ phi = planes + q_words, *__restrict__ pva = planes + 2 * q_words;
    /* the first bases of the wavefront's (first three) reads */
    const uint64_t offa = off[r0], offb = off[r0 + 1 <= (uint32_t)n_reads ? r0 + 1 : (uint32_t)n_reads], offc = off[r0 + 2 <= (uint32_t)n_reads ? r0 + 2 : (uint32_t)n_reads];
    const float inv_hc = 1.0f / (float)hash_count;

    /* slot state.  ph: 0 idle | 1 table entry of the 16-mer wanted | 3 of the W-mer | 2 stepping | 4 code window used up, next one wanted |
     * 5 second half of a step across two blocks (bits 8..14 row's place in the block, 16..17 symbol, 18 bit 32 of the parked count,
     * 19 the step moves x0 past the sentinel row) */
    uint32_t ph[S], m[S], rlen[S], gidx[S], alo[S], ahi[S], aw[S], tix[S], gks[S];
    uint64_t P[S], x0[S], x1[S], sz[S];
#pragma unroll
    for (int s = 0; s < S; s++) { ph[s] = 0; m[s] = 0; rlen[s] = 0; gidx[s] = 0; alo[s] = 0; ahi[s] = 0; aw[s] = 0; tix[s] = 0; gks[s] = 0; P[s] = 0; x0[s] = 0; x1[s] = 0; sz[s] = 0; }

    for (;;) {
        /* ---- (1) idle slots take the wavefront's next samples; the code windows of new and of exhausted slots are asked for ---- */
        uint64_t wl[S], wh[S], wv[S]; uint32_t wsh[S]; bool win[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
            win[s] = ph[s] == 4u; wl[s] = wh[s] = wv[s] = 0; wsh[s] = 0;
            const uint64_t idle = lf_ballot(ph[s] == 0u);
            if (idle && nxt < end) {
                const uint32_t cand = nxt + (uint32_t)__popcll(idle & below);
                if (ph[s] == 0u && cand < end) {
                    const uint32_t local = cand - first;
                    uint32_t rr;
                    if (rpw <= 2u) rr = local >= hash_count ? 1u : 0u;
                    else { rr = (uint32_t)((float)local * inv_hc); const int rem = (int)(local - rr * hash_count); rr = rem < 0 ? rr - 1u : (rem >= (int)hash_count ? rr + 1u : rr); }
                    uint64_t ob, oe;
                    if (rpw <= 2u) { ob = rr ? offb : offa; oe = rr ? offc : offb; } else { ob = off[r0 + rr]; oe = off[r0 + rr + 1]; }
                    const uint32_t p = pos[cand];
                    gidx[s] = cand; P[s] = ob + p; m[s] = 0;
                    const uint32_t qLen = (uint32_t)(oe - ob);
                    rlen[s] = qLen > p ? qLen - p : 0u;
                    ph[s] = 1u; win[s] = true;
                }
                nxt += (uint32_t)__popcll(idle); nxt = nxt > end ? end : nxt;
            }
            if (win[s]) {
                const uint64_t bitpos = P[s] + m[s];
                if (PLANES) {
                    const uint64_t by = bitpos >> 3; wsh[s] = (uint32_t)(bitpos & 7);
                    __builtin_memcpy(&wl[s], reinterpret_cast<const char *>(plo) + by, 8); __builtin_memcpy(&wh[s], reinterpret_cast<const char *>(phi) + by, 8);
                    __builtin_memcpy(&wv[s], reinterpret_cast<const char *>(pva) + by, 8);
                } else {      /* bytes: 24 of them (the batch buffer has 64 bytes of slack) */
                    __builtin_memcpy(&wl[s], reads + bitpos, 8); __builtin_memcpy(&wh[s], reads + bitpos + 8, 8); __builtin_memcpy(&wv[s], reads + bitpos + 16, 8);
                }
            }
        }
        /* ---- (2) stepping slots: the next base and the block of this trip ---- */
        uint32_t bidx[S], rA[S], rB[S], sym[S]; bool go[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
            go[s] = false; bidx[s] = 0; rA[s] = 0; rB[s] = 1; sym[s] = 0;
            if (ph[s] == 2u) {
                const uint32_t ab = aw[s] & 0xffffffu, an = (aw[s] >> 24) & 63u, i = m[s] - ab;
                if (i < an) {
                    const uint32_t c = ((alo[s] >> i) & 1u) | (((ahi[s] >> i) & 1u) << 1);
                    const uint32_t a = 3u - c;                                   /* symbol prepended on the complement strand */
                    /* bwt_extend, is_back = 0 (lib/bwa/bwt.c:262-275): one backward step on x1 with the counts of rows x1 - 1 and x1 - 1 + s */
                    const uint64_t km = x1[s] - 1, l = km + sz[s];
                    const uint64_t kk = km - (km >= ix.primary ? 1ull : 0ull), ll = l - (l >= ix.primary ? 1ull : 0ull);
                    const uint32_t bK = (uint32_t)(kk >> 7), bL = (uint32_t)(ll >> 7), remK = (uint32_t)(kk & 127) + 1u, remL = (uint32_t)(ll & 127) + 1u;
                    sym[s] = a; bidx[s] = bK; go[s] = true;
                    if (bK == bL) { rA[s] = remK; rB[s] = remL; n_blk += 1u; }
                    else {
                        /* two blocks: this trip the first row's counts, the next trip the second row's (touches counted like bwt_2occ) */
                        rA[s] = 0; rB[s] = remK; n_blk += 2u;
                        tix[s] = bL;
                        const uint32_t sent = (x1[s] <= ix.primary && x1[s] + sz[s] - 1 >= ix.primary) ? 1u : 0u;
                        ph[s] = 6u | ((remL - 1u) << 8) | (a << 16) | (sent << 19);      /* 6: first half (this trip); becomes 5 */
                    }
                } else if (aw[s] >> 31) ph[s] = 4u;                              /* window used up, the read goes on: fetch the next 32 codes */
                else ph[s] = 7u;                                                  /* no further base: the search ends (written in (4)) */
            } else if ((ph[s] & 255u) == 5u) {
                go[s] = true; bidx[s] = tix[s]; rA[s] = 0; rB[s] = ((ph[s] >> 8) & 127u) + 1u; sym[s] = (ph[s] >> 16) & 3u;
            }
        }
        ulonglong2 hd[S], blo[S], bhi[S]; uint64_t c3[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
            hd[s] = make_ulonglong2(0, 0); blo[s] = hd[s]; bhi[s] = hd[s]; c3[s] = 0;
            if (go[s]) {
                const uint8_t *blk = ix.occ2 + ((uint64_t)bidx[s] << 6);
                hd[s] = *reinterpret_cast<const ulonglong2 *>(blk + 8u * (sym[s] ? sym[s] - 1u : 0u));      /* (C[a-1], C[a]); a = 0: (C[0], C[1]) */
                c3[s] = *reinterpret_cast<const uint64_t *>(blk + 24);
                blo[s] = *reinterpret_cast<const ulonglong2 *>(blk + 32); bhi[s] = *reinterpret_cast<const ulonglong2 *>(blk + 48);
            }
        }
        /* ---- (3) new code windows: usable bases, table index; the table entries of this trip ---- */
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (win[s]) {
                uint64_t lo56, hi56; uint32_t lead;
                if (PLANES) {
                    lo56 = wl[s] >> wsh[s]; hi56 = wh[s] >> wsh[s];
                    const uint64_t bad = ~(wv[s] >> wsh[s]) | (1ull << 56);
                    lead = (uint32_t)__ffsll((long long)bad) - 1u;               /* usable bases in front of the first one outside ACGT */
                } else {
                    lo56 = 0; hi56 = 0; uint32_t bad = 1u << 24;
#pragma unroll
                    for (int t = 0; t < 24; t++) {
                        const int cd = lf_nt4((unsigned char)((t < 8 ? wl[s] >> (8 * t) : t < 16 ? wh[s] >> (8 * (t - 8)) : wv[s] >> (8 * (t - 16))) & 0xff));
                        bad |= (cd > 3 ? 1u : 0u) << t; lo56 |= (uint64_t)(cd & 1) << t; hi56 |= (uint64_t)((cd >> 1) & 1) << t;
                    }
                    lead = (uint32_t)__ffs((int)bad) - 1u;
                }
                const uint32_t left = rlen[s] > m[s] ? rlen[s] - m[s] : 0u;       /* bases of the read from here on */
                lead = lead < left ? lead : left;
                if (ph[s] == 4u) {
                    /* a search in progress: the codes of offsets m .. m + 31 */
                    const uint32_t CAP = PLANES ? 32u : 24u, an = lead < CAP ? lead : CAP;
                    alo[s] = (uint32_t)lo56; ahi[s] = (uint32_t)hi56;
                    aw[s] = m[s] | (an << 24) | (an == CAP ? 0x80000000u : 0u);
                    ph[s] = an ? 2u : 7u;
                } else {
                    /* a new sample.  Table index of revcomp(P_W): base-4 number whose digit W - 1 - t is the complement of P's character t
                     * (the LAST character of a pattern is the most significant digit, src/BWT.cpp:270-277): the complemented code bits,
                     * bit-reversed, interleaved.  idx14 = idx16 >> 4, idx12 = idx16 >> 8. */
                    const uint32_t L16 = __brev(~(uint32_t)lo56) >> 16, H16 = __brev(~(uint32_t)hi56) >> 16;
                    const uint32_t idx16 = lf_spread16(L16) | (lf_spread16(H16) << 1);
                    const uint32_t CAP = PLANES ? 32u : 12u;
                    const uint32_t an = lead > 12u ? (lead - 12u < CAP ? lead - 12u : CAP) : 0u;
                    alo[s] = (uint32_t)(lo56 >> 12); ahi[s] = (uint32_t)(hi56 >> 12);
                    aw[s] = 12u | (an << 24) | (an == CAP ? 0x80000000u : 0u) | (lead >= W ? 0x40000000u : 0u);
                    /* widest table first: a 16-mer that occurs starts the search at m = 16 (nothing shorter than kmin is wanted, and a sample
                     * that reaches 16 would have passed through every shorter length with a non-empty interval) */
                    if (rlen[s] >= (uint32_t)kmin && ix.cache16 != nullptr && lead >= 16u) { ph[s] = 1u; tix[s] = idx16; }
                    else if (rlen[s] >= (uint32_t)kmin && lead >= W) { ph[s] = 3u; tix[s] = idx16 >> (32u - 2u * W); }
                    else ph[s] = 7u;
                }
            }
        }
        ulonglong2 te[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
            te[s] = make_ulonglong2(0, 0);
            if (ph[s] == 1u || ph[s] == 3u) {
                const uint64_t *T = ph[s] == 1u ? ix.cache16 : tabW;
                te[s] = *reinterpret_cast<const ulonglong2 *>(T + 2 * (size_t)tix[s]);
                n_cache += 1u;
            }
        }
        /* ---- (4) this trip's blocks and table entries ---- */
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (go[s]) {
                const uint32_t a = sym[s];
                const lf_occ2_masks M = lf_occ2_eq_gt(blo[s].x, blo[s].y, bhi[s].x, bhi[s].y, (int)a);
                uint64_t a0, a1, b0, b1;
                lf_occ2_first(rA[s], a0, a1); lf_occ2_first(rB[s], b0, b1);
                const uint64_t d0 = a0 ^ b0, d1 = a1 ^ b1;                        /* rows rA .. rB - 1 of the block (rA <= rB) */
                const uint32_t EA = (uint32_t)__popcll(M.e0 & a0) + (uint32_t)__popcll(M.e1 & a1);
                const uint32_t ED = (uint32_t)__popcll(M.e0 & d0) + (uint32_t)__popcll(M.e1 & d1);
                const uint32_t GD = (uint32_t)__popcll(M.g0 & d0) + (uint32_t)__popcll(M.g1 & d1);
                const uint64_t Ca = a ? hd[s].y : hd[s].x, Cm = a ? hd[s].x : 0ull;
                const uint64_t HE = Ca - Cm, HG = c3[s] - Ca;                     /* occurrences of a / of everything above a in front of the block */
                const uint64_t L2a = a == 0u ? ix.L2[0] : a == 1u ? ix.L2[1] : a == 2u ? ix.L2[2] : ix.L2[3];
                const uint32_t mode = ph[s] & 255u;
                if (mode == 6u) {
                    /* first half: park ek (as the row it leads to) and gk; x1 is not needed any more */
                    const uint64_t gk = HG + GD;
                    x1[s] = L2a + 1 + HE + ED;
                    gks[s] = (uint32_t)gk;
                    ph[s] = (ph[s] & ~255u & ~(1u << 18)) | 5u | ((uint32_t)(gk >> 32) & 1u) << 18;
                } else {
                    uint64_t ns, gd, nx1; uint32_t sent;
                    if (mode == 2u) { ns = ED; gd = GD; nx1 = L2a + 1 + HE + EA; sent = (x1[s] <= ix.primary && x1[s] + sz[s] - 1 >= ix.primary) ? 1u : 0u; }
                    else {
                        const uint64_t el = HE + ED, gl = HG + GD, ek = x1[s] - L2a - 1, gk = (uint64_t)gks[s] | ((uint64_t)((ph[s] >> 18) & 1u) << 32);
                        ns = el - ek; gd = gl - gk; nx1 = x1[s]; sent = (ph[s] >> 19) & 1u;
                    }
                    if (ns != 0) {
                        /* rows of P.d for d = 0..3 follow each other inside P's interval in the order of d; on the complement strand d appears
                         * as 3 - d, so P.c starts after the sentinel row (if P's complement interval spans it) and all a' > a */
                        x0[s] += sent + gd; x1[s] = nx1; sz[s] = ns; m[s] += 1u; ph[s] = 2u;
                    } else ph[s] = 7u;
                }
            }
            if (ph[s] == 1u || ph[s] == 3u) {
                const uint64_t size = (te[s].x >> 33) | ((te[s].y >> 33) << 31);
                if (size != 0) { x1[s] = te[s].x & LF_M33; x0[s] = te[s].y & LF_M33; sz[s] = size; m[s] = ph[s] == 1u ? 16u : W; ph[s] = 2u; }
                else if (ph[s] == 1u && (aw[s] & 0x40000000u)) { ph[s] = 3u; tix[s] >>= (32u - 2u * W); }      /* the 16-mer does not occur: the narrower table */
                else ph[s] = 7u;
            }
            if (ph[s] == 7u) {
                lf_sample_t res; res.sp = 0; res.occ = 0; res.m = 0;
                if ((int)m[s] >= kmin) { res.sp = x0[s]; res.m = m[s]; res.occ = sz[s] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sz[s]; }
                out[gidx[s]] = res;
                ph[s] = 0u; m[s] = 0;
            }
        }
        bool busy = false;
#pragma unroll
        for (int s = 0; s < S; s++) busy = busy || ph[s] != 0u;
        if (!lf_any(busy) && nxt >= end) break;
    }
    }
    /* SURVEY 8(d) counters: one atomic pair per BLOCK (same-address atomics serialise) */
    __shared__ unsigned long long s_cnt[2];
    if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) { n_cache += __shfl_down(n_cache, o); n_blk += __shfl_down(n_blk, o); }
    if (lane == 0 && (n_cache | n_blk)) { atomicAdd(&s_cnt[0], (unsigned long long)n_cache); atomicAdd(&s_cnt[1], (unsigned long long)n_blk); }
    __syncthreads();
    if (threadIdx.x == 0 && (s_cnt[0] | s_cnt[1])) { atomicAdd(&counters[0], s_cnt[0]); atomicAdd(&counters[1], s_cnt[1]); }
}

/* acceptance is sequential per read: 0 < occ < MAX_REF_HITS and not contained in the previous accepted seed
 * (src/BWT.cpp:345,386: pos + len > lastPos, lastPos = end of the last ACCEPTED seed).  A usable seed that is rejected ends
 * at or before lastPos, so lastPos is also the maximum end over ALL usable seeds before it: acceptance is an exclusive prefix
 * maximum.  One wavefront per read, 64 samples per trip (coalesced 16-byte records), the maximum carried from trip to trip;
 * writes the number of hits to locate per sample. */
__device__ __forceinline__ uint32_t lf_wave_incl_max(uint32_t v)
{
    uint32_t x = v;
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));      /* row_shr:1 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));      /* row_shr:2 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));      /* row_shr:4 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));      /* row_shr:8 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));     /* row_bcast:15 -> rows 1, 3 */
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));     /* row_bcast:31 -> rows 2, 3 */
    return x;
}
__global__ void __launch_bounds__(256)
lf_seed_accept_kernel(int n_reads, uint32_t hash_count, uint32_t max_ref_hits, const uint32_t *__restrict__ pos /* [r * hash_count + i] */,
                      const lf_sample_t *__restrict__ smp, uint32_t *__restrict__ cnt)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const size_t row = (size_t)r * hash_count;
    uint32_t last_pos = 0;                                   /* wave-uniform */
    for (uint32_t i0 = 0; i0 < hash_count; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        uint32_t end = 0, occ = 0;                           /* end 0: not usable (a usable seed ends at >= 1) */
        if (i < hash_count) {
            const lf_sample_t s = smp[row + i];
            if (s.m && s.occ > 0 && s.occ < max_ref_hits) { end = pos[row + i] + s.m; occ = s.occ; }
        }
        const uint32_t inc = lf_wave_incl_max(end);
        uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138, 0xf, 0xf, false);      /* wave_shr:1, lane 0: 0 */
        before = max(before, last_pos);
        if (i < hash_count) cnt[row + i] = end > before ? occ : 0u;
        last_pos = max(last_pos, (uint32_t)__builtin_amdgcn_readlane((int)inc, 63));
    }
}

/* locate: rows sp..sp+cnt-1 of every accepted sample -> (tPos, qPos|len<<20, strand) in sample order then
 * SA-row order (src/BWT.cpp:348-384).
 * A wavefront owns 64 consecutive samples and spreads ITS HITS (not its samples) over the lanes: hit h of the wave
 * belongs to the last sample whose first hit is <= h (hit_off is the global exclusive scan, so the wave's output is
 * one contiguous range).  Output writes are fully coalesced, suffix-array reads are contiguous within a sample, and
 * a repeat with 999 hits no longer serialises 63 idle lanes behind it. */
__global__ void __launch_bounds__(256)
lf_seed_locate_kernel(lf_dev_index ix, int n_reads, const uint64_t *__restrict__ off, uint32_t hash_count,
                      const uint32_t *__restrict__ pos, const lf_sample_t *__restrict__ smp, const uint32_t *__restrict__ cnt,
                      const uint64_t *__restrict__ hit_off, uint32_t *__restrict__ tpos, uint32_t *__restrict__ qpl,
                      uint8_t *__restrict__ strand, unsigned long long *__restrict__ counters)
{
    __shared__ uint64_t s_sp[256]; __shared__ uint32_t s_rel[256], s_m[256], s_p[256], s_ql[256];
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_reads * hash_count;
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    uint32_t n_blk = 0, n_sa = 0;
    const uint64_t l_pac = (uint64_t)ix.l_pac;
    uint32_t c = 0; uint64_t o = 0;
    if (gid < total) { c = cnt[gid]; o = hit_off[gid]; }
    else o = hit_off[total];
    const uint64_t o0 = __shfl(o, 0);
    const uint32_t Tw = (uint32_t)(__shfl(o, 63) - o0) + __shfl(c, 63);
    if (Tw == 0) return;
    if (c) {
        const int r = (int)(gid / hash_count);
        const uint32_t i = (uint32_t)(gid % hash_count);
        const lf_sample_t s = smp[gid];
        s_sp[threadIdx.x] = s.sp; s_m[threadIdx.x] = s.m; s_p[threadIdx.x] = pos[gid]; s_ql[threadIdx.x] = (uint32_t)(off[r + 1] - off[r]);
    }
    s_rel[threadIdx.x] = (uint32_t)(o - o0);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (uint32_t h = (uint32_t)lane; h < Tw; h += 64) {
        int lo = 0, hi = 63;                         /* last sample of the wave whose first hit is <= h */
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_rel[wbase + mid] <= h) lo = mid; else hi = mid - 1; }
        const int k = wbase + lo;
        const uint32_t j = h - s_rel[k], m = s_m[k], p = s_p[k], qLen = s_ql[k];
        const uint64_t row = s_sp[k] + j;
        const uint64_t sapos = ix.sa_full ? lf_sa_full_get(ix.sa_full, row) : lf_sa_walk(ix, row, n_blk);
        n_sa++;
        uint32_t t, qp; uint8_t rv;
        if (sapos >= l_pac) { t = (uint32_t)((l_pac << 1) - sapos - m); qp = qLen - p - m; rv = 1; }
        else { t = (uint32_t)sapos; qp = p; rv = 0; }
        tpos[o0 + h] = t;
        qpl[o0 + h] = (qp & 0xFFFFFu) | ((m & 0xFFFu) << 20);
        strand[o0 + h] = rv;
    }
    /* N_sa of SURVEY 8(d) is the hit count (the host adds it); only the sampled-SA walk variant has block touches to
     * report -- 4 x 10^5 waves hammering one counter address were most of this kernel's time */
    (void)n_sa;
    if (!ix.sa_full) {
        for (int d = 32; d > 0; d >>= 1) n_blk += __shfl_down(n_blk, d);
        if (lane == 0 && n_blk) atomicAdd(&counters[1], (unsigned long long)n_blk);
    }
}

struct lf_widen_op { __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; } };

__global__ void lf_read_first_hit_kernel(int n_reads, uint32_t hash_count, const uint64_t *__restrict__ hit_off, uint64_t total_hits, uint64_t *__restrict__ read_off)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    read_off[r] = (r == n_reads) ? total_hits : hit_off[(size_t)r * hash_count];
}

/* device-resident input (lf_map_batch_dev): read k = src[src_off[k] .. + off[k + 1] - off[k]) -> the lane's resident batch at off[k].
 * A workgroup copies 8 bytes per lane and trip (neither side is aligned: unaligned 8-byte accesses are legal on global
 * memory), the last bytes of a read one by one. */
__global__ void __launch_bounds__(256)
lf_reads_gather_kernel(const unsigned char *__restrict__ src, const uint64_t *__restrict__ src_off, const uint64_t *__restrict__ off,
                       int n_reads, unsigned char *__restrict__ dst)
{
    const int k = blockIdx.x;
    if (k >= n_reads) return;
    const uint64_t o = off[k], len = off[k + 1] - o;
    const unsigned char *s = src + src_off[k];
    unsigned char *d = dst + o;
    const uint64_t words = len >> 3;
    for (uint64_t i = (uint64_t)blockIdx.y * 256 + threadIdx.x; i < words; i += (uint64_t)gridDim.y * 256) {
        uint64_t w; __builtin_memcpy(&w, s + 8 * i, 8); __builtin_memcpy(d + 8 * i, &w, 8);
    }
    if (blockIdx.y == 0 && threadIdx.x < (len & 7)) d[8 * words + threadIdx.x] = s[8 * words + threadIdx.x];
}
extern "C" int lfg_gather_reads(int device, void *stream, const void *d_src, const uint64_t *d_src_off, const uint64_t *d_off, int n_reads, void *d_dst)
{
    if (n_reads <= 0) return LF_OK;
    HIPCHK(hipSetDevice(device));
    hipLaunchKernelGGL(lf_reads_gather_kernel, dim3((unsigned)n_reads, 8), dim3(256), 0, (hipStream_t)stream, (const unsigned char *)d_src, d_src_off, d_off, n_reads, (unsigned char *)d_dst);
    HIPCHK(hipGetLastError());
    return LF_OK;
}

extern "C" int lfg_seed(const struct lf_index *ix, const lf_params_t *p, int n_reads, const char *reads,
                        const uint64_t *off, int want_hits, lfg_hits_t *out)
{
    return lfg_seed_src(ix, p, n_reads, reads, nullptr, nullptr, off, want_hits, out);
}

/* packed upload (lfg_seed_packed): the read bytes back from the bit planes -- 'A' 'C' 'G' 'T' by code where the valid bit is set,
 * 0 elsewhere (lf_patch_bytes_kernel then stores the few bytes that are not upper-case ACGT) */
__global__ void __launch_bounds__(256)
lf_unpack_planes_kernel(const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi, const uint64_t *__restrict__ valid, uint64_t n_bases, char *__restrict__ dst)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      /* four bases per thread */
    if (4 * t >= n_bases) return;
    const uint64_t w = t >> 4; const uint32_t sh = (uint32_t)(t & 15) * 4;
    const uint32_t l = (uint32_t)(lo[w] >> sh) & 15u, h = (uint32_t)(hi[w] >> sh) & 15u, v = (uint32_t)(valid[w] >> sh) & 15u;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t c = ((l >> k) & 1u) | (((h >> k) & 1u) << 1);
        const uint32_t ch = ((v >> k) & 1u) ? ((0x54474341u >> (c << 3)) & 0xffu) : 0u;
        out |= ch << (8 * k);
    }
    if (4 * t + 4 <= n_bases) *reinterpret_cast<uint32_t *>(dst + 4 * t) = out;      /* the buffer starts 256-byte aligned */
    else for (uint64_t k = 0; 4 * t + k < n_bases; k++) dst[4 * t + k] = (char)(out >> (8 * k));
}
__global__ void lf_patch_bytes_kernel(const uint64_t *__restrict__ pos, const uint8_t *__restrict__ byte, uint64_t n, uint64_t base, char *__restrict__ dst)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[pos[i] - base] = (char)byte[i];
}
/* a chunk of a prepacked batch: its planes are a BIT range of the batch's -- the words that hold it came up as they are (qs words per plane), this kernel
 * moves them to bit 0 and clears what lies behind the chunk's last base (the next chunk's first bases) */
__global__ void lf_planes_shift_kernel(const uint64_t *__restrict__ src, uint64_t qs, uint32_t shift, uint64_t n_bases, uint64_t *__restrict__ dst, uint64_t qw)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * qw) return;
    const uint64_t x = t / qw, w = t - x * qw;
    const uint64_t *S = src + x * qs;
    uint64_t v = 0;
    if (w < qs) { const uint64_t a = S[w], b = w + 1 < qs ? S[w + 1] : 0ull; v = shift ? (a >> shift) | (b << (64 - shift)) : a; }
    const uint64_t first = w * 64;
    if (first >= n_bases) v = 0; else if (n_bases - first < 64) v &= (1ull << (n_bases - first)) - 1;
    dst[t] = v;
}

static int lfg_seed_any(const struct lf_index *ix, const lf_params_t *p, int n_reads, const char *reads, const void *d_src, const uint64_t *src_off,
                        const lf_packed_src_t *pk, const uint64_t *off, int want_hits, lfg_hits_t *out);
extern "C" int lfg_seed_packed(const struct lf_index *ix, const lf_params_t *p, int n_reads, const lf_packed_src_t *pk, const uint64_t *off, int want_hits, lfg_hits_t *out)
{
    return lfg_seed_any(ix, p, n_reads, nullptr, nullptr, nullptr, pk, off, want_hits, out);
}
/* reads == NULL: the bases are already in HBM of this device (d_src + src_off[k], host array of n_reads offsets) */
extern "C" int lfg_seed_src(const struct lf_index *ix, const lf_params_t *p, int n_reads, const char *reads, const void *d_src, const uint64_t *src_off,
                            const uint64_t *off, int want_hits, lfg_hits_t *out)
{
    return lfg_seed_any(ix, p, n_reads, reads, d_src, src_off, nullptr, off, want_hits, out);
}
static int lfg_seed_any(const struct lf_index *ix, const lf_params_t *p, int n_reads, const char *reads, const void *d_src, const uint64_t *src_off,
                        const lf_packed_src_t *pk, const uint64_t *off, int want_hits, lfg_hits_t *out)
{
    memset(out, 0, sizeof(*out));
    lf_dev_state *st = (lf_dev_state *)ix->dev;
    if (!st) { lf_set_error("index is not on a device"); return LF_ERR_NO_DEVICE; }
    HIPCHK(hipSetDevice(ix->device));
    hipStream_t s = (hipStream_t)lfg_lane_stream(ix->device, 0);
    if (!s) return LF_ERR_HIP;
    const int dv = ix->device;
    const uint32_t hc = (uint32_t)p->sampling_count;
    const size_t total = (size_t)n_reads * hc;
    const uint64_t n_bases = off[n_reads];
    if (total + 1 >= (1ull << 31)) { lf_set_error("seed batch too large (%zu samples): split the batch", total); return LF_ERR_ARG; }

    /* persistent device buffers (lf_mem.hip): no hipMalloc in the steady state */
#define DSLOT(T, k, bytes) (T *)lfg_dev_slot(dv, LF_DS_SEED0 + (k), (bytes))
    char *d_reads = DSLOT(char, 0, n_bases + 64);
    uint64_t *d_off = DSLOT(uint64_t, 1, (size_t)(n_reads + 1) * 8);
    uint32_t *d_pos2 = DSLOT(uint32_t, 12, total * 4 + 4);
    lf_sample_t *d_smp = DSLOT(lf_sample_t, 3, total * sizeof(lf_sample_t) + 16);
    uint32_t *d_cnt = DSLOT(uint32_t, 4, (total + 1) * 4);
    uint64_t *d_hit_off = DSLOT(uint64_t, 5, (total + 1) * 8);
    uint64_t *d_read_off = DSLOT(uint64_t, 6, (size_t)(n_reads + 1) * 8);
    unsigned long long *d_counters = DSLOT(unsigned long long, 7, 64);
    if (!d_reads || !d_off || !d_pos2 || !d_smp || !d_cnt || !d_hit_off || !d_read_off || !d_counters) return LF_ERR_NOMEM;
    hipEvent_t ev[6];
    for (int i = 0; i < 6; i++) { ev[i] = (hipEvent_t)lfg_lane_event(dv, 34 + i); if (!ev[i]) return LF_ERR_HIP; }

    HIPCHK(hipMemsetAsync(d_counters, 0, 40, s));          /* [0..3] SURVEY 8(d) counters, [4] "a lower-case base somewhere in the batch" */
    HIPCHK(hipMemcpyAsync(d_off, off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s));
    const uint64_t qw = lf_plane_words(n_bases);
    uint64_t *d_planes = DSLOT(uint64_t, 14, 3 * qw * 8);
    if (!d_planes) return LF_ERR_NOMEM;
    /* one turn per DEVICE (a host link each): lanes of different devices in one process (lf_map_batch_multi) do not wait for each other.
     * Before a lane asks for its turn it lets its own stream drain: the stream may still be waiting for the lane's previous chunk's SAM
     * writer (which reads the buffers the upload overwrites), and that wait must not be spent holding the turn. */
    static std::mutex upload_turns[64];
    std::mutex &upload_turn = upload_turns[dv & 63];
    const bool turns = true;       /* (the lanes' uploads take turns: 99.4 against 106.6 ms per step without, profiles/r04_pcie/) */
    if (pk) {
        /* the batch arrives as bit planes (3 / 8 of the bytes): they are what the alignment kernels want anyway; the seed search
         * and the SAM writer get the bytes back from them */
        if (pk->qw != qw) { lf_set_error("lfg_seed_packed: %llu plane words for %llu bases", (unsigned long long)pk->qw, (unsigned long long)n_bases); return LF_ERR_ARG; }
        uint64_t *d_xpos = DSLOT(uint64_t, 2, pk->n_exc * 8 + 16); uint8_t *d_xbyte = DSLOT(uint8_t, 8, pk->n_exc + 16);
        if (!d_xpos || !d_xbyte) return LF_ERR_NOMEM;
        {
            if (turns) HIPCHK(hipStreamSynchronize(s));
            std::unique_lock<std::mutex> g(upload_turn, std::defer_lock);
            if (turns) g.lock();
            if (pk->src_qw) {
                /* prepacked batch: the three word ranges that hold the chunk's bits, then a shift on the device */
                const uint64_t qs = (n_bases + pk->shift + 63) / 64 + 1;
                uint64_t *d_stage = DSLOT(uint64_t, 15, 3 * qs * 8 + 64);
                if (!d_stage) return LF_ERR_NOMEM;
                for (int x = 0; x < 3; x++) HIPCHK(hipMemcpyAsync(d_stage + (size_t)x * qs, pk->planes + (size_t)x * pk->src_qw + pk->word0, qs * 8, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(lf_planes_shift_kernel, dim3((unsigned)((3 * qw + 255) / 256)), dim3(256), 0, s, (const uint64_t *)d_stage, qs, pk->shift, n_bases, d_planes, qw);
            } else
            HIPCHK(hipMemcpyAsync(d_planes, pk->planes, 3 * qw * 8, hipMemcpyHostToDevice, s));
            if (pk->n_exc) {
                HIPCHK(hipMemcpyAsync(d_xpos, pk->exc_pos, pk->n_exc * 8, hipMemcpyHostToDevice, s));
                HIPCHK(hipMemcpyAsync(d_xbyte, pk->exc_byte, pk->n_exc, hipMemcpyHostToDevice, s));
            }
            if (turns) HIPCHK(hipStreamSynchronize(s));
        }
        {   /* seeding is case-insensitive (nst_nt4_table), the planes are not: a lower-case base among the exceptions selects the byte variant of the search */
            bool lower = false;
            for (uint64_t i = 0; i < pk->n_exc && !lower; i++) { const uint8_t b = pk->exc_byte[i]; lower = b == 'a' || b == 'c' || b == 'g' || b == 't'; }
            if (lower) HIPCHK(hipMemsetAsync(d_counters + 4, 0xff, 8, s));
        }
        hipLaunchKernelGGL(lf_unpack_planes_kernel, dim3((unsigned)((n_bases / 4 + 256) / 256)), dim3(256), 0, s, d_planes, d_planes + qw, d_planes + 2 * qw, n_bases, d_reads);
        if (pk->n_exc) hipLaunchKernelGGL(lf_patch_bytes_kernel, dim3((unsigned)((pk->n_exc + 255) / 256)), dim3(256), 0, s, (const uint64_t *)d_xpos, (const uint8_t *)d_xbyte, pk->n_exc, pk->src_qw ? pk->exc_base : 0ull, d_reads);
    } else if (reads) {
        /* Uploads take turns: eight lanes start a step together, and eight concurrent copies share the link -- every lane would
         * get its bases after ~8 x the time of one copy.  In turn, the first lane's kernels start after one copy and the other
         * lanes' copies run under them.  (One stream wait per chunk; LF_UPLOAD_TURNS=0 switches it off for A / B runs.) */
        if (turns) {
            HIPCHK(hipStreamSynchronize(s));
            std::lock_guard<std::mutex> g(upload_turn);
            HIPCHK(hipMemcpyAsync(d_reads, reads, n_bases, hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        } else HIPCHK(hipMemcpyAsync(d_reads, reads, n_bases, hipMemcpyHostToDevice, s));
    } else {
        if (!d_src || !src_off) { lf_set_error("lfg_seed: no read bases"); return LF_ERR_ARG; }
        uint64_t *d_src_off = DSLOT(uint64_t, 13, (size_t)(n_reads + 1) * 8);       /* kept for the SAM writer's qualities (same layout) */
        if (!d_src_off) return LF_ERR_NOMEM;
        HIPCHK(hipMemcpyAsync(d_src_off, src_off, (size_t)n_reads * 8, hipMemcpyHostToDevice, s));
        const int grc = lfg_gather_reads(dv, (void *)s, d_src, d_src_off, d_off, n_reads, d_reads);
        if (grc != LF_OK) return grc;
    }

    {   /* the batch as three bit planes (code low / high bit, "is ACGT"): what the alignment kernels build their blocks from */
        if (!pk) lf_rsweep_pack_planes(s, (const unsigned char *)d_reads, n_bases, d_planes, qw, d_counters + 4);
        lfg_lane_set_value(dv, 0, qw);
    }
    hipLaunchKernelGGL(lf_seed_pos_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, n_reads, d_off, hc, d_pos2);
    HIPCHK(hipEventRecord(ev[0], s));
    {
        /* a wavefront owns the samples of rpw whole reads; both variants are launched and the batch's flag (a lower-case base somewhere:
         * seeding is case-insensitive, the bit planes are not) decides which of them works */
        const uint32_t rpw = hc >= LF_SEARCH_MIN_SPAN ? 1u : (LF_SEARCH_MIN_SPAN + hc - 1) / hc;
        const unsigned waves = (unsigned)(((uint32_t)n_reads + rpw - 1) / rpw), blocks = (waves + 3) / 4;
        /* one search per lane (two / three per lane measured equal, 9.96 / 10.5 against 10.2 ms: a slot costs as many registers as a wavefront does) */
#define LF_SEARCH_LAUNCH(SS, PL) hipLaunchKernelGGL((lf_seed_search_kernel<SS, PL>), dim3(blocks), dim3(256), 0, s, st->view, n_reads, d_reads, (const uint64_t *)d_planes, (int64_t)qw, \
                                                     d_off, hc, rpw, p->min_anchor_len, d_pos2, d_smp, d_counters)
        LF_SEARCH_LAUNCH(1, true); LF_SEARCH_LAUNCH(1, false);
#undef LF_SEARCH_LAUNCH
    }
    HIPCHK(hipEventRecord(ev[1], s));
    hipLaunchKernelGGL(lf_seed_accept_kernel, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, s, n_reads, hc, (uint32_t)p->max_ref_hits, d_pos2, d_smp, d_cnt);
    HIPCHK(hipMemsetAsync(d_cnt + total, 0, 4, s));
    /* u32 counts summed into u64 offsets: one launch (lf_scan.h) */
    { lf_scan_u32 f; f.p = d_cnt; const int src = lf_scan_excl(dv, 0, s, f, d_hit_off, total + 1); if (src != LF_OK) return src; }
    HIPCHK(hipEventRecord(ev[2], s));
    uint64_t *h_nhits = (uint64_t *)lfg_pin_slot(LF_PS_HITS_OFF, (size_t)(n_reads + 2 + 8) * 8);
    if (!h_nhits) return LF_ERR_NOMEM;
    HIPCHK(hipMemcpyAsync(h_nhits, d_hit_off + total, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const uint64_t n_hits = h_nhits[0];

    uint32_t *d_tpos = DSLOT(uint32_t, 9, (n_hits + 1) * 4);
    uint32_t *d_qpl = DSLOT(uint32_t, 10, (n_hits + 1) * 4);
    uint8_t *d_strand = DSLOT(uint8_t, 11, n_hits + 16);
    if (!d_tpos || !d_qpl || !d_strand) return LF_ERR_NOMEM;
    HIPCHK(hipEventRecord(ev[3], s));
    hipLaunchKernelGGL(lf_seed_locate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, st->view, n_reads, d_off, hc,
                       d_pos2, d_smp, d_cnt, d_hit_off, d_tpos, d_qpl, d_strand, d_counters);
    HIPCHK(hipEventRecord(ev[4], s));
    hipLaunchKernelGGL(lf_read_first_hit_kernel, dim3((n_reads + 1 + 255) / 256), dim3(256), 0, s, n_reads, hc, d_hit_off, n_hits, d_read_off);

    /* results land in pinned host slots (valid until the next lfg_seed call) */
    out->n_hits = n_hits;
    out->read_off = h_nhits;
    if (want_hits) {
        out->tpos = (uint32_t *)lfg_pin_slot(LF_PS_HITS_T, (n_hits + 1) * 4);
        out->qpl = (uint32_t *)lfg_pin_slot(LF_PS_HITS_Q, (n_hits + 1) * 4);
        out->strand = (uint8_t *)lfg_pin_slot(LF_PS_HITS_S, n_hits + 1);
        if (!out->tpos || !out->qpl || !out->strand) return LF_ERR_NOMEM;
        HIPCHK(hipMemcpyAsync(out->tpos, d_tpos, n_hits * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(out->qpl, d_qpl, n_hits * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(out->strand, d_strand, n_hits, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipMemcpyAsync(out->read_off, d_read_off, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, s));
    uint64_t *h_cnt = h_nhits + (n_reads + 2);             /* pinned (a copy into the caller's struct would make the runtime wait for the stream inside the call) */
    HIPCHK(hipMemcpyAsync(h_cnt, d_counters, 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int k = 0; k < 4; k++) out->counters[k] = h_cnt[k];
    HIPCHK(hipGetLastError());
    out->counters[3] = n_bases;
    out->counters[2] = n_hits;                 /* N_sa: one suffix-array read per hit */
    HIPCHK(hipEventElapsedTime(&out->ms_search, ev[0], ev[1]));
    HIPCHK(hipEventElapsedTime(&out->ms_accept, ev[1], ev[2]));
    HIPCHK(hipEventElapsedTime(&out->ms_locate, ev[3], ev[4]));
#undef DSLOT
    return LF_OK;
}

extern "C" void lfg_hits_free(lfg_hits_t *h)
{
    memset(h, 0, sizeof(*h));      /* the arrays live in persistent pinned slots */
}
