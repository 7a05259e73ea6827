/*
 * lf_indexer.hip -- GPU construction of the reference's index files, byte for byte:
 *   <fa>.pac .ann .amb   (lib/bwa/bntseq.c:224-322: forward-only 2-bit pac, N -> lrand48()&3 with srand48(11))
 *   <fa>.bwt             (lib/bwa/bwtindex.c:128-150 Occ interleave; lib/bwa/bwt.c:385-394 dump)
 *   <fa>.sa              (lib/bwa/bwt.c:62-84 sampling every 32nd ROW; :396-407 dump)
 *   <fa>.cache           (src/BWT.cpp:60-138: SA interval of every 12-mer)
 * The BWT of a text is unique, so any suffix sorter yields the same files as bwa's IS / BWT-SW builders.
 *
 * Suffix sorting on the GPU (text = forward + reverse complement, 2 bits/base in HBM):
 *   suffixes are bucketed by their first KB bases; each bucket is radix-sorted (hipCUB) on a 64-bit key =
 *   29 bases (58 bits) + min(remaining,29) (6 bits; a suffix that hits the end sorts before its
 *   extensions because '$' is the smallest symbol); groups that still tie are refined by the next 29
 *   bases at a time (MSD refinement on the compacted tie set) until every group is a singleton.
 * Then one gather pass produces the BWT with its interleaved Occ counters, the sampled SA and the 12-mer
 * table.  The 288 GB of HBM hold text, the full SA (8 B/row) and the sort buffers of one bucket at once.
 */
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <string>
#include <vector>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "lf_gpu_common.h"

#define DEPTH_BASES 29

struct lf_text { const uint64_t *w; uint64_t n; };      /* 32 bases per word, MSB first */

__device__ __forceinline__ uint64_t lf_key29(const lf_text T, uint64_t i)
{
    if (i >= T.n) return 0;
    const uint64_t wi = i >> 5; const int sh = (int)(i & 31) * 2;
    uint64_t x = T.w[wi] << sh;
    if (sh) x |= T.w[wi + 1] >> (64 - sh);
    const uint64_t rem = T.n - i;
    uint64_t bases = x >> 6;                                      /* top 58 bits = 29 bases */
    uint64_t len = DEPTH_BASES;
    if (rem < DEPTH_BASES) { len = rem; bases &= ~0ull << (2 * (DEPTH_BASES - (int)rem)); }
    return (bases << 6) | len;
}
__device__ __forceinline__ int lf_text_at(const lf_text T, uint64_t i) { return (int)((T.w[i >> 5] >> ((~i & 31) << 1)) & 3); }

/* text = forward codes followed by their reverse complement */
__global__ void lf_pack_text_kernel(const uint8_t *__restrict__ fwd, uint64_t l_pac, uint64_t *__restrict__ w, uint64_t n_words)
{
    const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    uint64_t x = 0;
    for (int k = 0; k < 32; k++) {
        const uint64_t i = wi * 32 + k;
        int c = 0;
        if (i < l_pac) c = fwd[i];
        else if (i < 2 * l_pac) c = 3 - fwd[2 * l_pac - 1 - i];
        x |= (uint64_t)c << ((31 - k) * 2);
    }
    w[wi] = x;
}

__device__ __forceinline__ uint32_t lf_bucket_of(const lf_text T, uint64_t i, int kb)
{
    uint32_t b = 0;
    for (int k = 0; k < kb; k++) b = b * 4 + (i + k < T.n ? (uint32_t)lf_text_at(T, i + k) : 0u);
    return b;
}
__global__ void lf_bucket_hist_kernel(lf_text T, int kb, unsigned long long *__restrict__ hist)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&hist[lf_bucket_of(T, i, kb)], 1ull);
}
struct lf_in_bucket {
    lf_text T; int kb; uint32_t b; uint64_t base;
    __device__ bool operator()(uint64_t off) const { return lf_bucket_of(T, base + off, kb) == b; }
};
__global__ void lf_add_base_kernel(uint64_t *p, uint64_t n, uint64_t base) { const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += base; }
__global__ void lf_keys_kernel(lf_text T, const uint64_t *__restrict__ pos, uint64_t n, uint64_t depth_off, uint64_t *__restrict__ keys)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = lf_key29(T, pos[i] + depth_off);
}
__global__ void lf_iota_kernel(uint32_t *p, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i; }

/* head flags / group ids on a sorted run: head[j] = first of its (gid,key) group; gidhead[j] = slot[j] at heads else 0 */
__global__ void lf_heads_kernel(const uint64_t *__restrict__ key, const uint32_t *__restrict__ gid, const uint32_t *__restrict__ slot,
                                uint32_t n, uint8_t *__restrict__ head, uint32_t *__restrict__ gidhead)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const bool h = (j == 0) || key[j] != key[j - 1] || (gid && gid[j] != gid[j - 1]);
    head[j] = h;
    gidhead[j] = h ? (slot ? slot[j] : j) : 0u;
}
__global__ void lf_tied_flag_kernel(const uint8_t *__restrict__ head, uint32_t n, uint8_t *__restrict__ tied)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    tied[j] = !(head[j] && (j + 1 == n || head[j + 1]));
}
template <class T> __global__ void lf_gather_kernel(const T *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, T *__restrict__ dst)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = src[idx[j]];
}
__global__ void lf_scatter_sa_kernel(uint64_t *__restrict__ sa_seg, const uint32_t *__restrict__ slot, const uint64_t *__restrict__ pos, uint32_t n)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) sa_seg[slot[j]] = pos[j];
}
struct lf_max_op { __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

/* BWT block j (128 symbols) from the SA: B[x] = T[SA[row]-1], row = x + (x >= primary) (rows 0..n, row 0 = '$' suffix) */
__global__ void lf_bwt_block_kernel(lf_text T, const uint64_t *__restrict__ sa, uint64_t primary, uint64_t n_blocks,
                                    uint32_t *__restrict__ bwt_out, uint32_t *__restrict__ cnt)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_blocks) return;
    uint32_t c4[4] = { 0, 0, 0, 0 };
    uint32_t *out = bwt_out + j * 16 + 8;                     /* final interleaved layout: 8 words of counts first */
    for (int wi = 0; wi < 8; wi++) {
        uint32_t word = 0;
        for (int k = 0; k < 16; k++) {
            const uint64_t x = j * 128 + (uint64_t)wi * 16 + k;
            if (x < T.n) {
                const uint64_t row = x + (x >= primary);
                const uint64_t s = sa[row];
                const int c = lf_text_at(T, s - 1);           /* s != 0 here: the primary row is skipped */
                c4[c]++;
                word |= (uint32_t)c << ((15 - k) * 2);
            }
        }
        if (j * 128 + (uint64_t)wi * 16 < T.n) out[wi] = word;
    }
    for (int c = 0; c < 4; c++) cnt[(size_t)c * n_blocks + j] = c4[c];
}
__global__ void lf_occ_header_kernel(const uint64_t *__restrict__ pre /* 4 x n_blocks exclusive */, uint64_t n_blocks, uint64_t n_occ,
                                     const uint64_t *__restrict__ totals, uint32_t *__restrict__ bwt_out, uint64_t last_off_words)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_blocks) {
        uint64_t *hdr = reinterpret_cast<uint64_t *>(bwt_out + j * 16);
        for (int c = 0; c < 4; c++) hdr[c] = pre[(size_t)c * n_blocks + j];
    }
    if (j == 0) {            /* trailing record with the totals (lib/bwa/bwtindex.c:145-146) */
        uint32_t *p = bwt_out + last_off_words;
        for (int c = 0; c < 4; c++) { p[2 * c] = (uint32_t)totals[c]; p[2 * c + 1] = (uint32_t)(totals[c] >> 32); }
    }
    (void)n_occ;
}
__global__ void lf_find_primary_kernel(const uint64_t *__restrict__ sa, uint64_t rows, unsigned long long *__restrict__ primary)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x)
        if (sa[r] == 0) *primary = r;
}
__global__ void lf_sample_sa_kernel(const uint64_t *__restrict__ sa, uint64_t n_sa, uint64_t *__restrict__ out)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_sa) out[j] = sa[j * 32];
}
__global__ void lf_count_bases_kernel(lf_text T, unsigned long long *__restrict__ cnt)
{
    unsigned long long c4[4] = { 0, 0, 0, 0 };
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.n; i += (uint64_t)gridDim.x * blockDim.x) c4[lf_text_at(T, i)]++;
    for (int c = 0; c < 4; c++) if (c4[c]) atomicAdd(&cnt[c], c4[c]);
}

/* ---------------------------------------------------------------- host: FASTA -> pac/ann/amb */
struct fa_seq { std::string name, comment; uint64_t offset; uint32_t len; int n_ambs; };
struct fa_hole { uint64_t offset; uint32_t len; char amb; };

static int nt4_host(int ch)
{
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

/* glibc lrand48 after srand48(seed): X = seed<<16 | 0x330E; X = X*0x5DEECE66D + 0xB mod 2^48; value = X >> 17 */
struct rand48 { uint64_t x; void seed(uint32_t s) { x = ((uint64_t)s << 16) | 0x330E; } long next() { x = (x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL; return (long)(x >> 17); } };

static int read_fasta(const char *path, std::vector<fa_seq> &seqs, std::vector<fa_hole> &holes, std::vector<uint8_t> &codes)
{
    gzFile fp = gzopen(path, "r");
    if (!fp) { lf_set_error("cannot open %s", path); return LF_ERR_IO; }
    rand48 rng; rng.seed(11);                                    /* bns->seed = 11, lib/bwa/bntseq.c:290-291 */
    std::vector<char> buf(1 << 22);
    std::string line;
    int lasts = 0; bool in_header = false, have = false;
    std::string header;
    auto start_seq = [&](const std::string &h) {
        fa_seq s; size_t k = 0;
        while (k < h.size() && !isspace((unsigned char)h[k])) k++;
        s.name = h.substr(0, k);
        s.comment = k < h.size() ? h.substr(k + 1) : std::string();
        while (!s.comment.empty() && (s.comment.back() == '\r')) s.comment.pop_back();
        s.offset = codes.size(); s.len = 0; s.n_ambs = 0;
        seqs.push_back(s); lasts = 0; have = true;
    };
    int n;
    while ((n = gzread(fp, buf.data(), (unsigned)buf.size())) > 0) {
        for (int i = 0; i < n; i++) {
            const char ch = buf[i];
            if (in_header) {
                if (ch == '\n') { start_seq(header); header.clear(); in_header = false; }
                else header.push_back(ch);
                continue;
            }
            if (ch == '>') { in_header = true; continue; }
            if (!have || !isgraph((unsigned char)ch)) continue;
            fa_seq &s = seqs.back();
            int c = nt4_host(ch);
            if (c >= 4) {                                          /* add1, lib/bwa/bntseq.c:241-258 */
                if (lasts == ch) holes.back().len++;
                else { fa_hole h; h.offset = s.offset + s.len; h.len = 1; h.amb = ch; holes.push_back(h); s.n_ambs++; }
                c = (int)(rng.next() & 3);
            }
            lasts = ch;
            codes.push_back((uint8_t)c);
            s.len++;
        }
    }
    if (in_header) start_seq(header);
    gzclose(fp);
    if (seqs.empty() || codes.empty()) { lf_set_error("%s holds no sequence", path); return LF_ERR_IO; }
    return LF_OK;
}

static int write_file(const std::string &path, const void *a, size_t na, const void *b = nullptr, size_t nb = 0)
{
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { lf_set_error("cannot write %s", path.c_str()); return LF_ERR_IO; }
    size_t ok = fwrite(a, 1, na, fp) == na;
    if (b && ok) ok = fwrite(b, 1, nb, fp) == nb;
    fclose(fp);
    if (!ok) { lf_set_error("short write on %s", path.c_str()); return LF_ERR_IO; }
    return LF_OK;
}

#define LAUNCH1D(kern, count, stream, ...) hipLaunchKernelGGL(kern, dim3((unsigned)(((count) + 255) / 256)), dim3(256), 0, stream, __VA_ARGS__)

extern "C" int lf_index_build(const char *fasta_path, int device)
{
    if (lfg_device_count() <= device) { lf_set_error("no gfx950 device %d visible (the indexer runs on the GPU)", device); return LF_ERR_NO_DEVICE; }
    std::vector<fa_seq> seqs; std::vector<fa_hole> holes; std::vector<uint8_t> codes;
    int rc = read_fasta(fasta_path, seqs, holes, codes);
    if (rc != LF_OK) return rc;
    const std::string prefix(fasta_path);
    const uint64_t l_pac = codes.size(), n = 2 * l_pac;
    if (l_pac >= 0xFFFFFFFFull) { lf_set_error("reference longer than 2^32-1 bases is not supported (Seed_t.tPos is 32 bit)"); return LF_ERR_ARG; }

    /* ---- .pac / .ann / .amb (forward only; lib/bwa/bntseq.c:309-322, 71-95) ---- */
    {
        std::vector<uint8_t> pac((size_t)(l_pac / 4 + 2), 0);
        for (uint64_t l = 0; l < l_pac; l++) pac[l >> 2] |= (uint8_t)(codes[l] << ((~l & 3) << 1));
        size_t nbytes = (size_t)(l_pac >> 2) + ((l_pac & 3) ? 1 : 0);
        std::vector<uint8_t> tail;
        if (l_pac % 4 == 0) tail.push_back(0);
        tail.push_back((uint8_t)(l_pac % 4));
        if ((rc = write_file(prefix + ".pac", pac.data(), nbytes, tail.data(), tail.size())) != LF_OK) return rc;
        std::string ann, amb; char tmp[256];
        snprintf(tmp, sizeof tmp, "%lld %d %u\n", (long long)l_pac, (int)seqs.size(), 11u); ann += tmp;
        for (auto &s : seqs) {
            ann += "0 " + s.name;
            ann += " " + (s.comment.empty() ? std::string("(null)") : s.comment) + "\n";
            snprintf(tmp, sizeof tmp, "%lld %d %d\n", (long long)s.offset, (int)s.len, s.n_ambs); ann += tmp;
        }
        snprintf(tmp, sizeof tmp, "%lld %d %u\n", (long long)l_pac, (int)seqs.size(), (unsigned)holes.size()); amb += tmp;
        for (auto &h : holes) { snprintf(tmp, sizeof tmp, "%lld %d %c\n", (long long)h.offset, (int)h.len, h.amb); amb += tmp; }
        if ((rc = write_file(prefix + ".ann", ann.data(), ann.size())) != LF_OK) return rc;
        if ((rc = write_file(prefix + ".amb", amb.data(), amb.size())) != LF_OK) return rc;
    }

    HIPCHK(hipSetDevice(device));
    lfg_quiesce(device);                     /* hipMalloc / hipFree below synchronise the device: see lf_mem.hip */
    hipStream_t s; HIPCHK(hipStreamCreate(&s));
    /* ---- text in HBM ---- */
    uint8_t *d_fwd; uint64_t *d_w;
    const uint64_t n_words = (n + 31) / 32 + 2;
    HIPCHK(hipMalloc(&d_fwd, l_pac + 16));
    HIPCHK(hipMemcpy(d_fwd, codes.data(), l_pac, hipMemcpyHostToDevice));
    std::vector<uint8_t>().swap(codes);
    HIPCHK(hipMalloc(&d_w, n_words * 8));
    LAUNCH1D(lf_pack_text_kernel, n_words, s, d_fwd, l_pac, d_w, n_words);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(d_fwd));
    lf_text T; T.w = d_w; T.n = n;

    /* ---- buckets ---- */
    const uint64_t CAP = 192ull << 20;
    int kb = 0;
    while ((n >> (2 * kb)) > CAP / 3 && kb < 8) kb++;            /* average bucket <= CAP/3: head-room for skew */
    const uint32_t n_buckets = 1u << (2 * kb);
    unsigned long long *d_hist;
    HIPCHK(hipMalloc(&d_hist, (size_t)n_buckets * 8));
    HIPCHK(hipMemset(d_hist, 0, (size_t)n_buckets * 8));
    hipLaunchKernelGGL(lf_bucket_hist_kernel, dim3(4096), dim3(256), 0, s, T, kb, d_hist);
    std::vector<unsigned long long> hist(n_buckets);
    HIPCHK(hipMemcpyAsync(hist.data(), d_hist, (size_t)n_buckets * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint64_t maxb = 0;
    for (auto v : hist) maxb = std::max<uint64_t>(maxb, v);
    if (maxb >= (1ull << 31)) { lf_set_error("suffix bucket of %llu entries exceeds the sorter's 2^31 limit", (unsigned long long)maxb); return LF_ERR_NOMEM; }

    uint64_t *d_sa;                                               /* rows 0..n */
    HIPCHK(hipMalloc(&d_sa, (n + 2) * 8));
    { const uint64_t first = n; HIPCHK(hipMemcpy(d_sa, &first, 8, hipMemcpyHostToDevice)); }

    /* per-bucket work buffers */
    const size_t M = (size_t)maxb + 16;
    uint64_t *d_pos, *d_pos2, *d_key, *d_key2; uint32_t *d_idx, *d_idx2, *d_gid, *d_gid2, *d_slot, *d_slot2, *d_u32a; uint8_t *d_head, *d_tied;
    unsigned long long *d_nsel;
    HIPCHK(hipMalloc(&d_pos, M * 8)); HIPCHK(hipMalloc(&d_pos2, M * 8)); HIPCHK(hipMalloc(&d_key, M * 8)); HIPCHK(hipMalloc(&d_key2, M * 8));
    HIPCHK(hipMalloc(&d_idx, M * 4)); HIPCHK(hipMalloc(&d_idx2, M * 4)); HIPCHK(hipMalloc(&d_gid, M * 4)); HIPCHK(hipMalloc(&d_gid2, M * 4));
    HIPCHK(hipMalloc(&d_slot, M * 4)); HIPCHK(hipMalloc(&d_slot2, M * 4)); HIPCHK(hipMalloc(&d_u32a, M * 4));
    HIPCHK(hipMalloc(&d_head, M)); HIPCHK(hipMalloc(&d_tied, M)); HIPCHK(hipMalloc(&d_nsel, 8));
    size_t tmp_bytes = 0; void *d_tmp = nullptr;
    {   /* one temp buffer big enough for every hipCUB call below */
        size_t a = 0, b = 0, c = 0, d = 0, e = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, d_key, d_key2, d_pos, d_pos2, (int)maxb, 0, 64, s);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, d_key, d_key2, d_idx, d_idx2, (int)maxb, 0, 64, s);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, c, d_gid, d_gid2, d_idx, d_idx2, (int)maxb, 0, 32, s);
        (void)hipcub::DeviceScan::InclusiveScan(nullptr, d, d_u32a, d_gid, lf_max_op(), (int)maxb, s);
        (void)hipcub::DeviceSelect::Flagged(nullptr, e, d_pos, d_tied, d_pos2, d_nsel, (int)maxb, s);
        tmp_bytes = std::max({ a, b, c, d, e }) + (64u << 20);
        HIPCHK(hipMalloc(&d_tmp, tmp_bytes));
    }

    uint64_t row = 1;
    for (uint32_t b = 0; b < n_buckets; b++) {
        const uint64_t nb = hist[b];
        if (nb == 0) continue;
        /* positions of this bucket, ascending (chunked select over a counting iterator) */
        uint64_t got = 0;
        for (uint64_t base = 0; base < n; base += (1ull << 30)) {
            const uint64_t len = std::min<uint64_t>(1ull << 30, n - base);
            lf_in_bucket pred; pred.T = T; pred.kb = kb; pred.b = b; pred.base = base;
            size_t tb = tmp_bytes;
            HIPCHK(hipcub::DeviceSelect::If(d_tmp, tb, hipcub::CountingInputIterator<uint64_t>(0), d_pos + got, d_nsel, (int)len, pred, s));
            unsigned long long k = 0;
            HIPCHK(hipMemcpyAsync(&k, d_nsel, 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (k && base) LAUNCH1D(lf_add_base_kernel, k, s, d_pos + got, (uint64_t)k, base);
            got += k;
        }
        if (got != nb) { lf_set_error("bucket %u: selected %llu of %llu suffixes", b, (unsigned long long)got, (unsigned long long)nb); return LF_ERR_HIP; }
        uint64_t *seg = d_sa + row;
        LAUNCH1D(lf_keys_kernel, nb, s, T, d_pos, nb, (uint64_t)0, d_key);
        { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_key, d_key2, d_pos, seg, (int)nb, 0, 64, s)); }
        /* tie groups: members of runs of equal keys */
        LAUNCH1D(lf_heads_kernel, nb, s, d_key2, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t)nb, d_head, d_u32a);
        { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceScan::InclusiveScan(d_tmp, tb, d_u32a, d_gid2, lf_max_op(), (int)nb, s)); }
        LAUNCH1D(lf_tied_flag_kernel, nb, s, d_head, (uint32_t)nb, d_tied);
        LAUNCH1D(lf_iota_kernel, nb, s, d_idx, (uint32_t)nb);
        unsigned long long m = 0;
        { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceSelect::Flagged(d_tmp, tb, d_idx, d_tied, d_slot, d_nsel, (int)nb, s)); }
        HIPCHK(hipMemcpyAsync(&m, d_nsel, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (m) {
            LAUNCH1D(lf_gather_kernel<uint32_t>, m, s, d_gid2, d_slot, (uint32_t)m, d_gid);        /* group id = slot of the run head */
            LAUNCH1D(lf_gather_kernel<uint64_t>, m, s, (const uint64_t *)seg, d_slot, (uint32_t)m, d_pos);
        }
        for (uint64_t depth = 1; m > 0; depth++) {
            const uint32_t mm = (uint32_t)m;
            LAUNCH1D(lf_keys_kernel, mm, s, T, d_pos, (uint64_t)mm, depth * DEPTH_BASES, d_key);
            LAUNCH1D(lf_iota_kernel, mm, s, d_idx, mm);
            { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_key, d_key2, d_idx, d_idx2, (int)mm, 0, 64, s)); }
            LAUNCH1D(lf_gather_kernel<uint32_t>, mm, s, d_gid, d_idx2, mm, d_gid2);
            { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_gid2, d_u32a, d_idx2, d_idx, (int)mm, 0, 32, s)); }
            /* d_u32a = sorted gids, d_idx = permutation: element j of the refined order is old element d_idx[j] */
            LAUNCH1D(lf_gather_kernel<uint64_t>, mm, s, d_key, d_idx, mm, d_key2);
            LAUNCH1D(lf_gather_kernel<uint64_t>, mm, s, d_pos, d_idx, mm, d_pos2);
            LAUNCH1D(lf_scatter_sa_kernel, mm, s, seg, d_slot, d_pos2, mm);                        /* slots stay ascending */
            LAUNCH1D(lf_heads_kernel, mm, s, d_key2, d_u32a, d_slot, mm, d_head, d_gid2);
            { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceScan::InclusiveScan(d_tmp, tb, d_gid2, d_gid, lf_max_op(), (int)mm, s)); }
            LAUNCH1D(lf_tied_flag_kernel, mm, s, d_head, mm, d_tied);
            /* compact (pos, gid, slot) of the still-tied members */
            { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceSelect::Flagged(d_tmp, tb, d_pos2, d_tied, d_pos, d_nsel, (int)mm, s)); }
            { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceSelect::Flagged(d_tmp, tb, d_gid, d_tied, d_gid2, d_nsel, (int)mm, s)); }
            { size_t tb = tmp_bytes; HIPCHK(hipcub::DeviceSelect::Flagged(d_tmp, tb, d_slot, d_tied, d_slot2, d_nsel, (int)mm, s)); }
            HIPCHK(hipMemcpyAsync(&m, d_nsel, 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            std::swap(d_gid, d_gid2); std::swap(d_slot, d_slot2);
            if (depth > (n / DEPTH_BASES) + 2) { lf_set_error("suffix refinement did not converge"); return LF_ERR_HIP; }
        }
        row += nb;
    }
    if (row != n + 1) { lf_set_error("suffix array has %llu rows, expected %llu", (unsigned long long)row, (unsigned long long)(n + 1)); return LF_ERR_HIP; }
    (void)hipFree(d_pos); (void)hipFree(d_pos2); (void)hipFree(d_key); (void)hipFree(d_key2); (void)hipFree(d_idx); (void)hipFree(d_idx2);
    (void)hipFree(d_gid); (void)hipFree(d_gid2); (void)hipFree(d_slot); (void)hipFree(d_slot2); (void)hipFree(d_u32a); (void)hipFree(d_head); (void)hipFree(d_tied);

    /* ---- primary, L2, BWT + Occ, sampled SA ---- */
    unsigned long long *d_primary, *d_cnt4;
    HIPCHK(hipMalloc(&d_primary, 8)); HIPCHK(hipMalloc(&d_cnt4, 32)); HIPCHK(hipMemset(d_cnt4, 0, 32));
    hipLaunchKernelGGL(lf_find_primary_kernel, dim3(4096), dim3(256), 0, s, (const uint64_t *)d_sa, n + 1, d_primary);
    hipLaunchKernelGGL(lf_count_bases_kernel, dim3(2048), dim3(256), 0, s, T, d_cnt4);
    unsigned long long primary = 0, cnt4[4];
    HIPCHK(hipMemcpyAsync(&primary, d_primary, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(cnt4, d_cnt4, 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint64_t hdr[5]; hdr[0] = primary; hdr[1] = cnt4[0]; hdr[2] = hdr[1] + cnt4[1]; hdr[3] = hdr[2] + cnt4[2]; hdr[4] = hdr[3] + cnt4[3];

    const uint64_t n_blocks = (n + 127) / 128, n_occ = n_blocks + 1;
    const uint64_t bwt_words = (n + 15) / 16 + n_occ * 8;                      /* lib/bwa/bwtindex.c:133-134 */
    uint32_t *d_bwt, *d_cnt; uint64_t *d_pre;
    HIPCHK(hipMalloc(&d_bwt, (n_blocks + 2) * 64)); HIPCHK(hipMemset(d_bwt, 0, (n_blocks + 2) * 64));
    HIPCHK(hipMalloc(&d_cnt, n_blocks * 16)); HIPCHK(hipMalloc(&d_pre, n_blocks * 32));
    LAUNCH1D(lf_bwt_block_kernel, n_blocks, s, T, (const uint64_t *)d_sa, (uint64_t)primary, n_blocks, d_bwt, d_cnt);
    struct widen { __host__ __device__ uint64_t operator()(uint32_t x) const { return x; } };
    for (int c = 0; c < 4; c++) {
        size_t tb = tmp_bytes;
        hipcub::TransformInputIterator<uint64_t, widen, uint32_t *> in(d_cnt + (size_t)c * n_blocks, widen());
        for (uint64_t base = 0; base < n_blocks; base += (1ull << 30)) {      /* chunked: hipCUB counts are int */
            const uint64_t len = std::min<uint64_t>(1ull << 30, n_blocks - base);
            if (base) { lf_set_error("Occ scan over more than 2^30 blocks is not implemented"); return LF_ERR_ARG; }
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, in + base, d_pre + (size_t)c * n_blocks + base, (int)len, s));
        }
    }
    uint64_t *d_tot; HIPCHK(hipMalloc(&d_tot, 32));
    { uint64_t tot[4] = { cnt4[0], cnt4[1], cnt4[2], cnt4[3] }; HIPCHK(hipMemcpyAsync(d_tot, tot, 32, hipMemcpyHostToDevice, s)); HIPCHK(hipStreamSynchronize(s)); }
    /* the trailing record sits right after the last (possibly partial) block's words */
    const uint64_t last_off = bwt_words - 8;
    LAUNCH1D(lf_occ_header_kernel, n_blocks, s, (const uint64_t *)d_pre, n_blocks, n_occ, (const uint64_t *)d_tot, d_bwt, last_off);
    HIPCHK(hipStreamSynchronize(s));

    std::vector<uint32_t> h_bwt(bwt_words);
    HIPCHK(hipMemcpy(h_bwt.data(), d_bwt, bwt_words * 4, hipMemcpyDeviceToHost));
    if ((rc = write_file(prefix + ".bwt", hdr, 40, h_bwt.data(), bwt_words * 4)) != LF_OK) return rc;

    const uint64_t n_sa = (n + 32) / 32;
    uint64_t *d_ssa; HIPCHK(hipMalloc(&d_ssa, n_sa * 8));
    LAUNCH1D(lf_sample_sa_kernel, n_sa, s, (const uint64_t *)d_sa, n_sa, d_ssa);
    std::vector<uint64_t> h_sa(n_sa + 7);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(h_sa.data() + 7, d_ssa, n_sa * 8, hipMemcpyDeviceToHost));   /* [7] = sa[0], overwritten by the header below */
    {   /* header: primary, L2[1..4], sa_intv, seq_len ; body sa[1..] (lib/bwa/bwt.c:396-407) */
        uint64_t h7[7] = { primary, hdr[1], hdr[2], hdr[3], hdr[4], 32, n };
        if ((rc = write_file(prefix + ".sa", h7, 56, h_sa.data() + 8, (n_sa - 1) * 8)) != LF_OK) return rc;
    }

    /* ---- 12-mer table from the new BWT (src/BWT.cpp:60-138) ---- */
    {
        lf_dev_index v; memset(&v, 0, sizeof v);
        v.primary = primary; v.seq_len = n; v.l_pac = (int64_t)l_pac; v.bwt = d_bwt;
        v.L2[0] = 0; for (int i = 1; i <= 4; i++) v.L2[i] = hdr[i];
        const int K = 12;
        uint64_t *cur = nullptr;
        if ((rc = lfg_build_cache_table(&v, s, 12, &cur)) != LF_OK) return rc;
        std::vector<uint64_t> tab(((size_t)1 << (2 * K)) * 2);
        HIPCHK(hipMemcpy(tab.data(), cur, tab.size() * 8, hipMemcpyDeviceToHost));
        int32_t ch[2] = { K, 1 << (2 * K) };
        if ((rc = write_file(prefix + ".cache", ch, 8, tab.data(), tab.size() * 8)) != LF_OK) return rc;
        (void)hipFree(cur);
    }
    HIPCHK(hipGetLastError());
    (void)hipFree(d_w); (void)hipFree(d_hist); (void)hipFree(d_sa); (void)hipFree(d_tmp); (void)hipFree(d_nsel); (void)hipFree(d_primary);
    (void)hipFree(d_cnt4); (void)hipFree(d_bwt); (void)hipFree(d_cnt); (void)hipFree(d_pre); (void)hipFree(d_tot); (void)hipFree(d_ssa);
    (void)hipStreamDestroy(s);
    return LF_OK;
}
