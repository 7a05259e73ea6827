/*
 * lf_sam.hip -- SAM lines assembled on the device (printSamEntry, src/LordFAST.cpp:377-442).
 *
 * Everything a record line is made of is already in HBM when a chunk reaches its last stage: the read batch (SEQ), the
 * rendered CIGAR / MD text (lf_render.hip), the contig table.  The host decides WHAT is printed -- flags, MAPQ (the
 * reference's double arithmetic, :325-356), AS / NM, record order, SA:Z strings of split alignments -- as one 48-byte
 * line descriptor per record; this file turns descriptors into text:
 *
 *   lf_sam_len_kernel    one thread per line: its exact length (arithmetic only)          -> exclusive scan -> offsets
 *   lf_sam_write_kernel  one wavefront per line: coalesced copies of name / CIGAR / SEQ (reverse-complemented for
 *                        flag 16) / QUAL / MD, decimal fields by lane 0
 *
 * and one D2H copy of the chunk's text straight into the caller's output buffer replaces round 1's D2H of the CIGAR / MD
 * text plus two host passes (count, then memcpy into place) over ~25 kB per read.  Lines the device has no data for (reads
 * shorter than -l are not in the resident batch) arrive as literal text in a side blob.
 *
 * HOLES mode (reads came from host memory, the output buffer is pinned host memory): 37 % of a line is its SEQ column -- bytes
 * the caller gave us (printSamEntry prints the `query` / `query_rev` strings mapSeq was handed, src/LordFAST.cpp:377-402,
 * 501-502).  The writer leaves that column (SEQ, or SEQ \t QUAL for FASTQ) as a HOLE of known position and length,
 *   lf_sam_scatter_kernel  copies the two pieces of every line (in front of / behind the hole) from the chunk's device buffer
 *                          straight into the caller's buffer: kernel-driven stores over the host link, 16 bytes per lane,
 * and host threads fill the holes from the caller's own strings (memcpy, or an SSSE3 reverse complement) meanwhile:
 * 2.6 GB instead of 4.1 GB per 100 k reads cross the link.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include <algorithm>
#include "lf_internal.h"
#include "lf_gpu_common.h"
#include "lf_scan.h"

struct lf_sam_dev {
    const lf_samline_t *lines; int n_lines;
    const unsigned char *reads; const uint64_t *read_off;      /* resident read batch */
    const unsigned char *quals;                                /* same layout as reads, or NULL (FASTA: "*"; HOLES mode: the host has them) */
    int has_quals;                                             /* the batch has qualities (FASTQ) -- wherever they are: decides the LAYOUT of a line */
    const char *names, *blob;                                  /* read names; side blob: SA:Z strings and literal lines */
    const char *text; const uint64_t *toffs; const uint32_t *tlens;     /* rendered CIGAR / MD: 2 per record (lengths incl. NUL) */
    const char *ctg_names; const uint32_t *ctg_name_off;       /* contig names, offsets (n_ctg + 1) */
    const char *rg; uint32_t rg_len;                           /* "\tRG:Z:<id>" or empty */
    int holes;                                                 /* the SEQ (FASTQ: SEQ \t QUAL) column is left out: the host fills it */
};

__device__ __forceinline__ uint32_t lf_ndig(uint32_t v)
{
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u
         : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ uint32_t lf_ndig_i(int32_t v) { return v < 0 ? 1u + lf_ndig((uint32_t)(-(int64_t)v)) : lf_ndig((uint32_t)v); }

__device__ __forceinline__ uint64_t lf_sam_line_len(const lf_sam_dev &D, const lf_samline_t &Ln)
{
    if (Ln.kind == LF_SL_LITERAL) return Ln.sa_len;
    const uint32_t L = (uint32_t)(D.read_off[Ln.read + 1] - D.read_off[Ln.read]);
    const uint32_t ql = D.has_quals && Ln.is_fq ? L : 1u;
    uint64_t n = Ln.name_len + 1;
    if (Ln.kind == LF_SL_UNMAPPED) return n + 16 /* "4\t*\t0\t0\t*\t*\t0\t0\t" */ + L + 1 + ql + D.rg_len + 1;
    n += lf_ndig(Ln.flag) + 1;
    n += (D.ctg_name_off[Ln.rname + 1] - D.ctg_name_off[Ln.rname]) + 1;
    n += lf_ndig(Ln.pos1) + 1 + lf_ndig((uint32_t)Ln.mapq) + 1;
    n += (D.tlens[2 * (size_t)Ln.rec] - 1);                        /* CIGAR */
    n += 7;                                                        /* "\t*\t0\t0\t" */
    n += (uint64_t)L + 1 + ql;
    n += 6 + lf_ndig_i(Ln.as) + 13 + lf_ndig((uint32_t)Ln.nm) + 6 + (D.tlens[2 * (size_t)Ln.rec + 1] - 1);   /* \tAS:i: | \tXS:i:0\tNM:i: | \tMD:Z: */
    n += D.rg_len;
    if (Ln.sa_len) n += 6 + Ln.sa_len;                             /* \tSA:Z: */
    return n + 1;
}

/* where the SEQ column starts inside the line, and how long the hole is (SEQ, or SEQ \t QUAL when the read has qualities) */
__device__ __forceinline__ void lf_sam_hole(const lf_sam_dev &D, const lf_samline_t &Ln, uint32_t &hoff, uint32_t &hlen)
{
    hoff = 0; hlen = 0;
    if (!D.holes || Ln.kind == LF_SL_LITERAL) return;
    const uint32_t L = (uint32_t)(D.read_off[Ln.read + 1] - D.read_off[Ln.read]);
    hlen = (D.has_quals && Ln.is_fq) ? 2u * L + 1u : L;
    uint32_t n = Ln.name_len + 1u;
    if (Ln.kind == LF_SL_UNMAPPED) { hoff = n + 16u; return; }
    n += lf_ndig(Ln.flag) + 1;
    n += (D.ctg_name_off[Ln.rname + 1] - D.ctg_name_off[Ln.rname]) + 1;
    n += lf_ndig(Ln.pos1) + 1 + lf_ndig((uint32_t)Ln.mapq) + 1;
    n += (D.tlens[2 * (size_t)Ln.rec] - 1) + 7;
    hoff = n;
}

__global__ void lf_sam_len_kernel(lf_sam_dev D, uint64_t *__restrict__ lens, uint32_t *__restrict__ hole /* 2 per line: offset in the line, length */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.n_lines) return;
    const lf_samline_t Ln = D.lines[i];
    lens[i] = lf_sam_line_len(D, Ln);
    uint32_t ho, hl; lf_sam_hole(D, Ln, ho, hl);
    hole[2 * (size_t)i] = ho; hole[2 * (size_t)i + 1] = hl;
}

__global__ void __launch_bounds__(64)
lf_sam_write_kernel(lf_sam_dev D, const uint64_t *__restrict__ offs, char *__restrict__ out)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= D.n_lines) return;
    const lf_samline_t Ln = D.lines[i];
    char *o = out + offs[i];
    uint64_t w = 0;                                                /* wave-uniform cursor */
    auto copy = [&](const char *src, uint64_t n) { for (uint64_t k = lane; k < n; k += 64) o[w + k] = src[k]; w += n; };
    auto lit = [&](const char *s2, uint32_t n) { if ((uint32_t)lane < n) o[w + lane] = s2[lane]; w += n; };      /* n <= 64 */
    auto put_u = [&](uint32_t v) { const uint32_t nd = lf_ndig(v); if (lane == 0) { uint32_t x = v; for (int k = (int)nd - 1; k >= 0; k--) { o[w + k] = (char)('0' + x % 10u); x /= 10u; } } w += nd; };
    auto put_i = [&](int32_t v) { if (v < 0) { if (lane == 0) o[w] = '-'; w += 1; put_u((uint32_t)(-(int64_t)v)); } else put_u((uint32_t)v); };
    auto put_c = [&](char c) { if (lane == 0) o[w] = c; w += 1; };
    if (Ln.kind == LF_SL_LITERAL) { copy(D.blob + Ln.sa_off, Ln.sa_len); return; }
    const uint64_t ro = D.read_off[Ln.read];
    const uint32_t L = (uint32_t)(D.read_off[Ln.read + 1] - ro);
    const bool rev = Ln.kind == LF_SL_MAPPED && (Ln.flag & 16);
    auto put_seq_qual = [&]() {
        if (D.holes) {                                   /* the host fills SEQ (and QUAL) from the caller's own strings */
            w += L;
            if (D.has_quals && Ln.is_fq) w += 1 + (uint64_t)L; else { put_c('\t'); put_c('*'); }
            return;
        }
        const unsigned char *s = D.reads + ro;
        if (!rev) for (uint32_t k = lane; k < L; k += 64) o[w + k] = (char)s[k];
        else for (uint32_t k = lane; k < L; k += 64) o[w + k] = (char)lf_rc_char(s[L - 1 - k]);     /* reverseComplement, :501 */
        w += L; put_c('\t');
        if (D.quals && Ln.is_fq) {
            const unsigned char *q = D.quals + ro;
            if (!rev) for (uint32_t k = lane; k < L; k += 64) o[w + k] = (char)q[k];
            else for (uint32_t k = lane; k < L; k += 64) o[w + k] = (char)q[L - 1 - k];               /* reverse, :502 */
            w += L;
        } else put_c('*');
    };
    copy(D.names + Ln.name_off, Ln.name_len); put_c('\t');
    if (Ln.kind == LF_SL_UNMAPPED) {
        lit("4\t*\t0\t0\t*\t*\t0\t0\t", 16);
        put_seq_qual();
        copy(D.rg, D.rg_len); put_c('\n');
        return;
    }
    put_u(Ln.flag); put_c('\t');
    copy(D.ctg_names + D.ctg_name_off[Ln.rname], D.ctg_name_off[Ln.rname + 1] - D.ctg_name_off[Ln.rname]); put_c('\t');
    put_u(Ln.pos1); put_c('\t'); put_u((uint32_t)Ln.mapq); put_c('\t');
    copy(D.text + D.toffs[2 * (size_t)Ln.rec], D.tlens[2 * (size_t)Ln.rec] - 1);
    lit("\t*\t0\t0\t", 7);
    put_seq_qual();
    lit("\tAS:i:", 6); put_i(Ln.as);
    lit("\tXS:i:0\tNM:i:", 13); put_u((uint32_t)Ln.nm);
    lit("\tMD:Z:", 6);
    copy(D.text + D.toffs[2 * (size_t)Ln.rec + 1], D.tlens[2 * (size_t)Ln.rec + 1] - 1);
    copy(D.rg, D.rg_len);
    if (Ln.sa_len) { lit("\tSA:Z:", 6); copy(D.blob + Ln.sa_off, Ln.sa_len); }
    put_c('\n');
}

/* HOLES mode egress: line i of the chunk's device buffer -> the same offset of the caller's (pinned host) buffer, without
 * its hole.  A wavefront per line and trip; 16-byte stores aligned on the DESTINATION (the link carries them as full
 * write bursts), unaligned 16-byte loads from HBM, single bytes at the edges of a piece. */
__device__ __forceinline__ void lf_wave_copy(const char *__restrict__ s, char *__restrict__ d, uint64_t n, int lane)
{
    if (n == 0) return;
    uint64_t head = (16u - ((uintptr_t)d & 15u)) & 15u; if (head > n) head = n;
    if ((uint64_t)lane < head) d[lane] = s[lane];
    const uint64_t body = (n - head) >> 4;
    const char *sb = s + head; uint4 *db = reinterpret_cast<uint4 *>(d + head);
    uint64_t k = (uint64_t)lane;
    for (; k + 192 < body; k += 256) {                   /* four loads in flight per lane */
        uint4 a, b, c, e;
        __builtin_memcpy(&a, sb + 16 * k, 16); __builtin_memcpy(&b, sb + 16 * (k + 64), 16); __builtin_memcpy(&c, sb + 16 * (k + 128), 16); __builtin_memcpy(&e, sb + 16 * (k + 192), 16);
        db[k] = a; db[k + 64] = b; db[k + 128] = c; db[k + 192] = e;
    }
    for (; k < body; k += 64) { uint4 a; __builtin_memcpy(&a, sb + 16 * k, 16); db[k] = a; }
    const uint64_t done = head + (body << 4);
    if (done + (uint64_t)lane < n) d[done + lane] = s[done + lane];
}
__global__ void __launch_bounds__(256)
lf_sam_scatter_kernel(const char *__restrict__ src, char *__restrict__ dst, const uint64_t *__restrict__ offs, const uint32_t *__restrict__ hole,
                      int n_lines, uint64_t total)
{
    const int lane = threadIdx.x & 63;
    for (int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); i < n_lines; i += (int)((gridDim.x * blockDim.x) >> 6)) {
        const uint64_t o = offs[i], e = i + 1 < n_lines ? offs[i + 1] : total;
        const uint32_t ho = hole[2 * (size_t)i], hl = hole[2 * (size_t)i + 1];
        if (hl == 0) { lf_wave_copy(src + o, dst + o, e - o, lane); continue; }
        lf_wave_copy(src + o, dst + o, ho, lane);
        lf_wave_copy(src + o + ho + hl, dst + o + ho + hl, e - o - ho - hl, lane);
    }
}

#define SSLOT(T, k, bytes) (T *)lfg_dev_slot(dv, LF_DS_SAM0 + (k), (bytes))

/* lines / names / blob: host memory (pinned slots of the caller).  Returns the chunk's text size; the text itself is written
 * on the lane's stream and fetched with lfg_sam_fetch once the caller knows where it goes.
 * holes != 0: the SEQ (/ QUAL) column of every line is left out of the device text; *h_offs (n_lines line offsets) and *h_hole
 * (2 per line: offset of the hole inside the line, its length) tell the host where to put it.  They stay valid until this
 * lane's next build with the same parity. */
extern "C" int lfg_sam_build(const struct lf_index *ix, const lf_params_t *p, int n_lines, const lf_samline_t *lines,
                             const char *names, uint64_t names_bytes, const char *blob, uint64_t blob_bytes,
                             const char *quals, uint64_t quals_bytes, const void *d_quals_src, int n_batch_reads,
                             const lfg_rtext_t *rt, int parity, int holes, uint64_t *total_out, const uint64_t **h_offs_out, const uint32_t **h_hole_out)
{
    *total_out = 0;
    if (h_offs_out) *h_offs_out = nullptr;
    if (h_hole_out) *h_hole_out = nullptr;
    if (n_lines == 0) return LF_OK;
    const int dv = ix->device;
    HIPCHK(hipSetDevice(dv));
    hipStream_t s = (hipStream_t)lfg_lane_stream(dv, 1);
    if (!s) return LF_ERR_HIP;
    lf_dev_state *st = (lf_dev_state *)ix->dev;
    if (!st) { lf_set_error("index is not on a device"); return LF_ERR_NO_DEVICE; }
    const size_t N = (size_t)n_lines;
    lf_samline_t *d_lines = SSLOT(lf_samline_t, 0, N * sizeof(lf_samline_t));
    char *d_names = SSLOT(char, 1, names_bytes + 64), *d_blob = SSLOT(char, 2, blob_bytes + 64);
    uint64_t *d_lens = SSLOT(uint64_t, 3, (N + 1) * 8);
    /* line offsets + holes of this buffer (parity): the scatter kernel of a chunk that waits for its place in the output reads
     * them after the lane has built its next chunk */
    uint64_t *d_offs = SSLOT(uint64_t, (parity & 1) ? 7 : 4, (N + 1) * 8 + 2 * N * 4);
    uint32_t *d_hole = d_offs ? reinterpret_cast<uint32_t *>(d_offs + (N + 1)) : nullptr;
    /* qualities: host bytes in the resident batch's layout, or (lf_map_batch_dev) a device blob in the caller's layout.  In HOLES
     * mode the host prints them itself; the device only needs to know that they exist (a non-null pointer). */
    const bool dev_quals = !holes && (quals || d_quals_src);
    unsigned char *d_quals = dev_quals ? SSLOT(unsigned char, 5, quals_bytes + 64) : nullptr;      /* HOLES mode: no qualities on the device, only the fact that there are some */
    uint64_t *h = (uint64_t *)lfg_pin_slot(LF_PS_SAM0 + 0, 64);
    uint64_t *h_offs = holes ? (uint64_t *)lfg_pin_slot(LF_PS_SAM0 + 1 + (parity & 1), (N + 1) * 8 + 2 * N * 4) : nullptr;
    if (!d_lines || !d_names || !d_blob || !d_lens || !d_offs || (dev_quals && !d_quals) || !h || (holes && !h_offs)) return LF_ERR_NOMEM;
    /* contig names: a few kB, once per index and device (kept with the device state) */
    static std::mutex ctg_mu;
    std::unique_lock<std::mutex> ctg_lock(ctg_mu);
    if (!st->ctg_names) {
        size_t nb = 0;
        for (int c = 0; c < ix->n_seqs; c++) nb += strlen(ix->contigs[c].name);
        char *hn = (char *)malloc(nb + 16); uint32_t *ho = (uint32_t *)malloc(((size_t)ix->n_seqs + 1) * 4);
        size_t o = 0;
        for (int c = 0; c < ix->n_seqs; c++) { ho[c] = (uint32_t)o; const size_t l = strlen(ix->contigs[c].name); memcpy(hn + o, ix->contigs[c].name, l); o += l; }
        ho[ix->n_seqs] = (uint32_t)o;
        char *dn = nullptr; uint32_t *dof = nullptr;
        hipError_t e1 = hipMalloc((void **)&dn, nb + 16), e2 = hipMalloc((void **)&dof, ((size_t)ix->n_seqs + 1) * 4);
        if (e1 == hipSuccess && e2 == hipSuccess) { e1 = hipMemcpy(dn, hn, nb, hipMemcpyHostToDevice); e2 = hipMemcpy(dof, ho, ((size_t)ix->n_seqs + 1) * 4, hipMemcpyHostToDevice); }
        free(hn); free(ho);
        if (e1 != hipSuccess || e2 != hipSuccess) { lf_set_error("lfg_sam_build: contig table upload failed"); return LF_ERR_HIP; }
        st->ctg_names = dn; st->ctg_name_off = dof;
    }
    ctg_lock.unlock();
    char rg[300]; uint32_t rg_len = 0;
    if (p->read_group_id[0]) rg_len = (uint32_t)snprintf(rg, sizeof rg, "\tRG:Z:%s", p->read_group_id);
    char *d_rg = SSLOT(char, 6, 512);
    if (!d_rg) return LF_ERR_NOMEM;
    hipEvent_t E = (hipEvent_t)lfg_lane_event(dv, 42 + (parity & 1)), F = (hipEvent_t)lfg_lane_event(dv, 44 + (parity & 1));
    hipStream_t s0 = (hipStream_t)lfg_lane_stream(dv, 0);
    if (!E || !F || !s0) return LF_ERR_HIP;
    /* this buffer's offsets / holes are read by the copy of its previous text: wait for that copy before overwriting them */
    HIPCHK(hipStreamWaitEvent(s, F, 0));
    HIPCHK(hipMemcpyAsync(d_lines, lines, N * sizeof(lf_samline_t), hipMemcpyHostToDevice, s));
    if (names_bytes) HIPCHK(hipMemcpyAsync(d_names, names, names_bytes, hipMemcpyHostToDevice, s));
    if (blob_bytes) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_bytes, hipMemcpyHostToDevice, s));
    if (dev_quals && quals) HIPCHK(hipMemcpyAsync(d_quals, quals, quals_bytes, hipMemcpyHostToDevice, s));
    else if (dev_quals && d_quals_src && n_batch_reads > 0) {
        const uint64_t *d_src_off = (const uint64_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 13, 0), *d_off = (const uint64_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 1, 0);
        if (!d_src_off || !d_off) { lf_set_error("lfg_sam_build: no resident read batch"); return LF_ERR_ARG; }
        const int grc = lfg_gather_reads(dv, (void *)s, d_quals_src, d_src_off, d_off, n_batch_reads, d_quals);
        if (grc != LF_OK) return grc;
    }
    if (rg_len) HIPCHK(hipMemcpyAsync(d_rg, rg, rg_len, hipMemcpyHostToDevice, s));      /* pageable source: copied before the call returns */
    lf_sam_dev D;
    D.lines = d_lines; D.n_lines = n_lines;
    D.reads = (const unsigned char *)lfg_dev_slot(dv, LF_DS_SEED0 + 0, 0); D.read_off = (const uint64_t *)lfg_dev_slot(dv, LF_DS_SEED0 + 1, 0);
    D.quals = d_quals; D.has_quals = (quals || d_quals_src) ? 1 : 0; D.names = d_names; D.blob = d_blob;
    D.text = (const char *)rt->d_text; D.toffs = (const uint64_t *)rt->d_offs; D.tlens = (const uint32_t *)rt->d_lens;
    D.ctg_names = (const char *)st->ctg_names; D.ctg_name_off = (const uint32_t *)st->ctg_name_off;
    D.rg = d_rg; D.rg_len = rg_len; D.holes = holes ? 1 : 0;
    hipLaunchKernelGGL(lf_sam_len_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, s, D, d_lens, d_hole);
    { lf_scan_u64 f; f.p = d_lens; const int src = lf_scan_excl(dv, 6, s, f, d_offs, (size_t)n_lines); if (src != LF_OK) return src; }
    HIPCHK(hipMemcpyAsync(h, d_offs + (N - 1), 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h + 1, d_lens + (N - 1), 8, hipMemcpyDeviceToHost, s));
    if (holes) {
        HIPCHK(hipMemcpyAsync(h_offs, d_offs, N * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h_offs + (N + 1), d_hole, 2 * N * 4, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    const uint64_t total = h[0] + h[1];
    char *d_out = SSLOT(char, 8 + (parity & 1), total + 64);      /* two text buffers: the previous chunk's may still wait for its place in the output */
    if (!d_out) return LF_ERR_NOMEM;
    /* events of this buffer: E = text written, F = text copied out (lfg_sam_fetch_async).  The writer waits for the copy of the
     * buffer's previous text (above); the seed stream -- the first one the NEXT chunk touches, with the upload of its reads --
     * waits for the writer, which still reads this chunk's batch.  No host wait anywhere. */
    hipLaunchKernelGGL(lf_sam_write_kernel, dim3((unsigned)n_lines), dim3(64), 0, s, D, (const uint64_t *)d_offs, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(E, s));
    HIPCHK(hipStreamWaitEvent(s0, E, 0));
    *total_out = total;
    if (holes) { if (h_offs_out) *h_offs_out = h_offs; if (h_hole_out) *h_hole_out = reinterpret_cast<const uint32_t *>(h_offs + (N + 1)); }
    /* what lfg_sam_fetch_async needs to know about this buffer */
    lfg_lane_set_value(dv, 2 + (parity & 1), holes ? (uint64_t)n_lines : 0ull);
    return LF_OK;
}

/* waits until the text of the lane's last lfg_sam_build is complete in its buffer (the kernels read the chunk's reads,
 * names and CIGAR / MD text: the next chunk may only overwrite those afterwards) */
extern "C" int lfg_sam_wait(const struct lf_index *ix)
{
    const int dv = ix->device;
    HIPCHK(hipSetDevice(dv));
    hipStream_t s = (hipStream_t)lfg_lane_stream(dv, 1);
    if (!s) return LF_ERR_HIP;
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return LF_OK;
}

/* asynchronous form of lfg_sam_fetch: the copy runs on the lane's copy stream behind the writer kernel; the lane maps its
 * next chunk meanwhile.  dst must stay valid (and should be pinned) until lfg_sam_fetch_wait. */
extern "C" int lfg_sam_fetch_async(const struct lf_index *ix, char *dst, uint64_t total, int parity)
{
    if (!total) return LF_OK;
    const int dv = ix->device;
    HIPCHK(hipSetDevice(dv));
    hipStream_t cs = (hipStream_t)lfg_lane_stream(dv, 14);
    hipEvent_t E = (hipEvent_t)lfg_lane_event(dv, 42 + (parity & 1)), F = (hipEvent_t)lfg_lane_event(dv, 44 + (parity & 1));
    if (!cs || !E || !F) return LF_ERR_HIP;
    const char *d_out = (const char *)lfg_dev_slot(dv, LF_DS_SAM0 + 8 + (parity & 1), 0);
    if (!d_out) { lf_set_error("lfg_sam_fetch_async: nothing was built"); return LF_ERR_ARG; }
    HIPCHK(hipStreamWaitEvent(cs, E, 0));
    const int n_lines = (int)lfg_lane_value(dv, 2 + (parity & 1));          /* > 0: the buffer was built in HOLES mode */
    if (n_lines > 0) {
        char *d_dst = nullptr;
        if (hipHostGetDevicePointer((void **)&d_dst, dst, 0) != hipSuccess || !d_dst) { (void)hipGetLastError(); lf_set_error("lfg_sam_fetch_async: the output buffer is not mapped into the device"); return LF_ERR_ARG; }
        const uint64_t *d_offs = (const uint64_t *)lfg_dev_slot(dv, LF_DS_SAM0 + ((parity & 1) ? 7 : 4), 0);
        if (!d_offs) { lf_set_error("lfg_sam_fetch_async: nothing was built"); return LF_ERR_ARG; }
        const uint32_t *d_hole = reinterpret_cast<const uint32_t *>(d_offs + ((size_t)n_lines + 1));
        /* a few dozen workgroups saturate the link (profiles/tools/ubench/pcie_shader.hip); more would only take CUs from the mapping kernels */
        const unsigned wg = (unsigned)std::min(64, (n_lines + 3) / 4);
        hipLaunchKernelGGL(lf_sam_scatter_kernel, dim3(wg), dim3(256), 0, cs, d_out, d_dst, d_offs, d_hole, n_lines, total);
        HIPCHK(hipGetLastError());
    } else
    HIPCHK(hipMemcpyAsync(dst, d_out, total, hipMemcpyDefault, cs));      /* dst: host memory, or HBM (lf_map_batch_dev) */
    HIPCHK(hipEventRecord(F, cs));
    return LF_OK;
}
/* can kernels of `device` store into [p, p + bytes)?  (pinned / registered host memory: HOLES mode's destination) */
extern "C" int lfg_host_mapped(int device, const void *p, size_t bytes)
{
    if (!p || hipSetDevice(device) != hipSuccess) return 0;
    hipPointerAttribute_t at; memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (at.type != hipMemoryTypeHost) return 0;
    hipPointerAttribute_t at2; memset(&at2, 0, sizeof at2);
    if (bytes > 1 && (hipPointerGetAttributes(&at2, (const char *)p + bytes - 1) != hipSuccess || at2.type != hipMemoryTypeHost)) { (void)hipGetLastError(); return 0; }
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void *>(p), 0) != hipSuccess || !d) { (void)hipGetLastError(); return 0; }
    return 1;
}
extern "C" int lfg_sam_fetch_wait(const struct lf_index *ix)
{
    const int dv = ix->device;
    HIPCHK(hipSetDevice(dv));
    hipStream_t cs = (hipStream_t)lfg_lane_stream(dv, 14);
    if (!cs) return LF_ERR_HIP;
    HIPCHK(hipStreamSynchronize(cs));
    HIPCHK(hipGetLastError());
    return LF_OK;
}

/* the text of this lane's last lfg_sam_build with the same parity -> dst (host; pinned memory copies at link speed) */
extern "C" int lfg_sam_fetch(const struct lf_index *ix, char *dst, uint64_t total, int parity)
{
    if (!total) return LF_OK;
    const int dv = ix->device;
    HIPCHK(hipSetDevice(dv));
    hipStream_t s = (hipStream_t)lfg_lane_stream(dv, 1);
    if (!s) return LF_ERR_HIP;
    const char *d_out = (const char *)lfg_dev_slot(dv, LF_DS_SAM0 + 8 + (parity & 1), 0);
    if (!d_out) { lf_set_error("lfg_sam_fetch: nothing was built"); return LF_ERR_ARG; }
    HIPCHK(hipMemcpyAsync(dst, d_out, total, hipMemcpyDefault, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return LF_OK;
}
