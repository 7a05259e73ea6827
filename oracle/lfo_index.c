/*
 * lfo_index.c -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * Index loader for the reference's on-disk formats and the FM-index primitives.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lf_oracle.h"

static void *slurp(const char *path, size_t skip, size_t *n_out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) { fprintf(stderr, "[lfo] cannot open %s\n", path); return NULL; }
    fseek(fp, 0, SEEK_END);
    size_t sz = (size_t)ftell(fp);
    if (sz < skip) { fclose(fp); return NULL; }
    fseek(fp, (long)skip, SEEK_SET);
    size_t n = sz - skip;
    void *buf = malloc(n ? n : 1);
    if (fread(buf, 1, n, fp) != n) { free(buf); fclose(fp); return NULL; }
    fclose(fp);
    *n_out = n;
    return buf;
}

void lfo_params_default(lfo_params_t *p)
{
    memset(p, 0, sizeof(*p));
    p->min_anchor_len = 14; p->sampling_count = 1000; p->max_map = 10;
    p->min_read_len = 1000; p->max_ref_hits = 1000; p->chain_alg = 0;
    p->chain_reward = 9.3; p->chain_penalty = 11.4; p->gap_penalty = 0.15; p->threads = 1;
}

static void cache_gen(lfo_index_t *ix);

/* Follows bwt_restore_bwt / bwt_restore_sa (lib/bwa/bwt.c:421-462), bns_restore_core
 * (lib/bwa/bntseq.c:100-160), bwa_idx_load_from_disk (lib/bwa/bwa.c:252-284) and bwt_cache_load
 * (src/BWT.cpp:159-187). */
lfo_index_t *lfo_index_load(const char *prefix)
{
    char path[4096];
    size_t n;
    lfo_index_t *ix = (lfo_index_t *)calloc(1, sizeof(*ix));

    snprintf(path, sizeof path, "%s.bwt", prefix);
    uint64_t *raw = (uint64_t *)slurp(path, 0, &n);
    if (!raw || n < 40) goto fail;
    ix->primary = raw[0];
    ix->L2[0] = 0;
    for (int i = 1; i <= 4; i++) ix->L2[i] = raw[i];
    ix->seq_len = ix->L2[4];
    ix->bwt_size = (n - 40) >> 2;
    ix->bwt = (uint32_t *)malloc(ix->bwt_size * 4 + 64);
    memcpy(ix->bwt, raw + 5, ix->bwt_size * 4);
    free(raw);

    snprintf(path, sizeof path, "%s.sa", prefix);
    raw = (uint64_t *)slurp(path, 0, &n);
    if (!raw || n < 56) goto fail;
    if (raw[0] != ix->primary || raw[6] != ix->seq_len) { fprintf(stderr, "[lfo] SA-BWT inconsistency\n"); goto fail; }
    ix->sa_intv = raw[5];
    ix->n_sa = (ix->seq_len + ix->sa_intv) / ix->sa_intv;
    ix->sa = (uint64_t *)malloc(ix->n_sa * 8);
    ix->sa[0] = (uint64_t)-1;
    memcpy(ix->sa + 1, raw + 7, (ix->n_sa - 1) * 8);
    free(raw);

    snprintf(path, sizeof path, "%s.ann", prefix);
    {
        FILE *fp = fopen(path, "r");
        if (!fp) goto fail;
        long long xx; unsigned seed; int nseq;
        if (fscanf(fp, "%lld%d%u", &xx, &nseq, &seed) != 3) { fclose(fp); goto fail; }
        ix->l_pac = xx; ix->n_seqs = nseq;
        ix->contigs = (lfo_contig_t *)calloc((size_t)nseq, sizeof(lfo_contig_t));
        for (int i = 0; i < nseq; i++) {
            unsigned gi; char name[8192]; int c, len, nambs;
            if (fscanf(fp, "%u%8191s", &gi, name) != 2) { fclose(fp); goto fail; }
            while ((c = fgetc(fp)) != '\n' && c != EOF) {}
            if (fscanf(fp, "%lld%d%d", &xx, &len, &nambs) != 3) { fclose(fp); goto fail; }
            ix->contigs[i].offset = xx; ix->contigs[i].len = len; ix->contigs[i].name = strdup(name);
        }
        fclose(fp);
    }

    snprintf(path, sizeof path, "%s.pac", prefix);
    ix->pac = (uint8_t *)slurp(path, 0, &n);
    if (!ix->pac || n < (size_t)(ix->l_pac / 4 + 1)) goto fail;

    snprintf(path, sizeof path, "%s.cache", prefix);
    {
        FILE *fp = fopen(path, "rb");
        if (!fp) { cache_gen(ix); return ix; }     /* 256 MiB file is optional: rebuild it in memory */
        int32_t hdr[2];
        if (fread(hdr, 4, 2, fp) != 2) { fclose(fp); goto fail; }
        ix->kcache = hdr[0];
        size_t cs = (size_t)hdr[1];
        ix->cache = (uint64_t *)malloc(cs * 16);
        if (fread(ix->cache, 16, cs, fp) != cs) { fclose(fp); goto fail; }
        fclose(fp);
    }
    return ix;
fail:
    fprintf(stderr, "[lfo] failed to load index %s\n", prefix);
    lfo_index_free(ix);
    return NULL;
}

void lfo_index_free(lfo_index_t *ix)
{
    if (!ix) return;
    free(ix->bwt); free(ix->sa); free(ix->pac); free(ix->cache);
    if (ix->contigs) { for (int i = 0; i < ix->n_seqs; i++) free(ix->contigs[i].name); free(ix->contigs); }
    free(ix);
}

void lfo_free(void *p) { free(p); }

/* ---------------- FM-index primitives ---------------- */

/* number of 2-bit symbols equal to c among the first r (1..32) symbols of a 64-bit chunk, symbols
 * stored most-significant first.  Same value as __occ_aux on the masked word (lib/bwa/bwt.c:98-105,
 * 121-123) including its "c==0" correction, written as a direct count. */
static inline int count_sym(uint64_t y, int c, int r)
{
    static const uint64_t rep[4] = { 0x0ull, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull, 0xFFFFFFFFFFFFFFFFull };
    uint64_t eq = ~(y ^ rep[c]);
    uint64_t m = eq & (eq >> 1) & 0x5555555555555555ull;
    if (r < 32) m &= ~0ull << (64 - 2 * r);
    return __builtin_popcountll(m);
}

/* Occ(k,c): occurrences of c in B[0..k] (lib/bwa/bwt.c:107-127). Block = 4 x u64 counts + 128 symbols
 * (lib/bwa/bwt.h:72-73). */
uint64_t lfo_occ(const lfo_index_t *ix, uint64_t k, int c)
{
    if (k == ix->seq_len) return ix->L2[c + 1] - ix->L2[c];
    if (k == (uint64_t)-1) return 0;
    if (k >= ix->primary) k--;                       /* $ is not stored */
    const uint32_t *blk = ix->bwt + ((k >> 7) << 4);
    uint64_t n;
    memcpy(&n, (const char *)blk + 8 * c, 8);
    const uint32_t *w = blk + 8;
    int rem = (int)(k & 127) + 1;                    /* symbols of this block to count */
    for (int i = 0; rem > 0; i += 2, rem -= 32) {
        uint64_t y = ((uint64_t)w[i] << 32) | w[i + 1];
        n += (uint64_t)count_sym(y, c, rem >= 32 ? 32 : rem);
    }
    return n;
}

/* bwt_B0 (lib/bwa/bwt.h:78) */
static inline int bwt_char(const lfo_index_t *ix, uint64_t x)
{
    uint32_t w = ix->bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)];
    return (int)((w >> ((~x & 15) << 1)) & 3);
}

/* bwt_invPsi + bwt_sa (lib/bwa/bwt.c:53-59,86-96) */
uint64_t lfo_sa(const lfo_index_t *ix, uint64_t k, uint32_t *steps)
{
    uint64_t off = 0, mask = ix->sa_intv - 1;
    while (k & mask) {
        off++;
        if (k == ix->primary) { k = 0; continue; }
        uint64_t x = k - (k > ix->primary);
        int c = bwt_char(ix, x);
        k = ix->L2[c] + lfo_occ(ix, k, c);
    }
    if (steps) *steps = (uint32_t)off;
    return off + ix->sa[k / ix->sa_intv];
}

static inline int nt4(char ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

/* bwt_count_exact_cached (src/BWT.cpp:265-298).  `avail` = characters of the read available at str;
 * the reference instead runs into the read's terminating NUL (code 4 -> "no match"), which is the
 * same result as refusing len > avail.  stats: [0] table lookups, [1] occ blocks. */
static uint64_t *g_stats_dummy;
static int64_t count_cached(const lfo_index_t *ix, const char *str, int len, int avail,
                            uint64_t *sp, uint64_t *ep, uint64_t *stats)
{
    if (len > avail) return 0;
    int kc = ix->kcache;
    int32_t idx = 0;
    for (int i = len - 1; i >= len - kc; --i) {
        int c = nt4(str[i]);
        if (c > 3) return 0;
        idx = idx * 4 + c;
    }
    if (stats) stats[0]++;
    uint64_t k = ix->cache[2 * (size_t)idx], l = ix->cache[2 * (size_t)idx + 1];
    if (k > l) return 0;
    for (int i = len - kc - 1; i >= 0; --i) {
        int c = nt4(str[i]);
        if (c > 3) return 0;
        if (stats) {
            /* touches counted like bwt_2occ: one block when k-1 and l share a block, else two
             * (lib/bwa/bwt.c:132-139) */
            uint64_t km = k - 1, _k = (km >= ix->primary) ? km - 1 : km, _l = (l >= ix->primary) ? l - 1 : l;
            if (km == (uint64_t)-1 || l == (uint64_t)-1 || (_k >> 7) != (_l >> 7)) stats[1] += 2; else stats[1] += 1;
        }
        uint64_t ok = lfo_occ(ix, k - 1, c), ol = lfo_occ(ix, l, c);
        k = ix->L2[c] + ok + 1;
        l = ix->L2[c] + ol;
        if (k > l) return 0;
    }
    *sp = k; *ep = l;
    return (int64_t)(l - k + 1);
}

/* bwt_cache_gen (src/BWT.cpp:60-115): SA interval of every 12-mer; entry index = base-4 number with the
 * LAST character most significant; children are expanded in place, high index first. An empty parent
 * (beg > end) hands its (beg,end) down unchanged. */
static void cache_gen(lfo_index_t *ix)
{
    const int K = 12;
    size_t cs = (size_t)1 << (2 * K);
    uint64_t *T = (uint64_t *)malloc(cs * 16);
    T[0] = 0; T[1] = ix->seq_len;
    for (int k = 0; k < K; k++) {
        long os = 1L << (2 * k);
        for (long i = os - 1; i >= 0; i--) {
            uint64_t bk = T[2 * i], bl = T[2 * i + 1];
            for (int j = 3; j >= 0; j--) {
                size_t ni = (size_t)i * 4 + (size_t)j;
                if (bk > bl) { T[2 * ni] = bk; T[2 * ni + 1] = bl; }
                else {
                    T[2 * ni] = ix->L2[j] + lfo_occ(ix, bk - 1, j) + 1;
                    T[2 * ni + 1] = ix->L2[j] + lfo_occ(ix, bl, j);
                }
            }
        }
    }
    ix->kcache = K;
    ix->cache = T;
}

int64_t lfo_count_exact_cached(const lfo_index_t *ix, const char *str, int len, int avail,
                               uint64_t *sp, uint64_t *ep)
{
    (void)g_stats_dummy;
    return count_cached(ix, str, len, avail, sp, ep, NULL);
}

/* getLocs_extend_whole_step (src/BWT.cpp:312-394) */
void lfo_seed(const lfo_index_t *ix, const lfo_params_t *p, const char *seq, uint32_t qLen,
              lfo_seed_t *F, uint32_t *nF, lfo_seed_t *R, uint32_t *nR, uint64_t *stats)
{
    uint32_t hash_count = (uint32_t)p->sampling_count;
    double step = (double)qLen / hash_count;
    double seed_pos = 0;
    uint32_t pos = 0, last_pos = 0, numF = 0, numR = 0;
    uint64_t sp = 0, ep = 0, sp2 = 0, ep2 = 0;

    for (uint32_t i = 0; i < hash_count; i++) {
        int m = p->min_anchor_len;
        int avail = (pos <= qLen) ? (int)(qLen - pos) : 0;
        uint64_t occ = (uint64_t)count_cached(ix, seq + pos, m, avail, &sp, &ep, stats);
        int64_t occ2;
        while ((occ2 = count_cached(ix, seq + pos, m + 1, avail, &sp2, &ep2, stats)) > 0) {
            occ = (uint64_t)occ2; sp = sp2; ep = ep2; m++;
        }
        if (occ > 0 && occ < (uint64_t)p->max_ref_hits && (pos + (uint32_t)m) > last_pos) {
            for (uint64_t j = sp; j <= ep; j++) {
                uint32_t steps;
                uint64_t sapos = lfo_sa(ix, j, &steps);
                if (stats) { stats[1] += steps; stats[2] += 1; }
                if (sapos >= (uint64_t)ix->l_pac) {        /* reverse strand (src/BWT.cpp:351-358) */
                    sapos = ((uint64_t)ix->l_pac << 1) - sapos - (uint64_t)m;
                    R[numR].tPos = (uint32_t)sapos;
                    R[numR].qPos = (qLen - pos - (uint32_t)m) & 0xFFFFF;
                    R[numR].len = (uint32_t)m & 0xFFF;
                    numR++;
                } else {
                    F[numF].tPos = (uint32_t)sapos;
                    F[numF].qPos = pos & 0xFFFFF;
                    F[numF].len = (uint32_t)m & 0xFFF;
                    numF++;
                }
            }
            last_pos = pos + (uint32_t)m;
        }
        seed_pos += step;
        pos = (uint32_t)seed_pos;
    }
    if (stats) stats[3] += qLen;
    *nF = numF; *nR = numR;
}

/* ---------------- reference fetch ---------------- */

/* _get_pac (src/BWT.cpp:310), bwt_str_pac2char (src/BWT.cpp:601-607) */
void lfo_pac2char(const lfo_index_t *ix, uint32_t beg, uint32_t len, char *out)
{
    for (uint32_t i = 0; i < len; i++) {
        uint32_t l = beg + i;
        out[i] = "ACGT"[(ix->pac[l >> 2] >> ((~l & 3) << 1)) & 3];
    }
}

/* bns_pos2rid (lib/bwa/bntseq.c:349-363): contig containing pos, by binary search over offsets.
 * For pos >= l_pac the reference returns -1 and then indexes anns[-1] (undefined
 * behaviour, unreachable for reads lying inside the reference); we clamp to the last contig. */
static int pos2rid(const lfo_index_t *ix, int64_t pos)
{
    if (pos >= ix->l_pac) return ix->n_seqs - 1;
    int lo = 0, hi = ix->n_seqs - 1;
    while (lo < hi) {                      /* last contig whose offset <= pos */
        int mid = (lo + hi + 1) >> 1;
        if (ix->contigs[mid].offset <= pos) lo = mid; else hi = mid - 1;
    }
    return lo;
}

/* bwt_get_chr_boundaries (src/BWT.cpp:653-666): boundaries of the contig holding the MIDPOINT */
void lfo_chr_boundaries(const lfo_index_t *ix, uint64_t beg, uint64_t end, uint32_t *cb, uint32_t *ce)
{
    int rid = pos2rid(ix, (int64_t)((beg + end) >> 1));
    *cb = (uint32_t)ix->contigs[rid].offset;
    *ce = (uint32_t)(ix->contigs[rid].offset + ix->contigs[rid].len - 1);
}

int lfo_pos2rid(const lfo_index_t *ix, int64_t pos) { return pos2rid(ix, pos); }
