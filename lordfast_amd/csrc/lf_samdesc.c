/* lf_samdesc.c -- printSamEntry (src/LordFAST.cpp:318-459) on the host side of the device path: MAPQ (the reference's double
 * arithmetic, :325-356), flags, record order, SA:Z strings, and ONE 48-byte descriptor per output line for lf_sam.hip, which
 * writes the text; the reverse complement of src/Common.cpp:31-40; the host fill of the SEQ / QUAL columns (HOLES mode); the
 * SAM header (src/BWT.cpp:668-681).  print_sam_entry is the reference's line assembly itself, used for the rare lines the
 * device cannot print (reads shorter than -l, names over 64 KiB, per-base fallback records) and by the host-SAM cross-check. */
#include "lf_pipe.h"

/* tableRev, src/Common.cpp:31-40: case kept, anything else 'N' */
char g_rc_tab[256];
void rc_tab_init(void)
{
    memset(g_rc_tab, 'N', sizeof g_rc_tab);
    g_rc_tab['A'] = 'T'; g_rc_tab['C'] = 'G'; g_rc_tab['G'] = 'C'; g_rc_tab['T'] = 'A';
    g_rc_tab['a'] = 't'; g_rc_tab['c'] = 'g'; g_rc_tab['g'] = 'c'; g_rc_tab['t'] = 'a';
}
void revcomp_into(const char *s, char *out, uint32_t len) { rc_copy(out, s, len); out[len] = 0; }
/* reverse complement / reversed copy written straight into the SAM text (src/LordFAST.cpp:501-502 build both strings
 * for every read; only records on the reverse strand ever print them) */
/* reverse complement of l bytes: 16 at a time with two nibble-indexed byte shuffles where the CPU has SSSE3 (every x86-64
 * server of the last 15 years).  A, C, G, T and their lower-case forms differ from every other letter in (low nibble, bit 6,
 * bit 5): the complement comes out of one table indexed by the low nibble, 'N' everywhere else. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("ssse3"))) static void rc_copy_ssse3(char *d, const char *s, size_t l)
{
    /* low nibble -> complement (upper case) for A=0x41 C=0x43 G=0x47 T=0x54: nibbles 1, 3, 7, 4 */
    const __m128i tab = _mm_setr_epi8('N', 'T', 'N', 'G', 'A', 'N', 'N', 'C', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N');
    /* the byte a nibble must come from to be a base: 1 -> 'A', 3 -> 'C', 7 -> 'G', 4 -> 'T' (upper case) */
    const __m128i src = _mm_setr_epi8(0x20, 'A', 0x20, 'C', 'T', 0x20, 0x20, 'G', 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20);   /* 0x20: no upper-cased byte equals it */
    const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m128i lo4 = _mm_set1_epi8(0x0f), caseb = _mm_set1_epi8(0x20), up = _mm_set1_epi8((char)0xDF), enn = _mm_set1_epi8('N');
    size_t i = 0;
    for (; i + 16 <= l; i += 16) {
        __m128i x = _mm_loadu_si128((const __m128i *)(s + l - 16 - i));
        x = _mm_shuffle_epi8(x, rev);
        const __m128i cs = _mm_and_si128(x, caseb), xu = _mm_and_si128(x, up), nib = _mm_and_si128(x, lo4);
        const __m128i ok = _mm_cmpeq_epi8(_mm_shuffle_epi8(src, nib), xu);           /* really one of ACGT / acgt */
        __m128i c = _mm_or_si128(_mm_shuffle_epi8(tab, nib), cs);                       /* complement, case kept */
        c = _mm_or_si128(_mm_and_si128(ok, c), _mm_andnot_si128(ok, enn));
        _mm_storeu_si128((__m128i *)(d + i), c);
    }
    for (; i < l; i++) d[i] = rc_char(s[l - 1 - i]);
}
#endif
/* Copies whose destination is not read again by this CPU -- the pinned staging buffer the bases are uploaded from, the SEQ /
 * QUAL holes of the caller's SAM buffer: non-temporal stores write them straight to memory, without first reading the
 * destination lines into the cache (a third of the memory traffic of a plain copy of this size) and without evicting what
 * the worker threads do need.  The destination is brought to a 16-byte boundary first. */
void lf_copy_stream(char *d, const char *s, size_t n)
{
#if defined(__x86_64__)
    if (n >= 256) {
        size_t head = (16 - ((uintptr_t)d & 15)) & 15;
        memcpy(d, s, head); d += head; s += head; n -= head;
        size_t i = 0;
        for (; i + 64 <= n; i += 64) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 16));
            const __m128i c = _mm_loadu_si128((const __m128i *)(s + i + 32)), e = _mm_loadu_si128((const __m128i *)(s + i + 48));
            _mm_stream_si128((__m128i *)(d + i), a); _mm_stream_si128((__m128i *)(d + i + 16), b);
            _mm_stream_si128((__m128i *)(d + i + 32), c); _mm_stream_si128((__m128i *)(d + i + 48), e);
        }
        memcpy(d + i, s + i, n - i);
        _mm_sfence();
        return;
    }
#endif
    memcpy(d, s, n);
}
/* ---- packed upload: a read's bases as bit planes, written at bit offset `a` of the chunk's planes (lo / hi / valid, `qw` words
 * each; bit i of word i / 64 describes base i of the concatenated chunk).  valid = the byte is one of "ACGT" (upper case: what
 * edlib's raw byte compare can match); code = A0 C1 G2 T3, zero where not valid -- exactly what lf_pack_planes_kernel makes of
 * the bytes on the device.  Whole words are plain stores; the (up to two) words a read shares with its neighbours are OR-ed in
 * atomically (the caller zeroes every word that holds a read boundary).  Bytes that are not valid go to the exception list
 * (position in the chunk, byte); returns 0 when the list is full (the caller uploads the bytes instead). ---- */
static inline void pack_bits_scalar(const unsigned char *s, size_t n, uint64_t *lo, uint64_t *hi, uint64_t *va)
{
    uint64_t l = 0, h = 0, v = 0;
    for (size_t i = 0; i < n; i++) {
        const unsigned c = s[i];
        const uint64_t ok = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
        const unsigned x = (c >> 1) & 3u, code = x ^ (x >> 1);
        l |= (ok & (code & 1u)) << i; h |= (ok & (code >> 1)) << i; v |= ok << i;
    }
    *lo = l; *hi = h; *va = v;
}
#if defined(__x86_64__)
/* 64 bases -> (low bit, high bit, valid) words, 32 bytes per step where the host has AVX2 (checked once): half the instructions of the
 * SSE2 form -- the pack is the first thing every lane of a host batch waits for, one lane after the other */
__attribute__((target("avx2"))) static inline void pack64_avx2(const unsigned char *p, uint64_t *lo, uint64_t *hi, uint64_t *va)
{
    uint64_t l = 0, h = 0, v = 0;
    for (int q = 0; q < 2; q++) {
        const __m256i x = _mm256_loadu_si256((const __m256i *)(p + 32 * q));
        const __m256i b2 = _mm256_slli_epi16(x, 5), b1 = _mm256_slli_epi16(x, 6);
        const __m256i okv = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(x, _mm256_set1_epi8('C'))),
                                            _mm256_or_si256(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(x, _mm256_set1_epi8('T'))));
        const uint64_t vm = (uint64_t)(uint32_t)_mm256_movemask_epi8(okv);
        const uint64_t hm = (uint64_t)(uint32_t)_mm256_movemask_epi8(b2) & vm;
        const uint64_t lm = (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_xor_si256(b1, b2)) & vm;
        l |= lm << (32 * q); h |= hm << (32 * q); v |= vm << (32 * q);
    }
    *lo = l; *hi = h; *va = v;
}
static int have_avx2(void) { static int x = -1; if (x < 0) x = __builtin_cpu_supports("avx2") ? 1 : 0; return x; }
#endif

int lf_pack_read(uint64_t *planes, uint64_t qw, uint64_t a, const char *seq, uint32_t len,
                 uint64_t *exc_pos, uint8_t *exc_byte, uint64_t exc_cap, uint64_t *exc_n)
{
    uint64_t *LO = planes, *HI = planes + qw, *VA = planes + 2 * qw;
    const unsigned char *s = (const unsigned char *)seq;
    size_t i = 0; int ok_all = 1;
    uint64_t pos = a;
#if defined(__x86_64__)
    const int avx2 = have_avx2();
#endif
#define LF_PACK_EXC(word_valid, count, base_i) do { \
        uint64_t bad_ = ~(word_valid) & ((count) >= 64 ? ~0ull : ((1ull << (count)) - 1)); \
        while (bad_) { const int b_ = __builtin_ctzll(bad_); bad_ &= bad_ - 1; \
            const uint64_t k_ = __atomic_fetch_add(exc_n, 1, __ATOMIC_RELAXED); \
            if (k_ < exc_cap) { exc_pos[k_] = a + (base_i) + (uint64_t)b_; exc_byte[k_] = s[(base_i) + (size_t)b_]; } else ok_all = 0; } } while (0)
    /* head: up to the next word boundary */
    if (pos & 63) {
        size_t n = 64 - (size_t)(pos & 63); if (n > len) n = len;
        uint64_t l, h, v; pack_bits_scalar(s, n, &l, &h, &v);
        const unsigned sh = (unsigned)(pos & 63);
        __atomic_fetch_or(&LO[pos >> 6], l << sh, __ATOMIC_RELAXED); __atomic_fetch_or(&HI[pos >> 6], h << sh, __ATOMIC_RELAXED); __atomic_fetch_or(&VA[pos >> 6], v << sh, __ATOMIC_RELAXED);
        LF_PACK_EXC(v, n, (size_t)0);
        i = n; pos += n;
    }
    /* whole words: 64 bases each */
    for (; i + 64 <= len; i += 64, pos += 64) {
        uint64_t l, h, v;
#if defined(__x86_64__)
        l = h = v = 0;
        if (avx2) pack64_avx2(s + i, &l, &h, &v);
        else for (int q = 0; q < 4; q++) {
            const __m128i x = _mm_loadu_si128((const __m128i *)(s + i + 16 * q));
            const __m128i b2 = _mm_slli_epi16(x, 5), b1 = _mm_slli_epi16(x, 6);           /* bit 2 / bit 1 of every byte in its top bit */
            const __m128i okv = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(x, _mm_set1_epi8('A')), _mm_cmpeq_epi8(x, _mm_set1_epi8('C'))),
                                             _mm_or_si128(_mm_cmpeq_epi8(x, _mm_set1_epi8('G')), _mm_cmpeq_epi8(x, _mm_set1_epi8('T'))));
            const uint64_t vm = (uint64_t)(uint32_t)_mm_movemask_epi8(okv);
            const uint64_t hm = (uint64_t)(uint32_t)_mm_movemask_epi8(b2) & vm;            /* code high bit = bit 2 (A C: 0, G T: 1) */
            const uint64_t lm = (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_xor_si128(b1, b2)) & vm;   /* low bit = bit 1 ^ bit 2 */
            l |= lm << (16 * q); h |= hm << (16 * q); v |= vm << (16 * q);
        }
#else
        pack_bits_scalar(s + i, 64, &l, &h, &v);
#endif
        LO[pos >> 6] = l; HI[pos >> 6] = h; VA[pos >> 6] = v;
        if (v != ~0ull) LF_PACK_EXC(v, 64, i);
    }
    /* tail */
    if (i < len) {
        const size_t n = len - i;
        uint64_t l, h, v; pack_bits_scalar(s + i, n, &l, &h, &v);
        __atomic_fetch_or(&LO[pos >> 6], l, __ATOMIC_RELAXED); __atomic_fetch_or(&HI[pos >> 6], h, __ATOMIC_RELAXED); __atomic_fetch_or(&VA[pos >> 6], v, __ATOMIC_RELAXED);
        LF_PACK_EXC(v, n, i);
    }
#undef LF_PACK_EXC
    return ok_all;
}

#if defined(__x86_64__)
__attribute__((target("ssse3"))) static void rc_copy_stream_ssse3(char *d, const char *s, size_t l)
{
    const __m128i tab = _mm_setr_epi8('N', 'T', 'N', 'G', 'A', 'N', 'N', 'C', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N');
    const __m128i src = _mm_setr_epi8(0x20, 'A', 0x20, 'C', 'T', 0x20, 0x20, 'G', 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20);
    const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m128i lo4 = _mm_set1_epi8(0x0f), caseb = _mm_set1_epi8(0x20), up = _mm_set1_epi8((char)0xDF), enn = _mm_set1_epi8('N');
    size_t i = 0;
    const size_t head = (16 - ((uintptr_t)d & 15)) & 15;
    for (; i < head && i < l; i++) d[i] = rc_char(s[l - 1 - i]);
    for (; i + 16 <= l; i += 16) {
        __m128i x = _mm_loadu_si128((const __m128i *)(s + l - 16 - i));
        x = _mm_shuffle_epi8(x, rev);
        const __m128i cs = _mm_and_si128(x, caseb), xu = _mm_and_si128(x, up), nib = _mm_and_si128(x, lo4);
        const __m128i ok = _mm_cmpeq_epi8(_mm_shuffle_epi8(src, nib), xu);
        __m128i c = _mm_or_si128(_mm_shuffle_epi8(tab, nib), cs);
        c = _mm_or_si128(_mm_and_si128(ok, c), _mm_andnot_si128(ok, enn));
        _mm_stream_si128((__m128i *)(d + i), c);
    }
    for (; i < l; i++) d[i] = rc_char(s[l - 1 - i]);
    _mm_sfence();
}
#endif
/* rc_copy with non-temporal stores (same bytes) */
void rc_copy_stream(char *d, const char *s, size_t l)
{
#if defined(__x86_64__)
    if (l >= 256 && __builtin_cpu_supports("ssse3")) { rc_copy_stream_ssse3(d, s, l); return; }
#endif
    rc_copy(d, s, l);
}
void rc_copy(char *d, const char *s, size_t l)
{
#if defined(__x86_64__)
    static int have = -1;
    if (have < 0) have = __builtin_cpu_supports("ssse3") ? 1 : 0;
    if (have) { rc_copy_ssse3(d, s, l); return; }
#endif
    for (size_t i = 0; i < l; i++) d[i] = rc_char(s[l - 1 - i]);
}
void str_put_rc(str_t *b, const char *s, size_t l)
{
    str_room(b, l);
    if (b->mode != 1) rc_copy(b->s + b->n, s, l);
    b->n += l;
    if (b->mode == 0 || b->mode == 3) b->s[b->n] = 0;
}
void str_put_rev(str_t *b, const char *s, size_t l)
{
    str_room(b, l);
    if (b->mode != 1) { char *d = b->s + b->n; for (size_t i = 0; i < l; i++) d[i] = s[l - 1 - i]; }
    b->n += l;
    if (b->mode == 0 || b->mode == 3) b->s[b->n] = 0;
}

/* ================================================================ E: printSamEntry (src/LordFAST.cpp:318-459) */
static void intv_info(const struct lf_index *ix, uint32_t pos, uint32_t posEnd, const char **name, uint32_t *cbeg)
{
    int rid = pos2rid(ix, (int64_t)(((uint64_t)pos + (uint64_t)posEnd) >> 1));
    *cbeg = (uint32_t)((uint64_t)pos - (uint64_t)ix->contigs[rid].offset);
    *name = ix->contigs[rid].name;
}

static void sam_line(str_t *o, const ctx_t *cx, const rd_t *r, const sam_t *s, int flag, const char *rname, uint32_t rstart, int mapq)
{
    str_puts(o, r->name); str_putc(o, '\t'); str_puti(o, flag); str_putc(o, '\t'); str_puts(o, rname); str_putc(o, '\t');
    str_putu(o, rstart + 1); str_putc(o, '\t'); str_puti(o, mapq >= 0 ? mapq : 0); str_putc(o, '\t');
    str_puts(o, s->cigar); str_puts(o, "\t*\t0\t0\t");
    if (s->flag & 16) { str_put_rc(o, r->seq, r->len); str_putc(o, '\t'); str_put_rev(o, r->qual, r->isFq ? r->len : 1); }
    else { str_putn(o, r->seq, r->len); str_putc(o, '\t'); str_puts(o, r->qual); }
    str_puts(o, "\tAS:i:"); str_puti(o, s->alnScore); str_puts(o, "\tXS:i:0\tNM:i:"); str_puti(o, abs(s->nmCount));
    str_puts(o, "\tMD:Z:"); str_puts(o, s->md);
    if (cx->p->read_group_id[0]) { str_puts(o, "\tRG:Z:"); str_puts(o, cx->p->read_group_id); }
}

void print_sam_entry(ctx_t *cx, rd_t *r, int num)
{
    str_t *o = &r->out;
    rd_host_bases(cx, r, &cx->arena[0]);      /* called from the lane driver's serial loop (or, host SAM path, never in device-input mode) */
    const samlist_t *mp = r->maps;
    const int readLen = (int)r->len, maxWin = cx->p->max_map;
    const double bestEdit = (num > 0 ? (double)(-1 * mp[0].totalScore) / readLen : 1);
    const double mapqPortion = 50.0 / (maxWin - 1);
    int x1 = 0, x2 = 0;
    for (int i = 0; i < num; i++) if (mp[i].n > 0) { x1++; if ((double)(-1 * mp[i].totalScore) / readLen * 0.95 < bestEdit) x2++; }
    const double mapq = (x2 > 1 ? 2.1 : (maxWin - x1) * mapqPortion);
    int32_t mapq_int;
    for (int i = 0; i < num; i++) {
        if (i == 0) {
            if (mp[0].n > 0) {
                const double e0 = (double)(-1 * mp[0].totalScore) / readLen;
                if (num == 1 || (num > 1 && e0 < 0.15 && e0 < 0.95 * (double)(-1 * mp[1].totalScore) / readLen)) mapq_int = 60;
                else mapq_int = (int32_t)(mapq + 5 * (0.2 - e0) / 0.2);
                const int ns = mp[0].n;
                str_t *sa = (str_t *)calloc((size_t)ns, sizeof(str_t));
                const char **rn = (const char **)calloc((size_t)ns, sizeof(char *));
                uint32_t *rs = (uint32_t *)calloc((size_t)ns, sizeof(uint32_t));
                for (int j = 0; j < ns; j++) {
                    const sam_t *s = &mp[0].v[j];
                    intv_info(cx->ix, s->pos, s->posEnd, &rn[j], &rs[j]);
                    if (ns > 1) {
                        str_init(&sa[j]);
                        str_puts(&sa[j], rn[j]); str_putc(&sa[j], ','); str_putu(&sa[j], rs[j] + 1); str_putc(&sa[j], ',');
                        str_puts(&sa[j], (s->flag & 16) ? "-," : "+,"); str_puts(&sa[j], s->cigar); str_putc(&sa[j], ',');
                        str_puti(&sa[j], mapq_int); str_putc(&sa[j], ','); str_puti(&sa[j], abs(s->nmCount)); str_putc(&sa[j], ';');
                    }
                }
                for (int j = 0; j < ns; j++) {
                    const sam_t *s = &mp[0].v[j];
                    sam_line(o, cx, r, s, j > 0 ? (s->flag | 2048) : s->flag, rn[j], rs[j], mapq_int);
                    if (ns > 1) { str_puts(o, "\tSA:Z:"); for (int z = 0; z < ns; z++) if (z != j) str_putn(o, sa[z].s, sa[z].n); }
                    str_putc(o, '\n');
                }
                if (ns > 1) for (int j = 0; j < ns; j++) free(sa[j].s);
                free(sa); free(rn); free(rs);
            } else {
                str_puts(o, r->name); str_puts(o, "\t4\t*\t0\t0\t*\t*\t0\t0\t"); str_putn(o, r->seq, r->len); str_putc(o, '\t'); str_puts(o, r->qual);
                if (cx->p->read_group_id[0]) { str_puts(o, "\tRG:Z:"); str_puts(o, cx->p->read_group_id); }
                str_putc(o, '\n');
            }
        } else if (mp[i].n > 0) {
            mapq_int = (int32_t)(mapq + 5 * (0.2 - (double)(-1 * mp[i].totalScore) / readLen) / 0.2);
            for (int j = 0; j < mp[i].n; j++) {
                const sam_t *s = &mp[i].v[j];
                const char *rn; uint32_t rs;
                intv_info(cx->ix, s->pos, s->posEnd, &rn, &rs);
                sam_line(o, cx, r, s, s->flag | 256, rn, rs, mapq_int);
                str_putc(o, '\n');
            }
        }
    }
}

/* ---- the same decisions as print_sam_entry, as 48-byte line descriptors for lf_sam.hip (which writes the text) ---- */
typedef struct {
    lf_samline_t *ln; int *rd; int n, cap;  /* rd: the read (index in the chunk) a line belongs to */
    char *blob; uint64_t nb, capb;          /* SA:Z values and literal lines */
    char *names; uint64_t nn, capn;
    int cur_rd;
} linevec_t;
static lf_samline_t *lv_line(linevec_t *v)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8192; v->ln = (lf_samline_t *)realloc(v->ln, (size_t)v->cap * sizeof(lf_samline_t)); v->rd = (int *)realloc(v->rd, (size_t)v->cap * sizeof(int)); }
    v->rd[v->n] = v->cur_rd;
    lf_samline_t *l = &v->ln[v->n++]; memset(l, 0, sizeof *l);
    return l;
}
static uint64_t lv_blob(linevec_t *v, const char *s, size_t n)
{
    if (v->nb + n + 1 > v->capb) { v->capb = (v->nb + n + 1) * 2 + 4096; v->blob = (char *)realloc(v->blob, v->capb); }
    memcpy(v->blob + v->nb, s, n); v->nb += n;
    return v->nb - n;
}
static size_t rec_index(const ctx_t *cx, const sam_t *s) { return s->rtid == -2 ? (size_t)s->rec : (size_t)cx->rrbase[s->rtid] + (size_t)s->rec; }
/* CIGAR of a record as a C string (SA:Z values of split alignments need a few of them on the host) */
static char *rec_cigar_host(ctx_t *cx, const sam_t *s, arena_t *ar)
{
    if (s->cigar) return s->cigar;
    const size_t g = rec_index(cx, s);
    const uint32_t len = cx->rlens[2 * g];                                  /* incl. NUL */
    char *buf = (char *)ar_alloc(ar, (size_t)len + 1);
    if (lfg_fetch(cx->ix->device, buf, (const char *)cx->rtext_dev.d_text + cx->roffs[2 * g], len) != LF_OK) buf[0] = 0;
    buf[len ? len - 1 : 0] = 0;
    return buf;
}
static int rid_of_record(const struct lf_index *ix, const sam_t *s, uint32_t *cbeg)
{
    const int rid = pos2rid(ix, (int64_t)(((uint64_t)s->pos + (uint64_t)s->posEnd) >> 1));     /* bwt_get_intv_info: contig of the midpoint */
    *cbeg = (uint32_t)((uint64_t)s->pos - (uint64_t)ix->contigs[rid].offset);
    return rid;
}
static void line_mapped(ctx_t *cx, linevec_t *v, const rd_t *r, uint32_t name_off, const sam_t *s, int flag, int rid, uint32_t rstart, int mapq, uint32_t sa_off, uint32_t sa_len)
{
    lf_samline_t *l = lv_line(v);
    l->kind = LF_SL_MAPPED; l->name_off = name_off; l->name_len = (uint16_t)strlen(r->name); l->flag = (uint16_t)flag;
    l->read = (uint32_t)r->seed_idx; l->rname = rid; l->pos1 = rstart + 1; l->mapq = mapq >= 0 ? mapq : 0;
    l->as = s->alnScore; l->nm = (uint32_t)abs(s->nmCount); l->rec = (uint32_t)rec_index(cx, s);
    l->sa_off = sa_off; l->sa_len = sa_len; l->is_fq = (uint8_t)r->isFq;
}
static void lines_sam_entry(ctx_t *cx, linevec_t *v, rd_t *r, uint32_t name_off, int num, arena_t *ar)
{   /* src/LordFAST.cpp:318-459; the arithmetic is print_sam_entry's, expression by expression */
    const samlist_t *mp = r->maps;
    const int readLen = (int)r->len, maxWin = cx->p->max_map;
    const double bestEdit = (num > 0 ? (double)(-1 * mp[0].totalScore) / readLen : 1);
    const double mapqPortion = 50.0 / (maxWin - 1);
    int x1 = 0, x2 = 0;
    for (int i = 0; i < num; i++) if (mp[i].n > 0) { x1++; if ((double)(-1 * mp[i].totalScore) / readLen * 0.95 < bestEdit) x2++; }
    const double mapq = (x2 > 1 ? 2.1 : (maxWin - x1) * mapqPortion);
    int32_t mapq_int;
    for (int i = 0; i < num; i++) {
        if (i == 0) {
            if (mp[0].n > 0) {
                const double e0 = (double)(-1 * mp[0].totalScore) / readLen;
                if (num == 1 || (num > 1 && e0 < 0.15 && e0 < 0.95 * (double)(-1 * mp[1].totalScore) / readLen)) mapq_int = 60;
                else mapq_int = (int32_t)(mapq + 5 * (0.2 - e0) / 0.2);
                const int ns = mp[0].n;
                if (ns == 1) {
                    uint32_t rs; const int rid = rid_of_record(cx->ix, &mp[0].v[0], &rs);
                    line_mapped(cx, v, r, name_off, &mp[0].v[0], mp[0].v[0].flag, rid, rs, mapq_int, 0, 0);
                } else {
                    /* split alignment: every record carries the others in SA:Z (rname,pos,strand,CIGAR,mapQ,NM;) (:358-373) */
                    str_t *sa = (str_t *)calloc((size_t)ns, sizeof(str_t));
                    int *rid = (int *)calloc((size_t)ns, sizeof(int)); uint32_t *rs = (uint32_t *)calloc((size_t)ns, sizeof(uint32_t));
                    for (int j = 0; j < ns; j++) {
                        const sam_t *s = &mp[0].v[j];
                        rid[j] = rid_of_record(cx->ix, s, &rs[j]);
                        str_init(&sa[j]);
                        str_puts(&sa[j], cx->ix->contigs[rid[j]].name); str_putc(&sa[j], ','); str_putu(&sa[j], rs[j] + 1); str_putc(&sa[j], ',');
                        str_puts(&sa[j], (s->flag & 16) ? "-," : "+,"); str_puts(&sa[j], rec_cigar_host(cx, s, ar)); str_putc(&sa[j], ',');
                        str_puti(&sa[j], mapq_int); str_putc(&sa[j], ','); str_puti(&sa[j], abs(s->nmCount)); str_putc(&sa[j], ';');
                    }
                    for (int j = 0; j < ns; j++) {
                        const sam_t *s = &mp[0].v[j];
                        uint64_t o0 = v->nb; uint32_t ln = 0;
                        for (int z = 0; z < ns; z++) if (z != j) { const uint64_t o = lv_blob(v, sa[z].s, sa[z].n); if (!ln) o0 = o; ln += (uint32_t)sa[z].n; }
                        line_mapped(cx, v, r, name_off, s, j > 0 ? (s->flag | 2048) : s->flag, rid[j], rs[j], mapq_int, (uint32_t)o0, ln);
                    }
                    for (int j = 0; j < ns; j++) free(sa[j].s);
                    free(sa); free(rid); free(rs);
                }
            } else {
                lf_samline_t *l = lv_line(v);
                l->kind = LF_SL_UNMAPPED; l->name_off = name_off; l->name_len = (uint16_t)strlen(r->name); l->read = (uint32_t)r->seed_idx; l->is_fq = (uint8_t)r->isFq;
            }
        } else if (mp[i].n > 0) {
            mapq_int = (int32_t)(mapq + 5 * (0.2 - (double)(-1 * mp[i].totalScore) / readLen) / 0.2);
            for (int j = 0; j < mp[i].n; j++) {
                const sam_t *s = &mp[i].v[j];
                uint32_t rs; const int rid = rid_of_record(cx->ix, s, &rs);
                line_mapped(cx, v, r, name_off, s, s->flag | 256, rid, rs, mapq_int, 0, 0);
            }
        }
    }
}

/* the SAM stage of a chunk on the device path: one 48-byte line descriptor per record (flags, MAPQ, SA:Z decided here, the text
 * written by lf_sam.hip); HOLES mode also lists where the host puts SEQ / QUAL (cx->fill) */
int sam_stage_dev(ctx_t *cx)
{
    const int n = cx->n_reads;
    int rc = LF_OK;
    /* SAM lines on the device: the host only says what is printed (48 bytes per line) */
    linevec_t V; memset(&V, 0, sizeof V);
    arena_t *ar = &cx->arena[0];
    int any_fq = 0;
    for (int i = 0; i < n; i++) {
        rd_t *r = &cx->reads[i];
        V.cur_rd = i;
        const size_t nl = strlen(r->name);
        if (V.nn + nl + 1 > V.capn) { V.capn = (V.nn + nl + 1) * 2 + 65536; V.names = (char *)realloc(V.names, V.capn); }
        memcpy(V.names + V.nn, r->name, nl);
        const uint32_t name_off = (uint32_t)V.nn; V.nn += nl;
        if (r->mode == 0) {                 /* shorter than -l: not in the resident batch; the whole line is literal text */
            str_init(&r->out); print_sam_entry(cx, r, 1);
            lf_samline_t *l = lv_line(&V); l->kind = LF_SL_LITERAL; l->sa_off = (uint32_t)lv_blob(&V, r->out.s, r->out.n); l->sa_len = (uint32_t)r->out.n;
            free(r->out.s); memset(&r->out, 0, sizeof r->out);
            continue;
        }
        any_fq |= r->isFq;
        const int num = r->mode == 3 ? r->nWins : 1;
        int host_strings = nl > 65535;       /* a name longer than the line descriptor's 16-bit length, or a record whose CIGAR / MD were built per base on the host (the reference's misaligned-MD branch) */
        if (r->mode >= 2) for (int w = 0; w < num; w++) for (int j = 0; j < r->maps[w].n; j++) host_strings |= r->maps[w].v[j].rec < 0;
        if (host_strings) {
            /* rare: print the whole entry on the host (its other records' text is fetched from the device) */
            for (int w = 0; w < num; w++) for (int j = 0; j < r->maps[w].n; j++) {
                sam_t *sr = &r->maps[w].v[j];
                if (sr->rec < 0) continue;
                const size_t g = rec_index(cx, sr);
                sr->cigar = rec_cigar_host(cx, sr, ar);
                const uint32_t ml = cx->rlens[2 * g + 1];
                sr->md = (char *)ar_alloc(ar, (size_t)ml + 1);
                if (lfg_fetch(cx->ix->device, sr->md, (const char *)cx->rtext_dev.d_text + cx->roffs[2 * g + 1], ml) != LF_OK) sr->md[0] = 0;
                sr->md[ml ? ml - 1 : 0] = 0;
            }
            str_init(&r->out); print_sam_entry(cx, r, num);
            lf_samline_t *l = lv_line(&V); l->kind = LF_SL_LITERAL; l->sa_off = (uint32_t)lv_blob(&V, r->out.s, r->out.n); l->sa_len = (uint32_t)r->out.n;
            free(r->out.s); memset(&r->out, 0, sizeof r->out);
            continue;
        }
        lines_sam_entry(cx, &V, r, name_off, num, ar);
    }
    if (V.nb >= 0xffffffffull || V.nn >= 0xffffffffull) { free(V.ln); free(V.rd); free(V.blob); free(V.names); lf_set_error("lf_map_batch: chunk too large for the SAM writer"); return LF_ERR_ARG; }
    char *qcat = NULL; uint64_t qbytes = 0; int n_batch = 0;
    for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) n_batch++;
    if (any_fq && cx->d_quals) { for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) qbytes += cx->reads[i].len; }
    else if (any_fq && cx->holes) qcat = (char *)"";      /* the host prints the qualities itself: the device only has to know that there are some */
    else if (any_fq) {                            /* FASTQ: the qualities in the layout of the resident read batch */
        for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) qbytes += cx->reads[i].len;
        qcat = (char *)malloc(qbytes + 1);
        uint64_t o = 0;
        for (int i = 0; i < n; i++) { const rd_t *r = &cx->reads[i]; if ((int)r->len < cx->p->min_read_len) continue; if (r->isFq) memcpy(qcat + o, r->qual, r->len); else memset(qcat + o, '*', r->len); o += r->len; }
    }
    const uint64_t *h_offs = NULL; const uint32_t *h_hole = NULL;
    rc = lfg_sam_build(cx->ix, cx->p, V.n, V.ln, V.names, V.nn, V.blob, V.nb, qcat, qbytes, (any_fq && cx->d_quals) ? cx->d_quals : NULL, n_batch,
                       &cx->rtext_dev, cx->sam_parity, cx->holes, &cx->sam_total, &h_offs, &h_hole);
    cx->fill = NULL; cx->n_fill = 0;
    if (rc == LF_OK && cx->holes && V.n > 0) {
        /* where the host puts SEQ (/ QUAL): one entry per line that has a hole */
        cx->fill = (fill_t *)malloc(((size_t)V.n + 1) * sizeof(fill_t));
        for (int k = 0; k < V.n; k++) {
            if (!h_hole[2 * (size_t)k + 1]) continue;
            const rd_t *r = &cx->reads[V.rd[k]];
            fill_t *f = &cx->fill[cx->n_fill++];
            f->pos = h_offs[k] + h_hole[2 * (size_t)k]; f->seq = r->seq; f->qual = r->qual; f->len = r->len;
            f->rev = (uint8_t)(V.ln[k].kind == LF_SL_MAPPED && (V.ln[k].flag & 16)); f->fq = (uint8_t)(h_hole[2 * (size_t)k + 1] > r->len);
        }
    }
    free(V.ln); free(V.rd); free(V.blob); free(V.names); if (!(any_fq && cx->holes)) free(qcat);
    if (rc != LF_OK) return rc;
    return LF_OK;
}

/* HOLES mode: SEQ (/ QUAL) of line k goes to out_base + fill[k].pos -- the strings printSamEntry prints (src/LordFAST.cpp:377-402):
 * the read as given, or its reverse complement / reversed qualities for a record on the reverse strand (:501-502) */
void phase_fill(ctx_t *cx, int tid, int k)
{
    (void)tid;
    const fill_t *f = &cx->fill[k];
    char *d = cx->out_base + f->pos;
    if (!f->rev) lf_copy_stream(d, f->seq, f->len); else rc_copy_stream(d, f->seq, f->len);
    if (f->fq) {
        d[f->len] = '\t';
        char *q = d + f->len + 1;
        if (!f->rev) lf_copy_stream(q, f->qual, f->len); else for (uint32_t i = 0; i < f->len; i++) q[i] = f->qual[f->len - 1 - i];
    }
}

/* printSamHeader (src/BWT.cpp:668-681) */
char *lf_sam_header(const lf_index_t *ix, const lf_params_t *p, const char *cmdline)
{
    str_t sb; str_init(&sb);
    str_puts(&sb, "@HD\tVN:1.5\tSO:unsorted\n");
    for (int i = 0; i < ix->n_seqs; i++) { str_puts(&sb, "@SQ\tSN:"); str_puts(&sb, ix->contigs[i].name); str_puts(&sb, "\tLN:"); str_puti(&sb, ix->contigs[i].len); str_putc(&sb, '\n'); }
    if (p && p->read_group_id[0] && p->read_group[0]) { str_puts(&sb, p->read_group); str_putc(&sb, '\n'); }      /* src/BWT.cpp:676-679 */
    str_puts(&sb, "@PG\tID:lordfast\tPN:lordfast\tVN:0.0.10\tCL:"); str_puts(&sb, cmdline ? cmdline : ""); str_putc(&sb, '\n');
    return sb.s;
}
