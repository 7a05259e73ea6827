/*
 * lfo_map.c -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * mapSeq and everything below it (src/LordFAST.cpp:318-2258), restated in plain C.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <pthread.h>
#include "lf_oracle.h"
#include "lfo_internal.h"

/* constants of src/LordFAST.cpp:78-92 */
#define CLIP_LEN   500
#define CLIP_SIM   0.75
#define SPLIT_LEN  80
#define SPLIT_SIM  0.40
#define REVERSE_SIM 0.60

/* ---------- small containers ---------- */
typedef struct { char *buf; size_t cap, beg, end; } cdq_t;        /* std::deque<char> stand-in */

static void cdq_init(cdq_t *d) { d->cap = 1 << 16; d->buf = (char *)malloc(d->cap); d->beg = d->end = d->cap / 2; }
static void cdq_free(cdq_t *d) { free(d->buf); }
static void cdq_clear(cdq_t *d) { d->beg = d->end = d->cap / 2; }
static size_t cdq_size(const cdq_t *d) { return d->end - d->beg; }
static void cdq_room(cdq_t *d, size_t front, size_t back)
{
    if (d->beg >= front && d->cap - d->end >= back) return;
    size_t n = cdq_size(d), ncap = (n + front + back) * 2 + 1024;
    char *nb = (char *)malloc(ncap);
    size_t nbeg = front + (ncap - n - front - back) / 2;
    memcpy(nb + nbeg, d->buf + d->beg, n);
    free(d->buf);
    d->buf = nb; d->cap = ncap; d->beg = nbeg; d->end = nbeg + n;
}
static void cdq_push_back(cdq_t *d, char c) { cdq_room(d, 0, 1); d->buf[d->end++] = c; }
static void cdq_push_front(cdq_t *d, char c) { cdq_room(d, 1, 0); d->buf[--d->beg] = c; }
static void cdq_back_n(cdq_t *d, size_t n, char c) { cdq_room(d, 0, n); memset(d->buf + d->end, c, n); d->end += n; }
static void cdq_front_n(cdq_t *d, size_t n, char c) { cdq_room(d, n, 0); d->beg -= n; memset(d->buf + d->beg, c, n); }

typedef struct { char *s; size_t n, cap; } sbuf_t;
static void sb_init(sbuf_t *b) { b->cap = 1 << 12; b->s = (char *)malloc(b->cap); b->n = 0; b->s[0] = 0; }
static void sb_room(sbuf_t *b, size_t extra)
{
    if (b->n + extra + 1 <= b->cap) return;
    while (b->n + extra + 1 > b->cap) b->cap *= 2;
    b->s = (char *)realloc(b->s, b->cap);
}
static void sb_puts(sbuf_t *b, const char *s) { size_t l = strlen(s); sb_room(b, l); memcpy(b->s + b->n, s, l + 1); b->n += l; }
static void sb_putc(sbuf_t *b, char c) { sb_room(b, 1); b->s[b->n++] = c; b->s[b->n] = 0; }
static void sb_printf(sbuf_t *b, const char *fmt, ...)
{
    char tmp[64];
    va_list ap; va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    sb_puts(b, tmp);
}

/* ---------- data model (src/LordFAST.h:30-118) ---------- */
typedef struct { uint32_t tStart, tEnd; uint8_t isReverse; float score; } win_t;
typedef struct { int readIdx; uint32_t cnt; } wincount_t;
typedef struct {
    uint32_t qStart, qEnd, pos, posEnd;
    uint16_t flag;
    int32_t alnScore, nmCount;
    char *cigar, *md;
} sam_t;
typedef struct { sam_t *v; int n, cap; int32_t totalScore; } samlist_t;

static void samlist_clear(samlist_t *l)
{
    for (int i = 0; i < l->n; i++) { free(l->v[i].cigar); free(l->v[i].md); }
    l->n = 0;
}
static void samlist_push(samlist_t *l, const sam_t *s)
{
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 4; l->v = (sam_t *)realloc(l->v, (size_t)l->cap * sizeof(sam_t)); }
    l->v[l->n] = *s;
    l->v[l->n].cigar = strdup(s->cigar);
    l->v[l->n].md = strdup(s->md);
    l->n++;
}

typedef struct {
    const lfo_index_t *ix;
    const lfo_params_t *p;
    int chunkSize;
    wincount_t *winCnt; uint32_t refWinNum;
    lfo_seed_t *F, *R, *sel, *chain;
    uint32_t nF, nR, nSel, chainLen; float chainScore; int64_t selLow;
    win_t *topWins; int nWins;
    samlist_t *mappings;
} ctx_t;

static int win_less(const void *a, const void *b, void *c) { (void)c; return ((const win_t *)a)->score > ((const win_t *)b)->score; }        /* compareWin :981-984 */
static int sam_less(const void *a, const void *b, void *c) { (void)c; return ((const samlist_t *)a)->totalScore > ((const samlist_t *)b)->totalScore; } /* compareSam :986-992 */

/* reverseComplement / reverse (src/Common.cpp:58-66,93-101): case kept, non-ACGT -> 'N' */
static void revcomp(const char *seq, char *out, int len)
{
    for (int i = 0; i < len; i++) {
        char c = seq[len - 1 - i], r;
        switch (c) {
        case 'A': r = 'T'; break; case 'C': r = 'G'; break; case 'G': r = 'C'; break; case 'T': r = 'A'; break;
        case 'a': r = 't'; break; case 'c': r = 'g'; break; case 'g': r = 'c'; break; case 't': r = 'a'; break;
        default: r = 'N';
        }
        out[i] = r;
    }
    out[len] = 0;
}
static void reverse_str(const char *s, char *out, int len) { for (int i = 0; i < len; i++) out[i] = s[len - 1 - i]; out[len] = 0; }

/* ---------- window voting (src/LordFAST.cpp:582-657, 819-904) ---------- */
static void vote(ctx_t *cx, uint32_t L, const lfo_seed_t *s, uint32_t n, int readIdx)
{
    wincount_t *w = cx->winCnt;
    for (uint32_t i = 0; i < n; i++) {
        int32_t winId = (int32_t)(s[i].tPos / L);
        int32_t weight = 1 + ((int32_t)s[i].len - cx->p->min_anchor_len);
        if (w[winId].readIdx == readIdx) w[winId].cnt += (uint32_t)weight;
        else { w[winId].readIdx = readIdx; w[winId].cnt = (uint32_t)weight; }
        if (winId - 1 >= 0) {
            if (w[winId - 1].readIdx == readIdx) w[winId - 1].cnt += (uint32_t)weight;
            else { w[winId - 1].readIdx = readIdx; w[winId - 1].cnt = (uint32_t)weight; }
        }
    }
}

static int is_local_max(const ctx_t *cx, int i, int readIdx)
{
    const wincount_t *w = cx->winCnt;
    return w[i].readIdx == readIdx &&
           (i == 0 || w[i - 1].readIdx != readIdx || w[i].cnt >= w[i - 1].cnt) &&
           (i == (int)cx->refWinNum - 1 || w[i + 1].readIdx != readIdx || w[i].cnt > w[i + 1].cnt);
}

static void top_push(ctx_t *cx, uint32_t i, uint32_t L, float score, int isRev)
{
    int maxWin = cx->p->max_map;
    win_t *l = cx->topWins;
    if (cx->nWins < maxWin) {
        l[cx->nWins].tStart = i * L; l[cx->nWins].tEnd = (i + 2) * L - 1;
        l[cx->nWins].score = score; l[cx->nWins].isReverse = (uint8_t)isRev;
        cx->nWins++;
        lfo_push_heap(l, (size_t)cx->nWins, sizeof(win_t), win_less, NULL);
    } else if (score > l[0].score) {
        lfo_pop_heap(l, (size_t)cx->nWins, sizeof(win_t), win_less, NULL);
        win_t *b = &l[cx->nWins - 1];
        b->tStart = i * L; b->tEnd = (i + 2) * L - 1; b->score = score; b->isReverse = (uint8_t)isRev;
        lfo_push_heap(l, (size_t)cx->nWins, sizeof(win_t), win_less, NULL);
    }
}

static int win_limit(const ctx_t *cx, uint32_t L)
{
    int lim = (int)((uint32_t)cx->ix->l_pac / L + 2);
    if (lim > (int)cx->refWinNum) lim = (int)cx->refWinNum;
    return lim;
}

static void top_wins_coarse(ctx_t *cx, uint32_t L, const lfo_seed_t *s, uint32_t n, int isRev, int readIdx)
{
    vote(cx, L, s, n, readIdx);
    int lim = win_limit(cx, L);
    for (int i = 0; i < lim; i++)
        if (is_local_max(cx, i, readIdx))
            /* the reference compares the u32 count with the float heap top (:644) and stores it as float (:638) */
            top_push(cx, (uint32_t)i, L, (float)cx->winCnt[i].cnt, isRev);
}

static void select_seeds(ctx_t *cx, uint32_t L, uint32_t tStart, uint32_t tEnd, int isRev)
{
    /* src/LordFAST.cpp:995-1018 (alignWin) == :659-680 (calcChainScore) */
    uint32_t margin = L >> 1, cb, ce;
    lfo_chr_boundaries(cx->ix, tStart, tEnd, &cb, &ce);
    int64_t lo = ((int64_t)tStart - (int64_t)margin > (int64_t)cb) ? (int64_t)tStart - (int64_t)margin : (int64_t)cb;
    int64_t hi = ((int64_t)tEnd + (int64_t)margin < (int64_t)ce) ? (int64_t)tEnd + (int64_t)margin : (int64_t)ce;
    const lfo_seed_t *s = isRev ? cx->R : cx->F;
    uint32_t n = isRev ? cx->nR : cx->nF;
    cx->nSel = 0;
    for (uint32_t i = 0; i < n; i++)
        if ((int64_t)s[i].tPos >= lo && (int64_t)s[i].tPos <= hi) cx->sel[cx->nSel++] = s[i];
    cx->selLow = lo;
}

static void chain_selected(ctx_t *cx)
{
    if (cx->p->chain_alg != 0) {
        /* src/LordFAST.cpp:682-692, 1028-1046: clasp keeps positions in `int`, so windows above 2e9 are shifted */
        int shift = cx->selLow > 2000000000;
        if (shift) for (uint32_t i = 0; i < cx->nSel; i++) cx->sel[i].tPos -= 2000000000u;
        lfo_chain_clasp(cx->sel, cx->nSel, cx->chain, &cx->chainLen, &cx->chainScore);
        if (shift) for (uint32_t i = 0; i < cx->chainLen; i++) cx->chain[i].tPos += 2000000000u;
        return;
    }
    lfo_chain_n2(cx->p, cx->sel, cx->nSel, cx->chain, &cx->chainLen, &cx->chainScore);
}

static void top_wins_fine(ctx_t *cx, uint32_t L, const lfo_seed_t *s, uint32_t n, int isRev, int readIdx, float minScore)
{
    vote(cx, L, s, n, readIdx);
    int lim = win_limit(cx, L);
    for (int i = 0; i < lim; i++) {
        if (cx->winCnt[i].cnt > minScore && is_local_max(cx, i, readIdx)) {   /* :875 u32 -> float compare */
            select_seeds(cx, L, (uint32_t)i * L, ((uint32_t)i + 2) * L - 1, isRev);
            chain_selected(cx);                                               /* calcChainScore :659-732 */
            top_push(cx, (uint32_t)i, L, cx->chainScore, isRev);
        }
    }
}

/* ---------- CIGAR / MD helpers (src/LordFAST.cpp:1570-1763) ---------- */
static const char OP2CH[4] = { 'M', 'I', 'D', 'M' };

static void cig_back(cdq_t *c, const uint8_t *ops, int n) { for (int i = 0; i < n; i++) cdq_push_back(c, OP2CH[ops[i]]); }
static void cig_front(cdq_t *c, const uint8_t *ops, int n) { for (int i = 0; i < n; i++) cdq_push_front(c, OP2CH[ops[i]]); }

static void md_back(cdq_t *md, const char *target, const uint8_t *ops, int n)
{
    int ti = 0;
    for (int i = 0; i < n; i++) {
        switch (ops[i]) {
        case 0: cdq_push_back(md, '='); ti++; break;
        case 1: cdq_push_back(md, '-'); break;
        default: cdq_push_back(md, target[ti]); ti++; break;       /* 2 (deletion) and 3 (mismatch) */
        }
    }
}

static char complement_upper(char c)
{   /* tableComplement of edlibMD_pushfront (:1677-1686): result is always upper case */
    switch (c) {
    case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C'; case 'T': case 't': return 'A';
    default: return 'N';
    }
}

static void md_front(cdq_t *md, const char *target, const uint8_t *ops, int n)
{
    int ti = 0;
    for (int i = 0; i < n; i++) {
        switch (ops[i]) {
        case 0: cdq_push_front(md, '='); ti++; break;
        case 1: cdq_push_front(md, '-'); break;
        default: cdq_push_front(md, complement_upper(target[ti])); ti++; break;
        }
    }
}

static char *cigar_to_string(const cdq_t *c)
{   /* edlibCigar_toString (:1596-1626): leading / trailing I-runs become S */
    sbuf_t sb; sb_init(&sb);
    char ch = 0; int num = 0, opn = 0;
    size_t n = cdq_size(c);
    for (size_t i = 0; i < n; i++) {
        char x = c->buf[c->beg + i];
        if (x != ch) {
            if (ch != 0) { sb_printf(&sb, "%d", num); sb_putc(&sb, (opn == 0 && ch == 'I') ? 'S' : ch); opn++; }
            num = 1; ch = x;
        } else num++;
    }
    if (num) { sb_printf(&sb, "%d", num); sb_putc(&sb, ch == 'I' ? 'S' : ch); }
    return sb.s;
}

static char *md_to_string(const cdq_t *md, const cdq_t *cg)
{   /* edlibMD_toString (:1717-1763) */
    sbuf_t sb; sb_init(&sb);
    int num = 0; char last = '=';
    size_t n = cdq_size(md);
    for (size_t i = 0; i < n; i++) {
        char m = md->buf[md->beg + i], c = cg->buf[cg->beg + i];
        if (m == '=') { num++; last = '='; }
        else if (m == '-') { last = 'I'; }
        else if (c == 'M') { sb_printf(&sb, "%d", num); num = 0; sb_putc(&sb, m); last = 'X'; }
        else if (c == 'D') {
            if (last != 'D') { sb_printf(&sb, "%d", num); num = 0; sb_putc(&sb, '^'); }
            sb_putc(&sb, m); last = 'D';
        }
    }
    sb_printf(&sb, "%d", num);
    return sb.s;
}

static void char2int(uint8_t *out, const char *s, int n)
{   /* _pf_char2int (:158-164) */
    for (int i = 0; i < n; i++) {
        switch (s[i]) {
        case 'A': case 'a': out[i] = 0; break; case 'C': case 'c': out[i] = 1; break;
        case 'G': case 'g': out[i] = 2; break; case 'T': case 't': out[i] = 3; break;
        default: out[i] = 4;
        }
    }
}
static void revcomp_int(uint8_t *dst, const uint8_t *src, int n) { for (int i = 0; i < n; i++) dst[i] = (uint8_t)(3 - src[n - i - 1]); }
static void pac2int(const lfo_index_t *ix, uint32_t beg, uint32_t len, uint8_t *out)
{
    for (uint32_t i = 0; i < len; i++) { uint32_t l = beg + i; out[i] = (ix->pac[l >> 2] >> ((~l & 3) << 1)) & 3; }
}

static void set_sam_strings(sam_t *s, const cdq_t *cg, const cdq_t *md)
{
    free(s->cigar); free(s->md);
    s->cigar = cigar_to_string(cg);
    s->md = md_to_string(md, cg);
}

/* ---------- alignChain_edlib (src/LordFAST.cpp:1765-2258) ---------- */
static void align_chain(ctx_t *cx, const lfo_seed_t *s, uint32_t chainLen, const char *query, int32_t readLen,
                        int isRev, samlist_t *map)
{
    const lfo_index_t *ix = cx->ix;
    size_t cap = (size_t)readLen * 4 + 4096;
    char *readAlnSeq = (char *)malloc(cap), *readAlnSeq_rev = (char *)malloc(cap);
    char *refAlnSeq = (char *)malloc(cap), *refAlnSeq_rev = (char *)malloc(cap);
    uint8_t *qk = (uint8_t *)malloc(cap), *tk = (uint8_t *)malloc(cap), *qk_rev = (uint8_t *)malloc(cap), *tk_rev = (uint8_t *)malloc(cap);
    uint8_t *ops = (uint8_t *)malloc(2 * cap), *ops2 = (uint8_t *)malloc(2 * cap);
    uint32_t readAlnStart, refAlnStart, readAlnEnd, refAlnEnd;
    int32_t readAlnLen, refAlnLen;
    int qle, tle, nops, nops2, endLoc, endLoc2, ed, ed2;
    int32_t editScore = 0;
    cdq_t cg, md; cdq_init(&cg); cdq_init(&md);
    sam_t tmp; memset(&tmp, 0, sizeof tmp);
    uint32_t chrBeg, chrEnd;
    uint32_t i;

    lfo_chr_boundaries(ix, s[0].tPos, s[chainLen - 1].tPos, &chrBeg, &chrEnd);          /* :1799 */
    tmp.flag = isRev ? 16 : 0;
    tmp.pos = s[0].tPos;
    tmp.qStart = s[0].qPos;

    /* ---- extend before the first seed (:1820-1899) ---- */
    readAlnLen = (int32_t)s[0].qPos;
    refAlnLen = readAlnLen + 20;
    if (readAlnLen > 0) {
        if ((int64_t)s[0].tPos - refAlnLen >= (int64_t)chrBeg) {
            revcomp(query, readAlnSeq, readAlnLen);
            refAlnStart = s[0].tPos - (uint32_t)refAlnLen;
            lfo_pac2char(ix, refAlnStart, (uint32_t)refAlnLen, refAlnSeq); refAlnSeq[refAlnLen] = 0;
            revcomp(refAlnSeq, refAlnSeq_rev, refAlnLen);
            ed = lfo_edlib(readAlnSeq, readAlnLen, refAlnSeq_rev, refAlnLen, 1, &endLoc, ops, &nops);
            int realigned = 0;
            if (readAlnLen > CLIP_LEN && (1 - ((float)ed / readAlnLen)) < CLIP_SIM) {
                char2int(qk_rev, query, readAlnLen);
                revcomp_int(qk, qk_rev, readAlnLen);
                pac2int(ix, refAlnStart, (uint32_t)refAlnLen, tk_rev);
                revcomp_int(tk, tk_rev, refAlnLen);
                lfo_ksw_extend2(readAlnLen, qk, refAlnLen, tk, 0, 1, 0, 1, 40, 40, readAlnLen, &qle, &tle);   /* ksw_extend :1848 */
                if (qle > 0 && qle < readAlnLen) {
                    ed = lfo_edlib(readAlnSeq, qle, refAlnSeq_rev, tle, 0, &endLoc, ops, &nops);
                    cig_front(&cg, ops, nops);
                    md_front(&md, refAlnSeq_rev, ops, nops);
                    editScore -= ed;
                    tmp.pos = s[0].tPos - (uint32_t)endLoc - 1;
                    tmp.qStart = s[0].qPos - (uint32_t)qle;
                    cdq_front_n(&cg, (size_t)(readAlnLen - qle), 'I');
                    cdq_front_n(&md, (size_t)(readAlnLen - qle), '-');
                    realigned = 1;
                }
            }
            if (!realigned) {
                editScore -= ed;
                cig_front(&cg, ops, nops);
                md_front(&md, refAlnSeq_rev, ops, nops);
                tmp.pos = s[0].tPos - (uint32_t)endLoc - 1;
                tmp.qStart = 0;
            }
        } else {                                     /* not enough reference left: soft clip */
            cdq_front_n(&cg, (size_t)readAlnLen, 'I');
            cdq_front_n(&md, (size_t)readAlnLen, '-');
        }
    }

    /* ---- between adjacent anchors (:1901-2137) ---- */
    int numAnchorsSoFar = 1;
    for (i = 0; i + 1 < chainLen; i++) {
        cdq_back_n(&cg, s[i].len, 'M');
        cdq_back_n(&md, s[i].len, '=');
        readAlnStart = s[i].qPos + s[i].len;
        refAlnStart = s[i].tPos + s[i].len;
        readAlnEnd = s[i + 1].qPos;
        refAlnEnd = s[i + 1].tPos;
        readAlnLen = (int32_t)(readAlnEnd - readAlnStart);
        refAlnLen = (int32_t)(refAlnEnd - refAlnStart);

        if (readAlnLen > 0 && refAlnLen > 0) {
            lfo_pac2char(ix, refAlnStart, (uint32_t)refAlnLen, refAlnSeq);
            ed = lfo_edlib(query + readAlnStart, readAlnLen, refAlnSeq, refAlnLen, 0, &endLoc, ops, &nops);
            int handled = 0;
            if (getenv("LFO_TRACE") && (readAlnLen > 300 || refAlnLen > 300))
                fprintf(stderr, "[trace] gap %u q[%u,%u) len %d t len %d ed %d sim %f\n", i, readAlnStart, readAlnEnd, readAlnLen, refAlnLen, ed, (1 - ((float)ed / readAlnLen)));
            if (abs(readAlnLen - refAlnLen) >= SPLIT_LEN && (1 - ((float)ed / readAlnLen)) < SPLIT_SIM) {
                uint32_t rs_new, ts_new, re_new, te_new;
                int32_t rl_new, tl_new;
                /* start of the potential split (:1967-1973) */
                char2int(qk, query + readAlnStart, readAlnLen);
                pac2int(ix, refAlnStart, (uint32_t)refAlnLen, tk);
                lfo_ksw_extend2(readAlnLen, qk, refAlnLen, tk, 8, 1, 4, 1, 100, 200, readAlnLen, &qle, &tle);
                rs_new = readAlnStart + (uint32_t)qle;
                ts_new = refAlnStart + (uint32_t)tle;
                /* end of the potential split (:1975-1983) */
                char2int(qk_rev, query + readAlnStart, readAlnLen);
                revcomp_int(qk, qk_rev, readAlnLen);
                pac2int(ix, refAlnStart, (uint32_t)refAlnLen, tk_rev);
                revcomp_int(tk, tk_rev, refAlnLen);
                lfo_ksw_extend2(readAlnLen, qk, refAlnLen, tk, 8, 1, 4, 1, 100, 200, readAlnLen, &qle, &tle);
                re_new = readAlnEnd - (uint32_t)qle;
                te_new = refAlnEnd - (uint32_t)tle;
                tl_new = (int32_t)(te_new - ts_new);
                rl_new = (int32_t)(re_new - rs_new);

                if (rs_new < re_new || ts_new < te_new) {                       /* extensions do not cross (:1995) */
                    handled = 1;
                    /* first part (:1998-2031) */
                    if (rs_new > readAlnStart || ts_new > refAlnStart) {
                        ed = lfo_edlib(query + readAlnStart, (int)(rs_new - readAlnStart), refAlnSeq, (int)(ts_new - refAlnStart), 0, &endLoc, ops, &nops);
                        cig_back(&cg, ops, nops);
                        md_back(&md, refAlnSeq, ops, nops);
                        editScore -= ed;
                    }
                    cdq_back_n(&cg, (size_t)((uint32_t)readLen - rs_new), 'I');
                    cdq_back_n(&md, (size_t)((uint32_t)readLen - rs_new), '-');
                    set_sam_strings(&tmp, &cg, &md);
                    tmp.posEnd = ts_new;
                    tmp.qEnd = rs_new;
                    tmp.nmCount = editScore;
                    if (numAnchorsSoFar > 1) samlist_push(map, &tmp);
                    cdq_clear(&cg); cdq_clear(&md);
                    editScore = 0;
                    /* middle part: does its reverse complement align better? (:2033-2077) */
                    if (rs_new < re_new && ts_new < te_new) {
                        lfo_pac2char(ix, ts_new, (uint32_t)tl_new, refAlnSeq);
                        ed = lfo_edlib(query + rs_new, rl_new, refAlnSeq, tl_new, 0, &endLoc, ops, &nops);
                        revcomp(query + rs_new, readAlnSeq_rev, rl_new);
                        ed2 = lfo_edlib(readAlnSeq_rev, rl_new, refAlnSeq, tl_new, 0, &endLoc2, ops2, &nops2);
                        if ((1 - ((double)ed2 / rl_new)) > (1 - ((double)ed / rl_new)) && (1 - ((double)ed2 / rl_new)) > REVERSE_SIM) {
                            tmp.flag = isRev ? 0 : 16;
                            tmp.pos = ts_new; tmp.qStart = rs_new; tmp.posEnd = te_new; tmp.qEnd = re_new;
                            cdq_back_n(&cg, rs_new, 'I');
                            cdq_back_n(&md, rs_new, '-');
                            cig_back(&cg, ops2, nops2);
                            md_back(&md, refAlnSeq, ops2, nops2);
                            editScore -= ed2;
                            cdq_back_n(&cg, (size_t)((uint32_t)readLen - re_new), 'I');
                            cdq_front_n(&md, (size_t)((uint32_t)readLen - re_new), '-');   /* sic: front, :2057 (App. B #3) */
                            set_sam_strings(&tmp, &cg, &md);
                            tmp.nmCount = editScore;
                            samlist_push(map, &tmp);
                            cdq_clear(&cg); cdq_clear(&md);
                            editScore = 0;
                        }
                    }
                    /* second part (:2079-2097) */
                    if (re_new < readAlnEnd || te_new < refAlnEnd) {
                        revcomp(query + readAlnStart, readAlnSeq, readAlnLen);
                        /* refAlnSeq may have been partly overwritten by the middle part, exactly as in the
                         * reference; only its untouched tail is consumed below */
                        refAlnSeq[refAlnLen] = 0;
                        revcomp(refAlnSeq, refAlnSeq_rev, refAlnLen);
                        ed = lfo_edlib(readAlnSeq, (int)(readAlnEnd - re_new), refAlnSeq_rev, (int)(refAlnEnd - te_new), 0, &endLoc, ops, &nops);
                        cig_front(&cg, ops, nops);
                        md_front(&md, refAlnSeq_rev, ops, nops);
                        editScore -= ed;
                    }
                    cdq_front_n(&cg, re_new, 'I');
                    cdq_front_n(&md, re_new, '-');
                    tmp.flag = isRev ? 16 : 0;
                    tmp.pos = te_new;
                    tmp.qStart = re_new;
                    numAnchorsSoFar = 0;
                }
            }
            if (!handled) {
                editScore -= ed;
                cig_back(&cg, ops, nops);
                md_back(&md, refAlnSeq, ops, nops);
            }
        } else if (readAlnLen > 0) {                 /* pure insertion (:2119-2125) */
            cdq_back_n(&cg, (size_t)readAlnLen, 'I');
            cdq_back_n(&md, (size_t)readAlnLen, '-');
            editScore -= readAlnLen;
        } else {                                     /* pure deletion (:2126-2134); refAlnLen may be <= 0 */
            if (refAlnLen > 0) {
                cdq_back_n(&cg, (size_t)refAlnLen, 'D');
                lfo_pac2char(ix, refAlnStart, (uint32_t)refAlnLen, refAlnSeq);
                for (int j = 0; j < refAlnLen; j++) cdq_push_back(&md, refAlnSeq[j]);
            }
            editScore -= refAlnLen;
        }
        numAnchorsSoFar++;
    }

    /* ---- last seed and extension after it (:2149-2230) ---- */
    cdq_back_n(&cg, s[i].len, 'M');
    cdq_back_n(&md, s[i].len, '=');
    tmp.posEnd = s[i].tPos + s[i].len - 1;
    tmp.qEnd = s[i].qPos + s[i].len - 1;
    readAlnStart = s[i].qPos + s[i].len;
    readAlnLen = readLen - (int32_t)readAlnStart;
    refAlnLen = readAlnLen + 20;
    if (readAlnLen > 0) {
        if (s[i].tPos + s[i].len + (uint32_t)refAlnLen - 1 <= chrEnd) {
            refAlnStart = s[i].tPos + s[i].len;
            lfo_pac2char(ix, refAlnStart, (uint32_t)refAlnLen, refAlnSeq);
            ed = lfo_edlib(query + readAlnStart, readAlnLen, refAlnSeq, refAlnLen, 1, &endLoc, ops, &nops);
            int realigned = 0;
            if (readAlnLen > CLIP_LEN && (1 - ((float)ed / readAlnLen)) < CLIP_SIM) {
                char2int(qk, query + readAlnStart, readAlnLen);
                pac2int(ix, refAlnStart, (uint32_t)refAlnLen, tk);
                lfo_ksw_extend2(readAlnLen, qk, refAlnLen, tk, 0, 1, 0, 1, 40, 40, readAlnLen, &qle, &tle);
                if (qle > 0 && qle < readAlnLen) {
                    ed = lfo_edlib(query + readAlnStart, qle, refAlnSeq, tle, 0, &endLoc, ops, &nops);
                    cig_back(&cg, ops, nops);
                    md_back(&md, refAlnSeq, ops, nops);
                    editScore -= ed;
                    tmp.posEnd = refAlnStart + (uint32_t)endLoc;
                    tmp.qEnd = readAlnStart + (uint32_t)qle;
                    cdq_back_n(&cg, (size_t)(readAlnLen - qle), 'I');
                    cdq_back_n(&md, (size_t)(readAlnLen - qle), '-');
                    realigned = 1;
                }
            }
            if (!realigned) {
                editScore -= ed;
                cig_back(&cg, ops, nops);
                md_back(&md, refAlnSeq, ops, nops);
                tmp.posEnd = refAlnStart + (uint32_t)endLoc;
                tmp.qEnd = (uint32_t)readLen;
            }
        } else {
            cdq_back_n(&cg, (size_t)readAlnLen, 'I');
            cdq_back_n(&md, (size_t)readAlnLen, '-');
        }
    }
    set_sam_strings(&tmp, &cg, &md);
    tmp.nmCount = editScore;
    samlist_push(map, &tmp);

    free(tmp.cigar); free(tmp.md);
    cdq_free(&cg); cdq_free(&md);
    free(readAlnSeq); free(readAlnSeq_rev); free(refAlnSeq); free(refAlnSeq_rev);
    free(qk); free(tk); free(qk_rev); free(tk_rev); free(ops); free(ops2);
}

/* ---------- alignWin (src/LordFAST.cpp:995-1189) ---------- */
static void align_win(ctx_t *cx, const win_t *win, const char *query, const char *query_rev, uint32_t rLen, samlist_t *map)
{
    select_seeds(cx, rLen, win->tStart, win->tEnd, win->isReverse);
    chain_selected(cx);
    if (cx->chainLen > 1) {
        align_chain(cx, cx->chain, cx->chainLen, win->isReverse ? query_rev : query, (int32_t)rLen, win->isReverse, map);
        map->totalScore = 0;
        for (int i = 0; i < map->n; i++) {
            map->v[i].alnScore = (int32_t)((uint32_t)map->v[i].nmCount + (map->v[i].qEnd - map->v[i].qStart));
            map->totalScore += map->v[i].nmCount;
        }
        /* forward strand uses the literal 0.15, reverse uses gapPenalty (:1162 vs :1077) */
        double gp = win->isReverse ? cx->p->gap_penalty : 0.15;
        for (int i = 0; i + 1 < map->n; i++) {
            int64_t a = (int64_t)map->v[i + 1].pos - (int64_t)map->v[i].posEnd;
            int64_t b = (int64_t)map->v[i + 1].qStart - (int64_t)map->v[i].qEnd;
            uint32_t diff = (uint32_t)((a < 0 ? -a : a) + (b < 0 ? -b : b));
            map->totalScore = (int32_t)((double)map->totalScore - gp * (double)diff);
        }
        map->totalScore = (int32_t)((uint32_t)map->totalScore - map->v[0].qStart);
        map->totalScore = (int32_t)((uint32_t)map->totalScore - (rLen - map->v[map->n - 1].qEnd));
    } else {
        map->totalScore = (int32_t)((uint32_t)-2 * rLen);
    }
}

/* ---------- printSamEntry (src/LordFAST.cpp:318-459) ---------- */
static void intv_info(const lfo_index_t *ix, uint32_t pos, uint32_t posEnd, const char **name, uint32_t *cbeg)
{   /* bwt_get_intv_info (src/BWT.cpp:636-651) */
    uint64_t mid = ((uint64_t)pos + (uint64_t)posEnd) >> 1;
    int rid = lfo_pos2rid(ix, (int64_t)mid);
    *cbeg = (uint32_t)((uint64_t)pos - (uint64_t)ix->contigs[rid].offset);
    *name = ix->contigs[rid].name;
}

typedef struct { const char *qName, *seq, *seq_rev, *qual, *qual_rev; } readinfo_t;

static void sam_record(sbuf_t *out, const ctx_t *cx, const readinfo_t *ri, const sam_t *s, int flag, const char *rname,
                       uint32_t rstart, int mapq_int)
{
    sb_puts(out, ri->qName); sb_putc(out, '\t');
    sb_printf(out, "%d", flag); sb_putc(out, '\t');
    sb_puts(out, rname); sb_putc(out, '\t');
    sb_printf(out, "%u", rstart + 1); sb_putc(out, '\t');
    sb_printf(out, "%d", mapq_int >= 0 ? mapq_int : 0); sb_putc(out, '\t');
    sb_puts(out, s->cigar); sb_puts(out, "\t*\t0\t0\t");
    sb_puts(out, (s->flag & 16) ? ri->seq_rev : ri->seq); sb_putc(out, '\t');
    sb_puts(out, (s->flag & 16) ? ri->qual_rev : ri->qual); sb_putc(out, '\t');
    sb_printf(out, "AS:i:%d", s->alnScore); sb_putc(out, '\t');
    sb_puts(out, "XS:i:0\t");
    sb_printf(out, "NM:i:%d", abs(s->nmCount)); sb_putc(out, '\t');
    sb_puts(out, "MD:Z:"); sb_puts(out, s->md);
    if (cx->p->read_group_id[0]) { sb_puts(out, "\tRG:Z:"); sb_puts(out, cx->p->read_group_id); }
}

static void print_sam_entry(sbuf_t *out, const ctx_t *cx, const readinfo_t *ri, int readLen, int num)
{
    const samlist_t *mp = cx->mappings;
    int maxWin = cx->p->max_map;
    double bestEdit = (num > 0 ? (double)(-1 * mp[0].totalScore) / readLen : 1);
    double mapqPortion = 50.0 / (maxWin - 1);
    int x1 = 0, x2 = 0;
    for (int i = 0; i < num; i++)
        if (mp[i].n > 0) { x1++; if ((double)(-1 * mp[i].totalScore) / readLen * 0.95 < bestEdit) x2++; }
    double mapq = (x2 > 1 ? 2.1 : (maxWin - x1) * mapqPortion);
    int32_t mapq_int;

    for (int i = 0; i < num; i++) {
        if (i == 0) {
            if (mp[0].n > 0) {
                double e0 = (double)(-1 * mp[0].totalScore) / readLen;
                if (num == 1 || (num > 1 && e0 < 0.15 && e0 < 0.95 * (double)(-1 * mp[1].totalScore) / readLen)) mapq_int = 60;
                else mapq_int = (int32_t)(mapq + 5 * (0.2 - e0) / 0.2);
                int ns = mp[0].n;
                char **sa = (char **)calloc((size_t)ns, sizeof(char *));
                const char **rn = (const char **)calloc((size_t)ns, sizeof(char *));
                uint32_t *rs = (uint32_t *)calloc((size_t)ns, sizeof(uint32_t));
                for (int j = 0; j < ns; j++) {
                    const sam_t *s = &mp[0].v[j];
                    intv_info(cx->ix, s->pos, s->posEnd, &rn[j], &rs[j]);
                    sbuf_t t; sb_init(&t);
                    sb_puts(&t, rn[j]); sb_putc(&t, ',');
                    sb_printf(&t, "%u", rs[j] + 1); sb_putc(&t, ',');
                    sb_puts(&t, (s->flag & 16) ? "-," : "+,");
                    sb_puts(&t, s->cigar); sb_putc(&t, ',');
                    sb_printf(&t, "%d", mapq_int); sb_putc(&t, ',');
                    sb_printf(&t, "%d", abs(s->nmCount)); sb_putc(&t, ';');
                    sa[j] = t.s;
                }
                for (int j = 0; j < ns; j++) {
                    const sam_t *s = &mp[0].v[j];
                    sam_record(out, cx, ri, s, j > 0 ? (s->flag | 2048) : s->flag, rn[j], rs[j], mapq_int);
                    if (ns > 1) {
                        sb_puts(out, "\tSA:Z:");
                        for (int z = 0; z < ns; z++) if (z != j) sb_puts(out, sa[z]);
                    }
                    sb_putc(out, '\n');
                }
                for (int j = 0; j < ns; j++) free(sa[j]);
                free(sa); free(rn); free(rs);
            } else {
                sb_puts(out, ri->qName); sb_puts(out, "\t4\t*\t0\t0\t*\t*\t0\t0\t");
                sb_puts(out, ri->seq); sb_putc(out, '\t'); sb_puts(out, ri->qual);
                if (cx->p->read_group_id[0]) { sb_puts(out, "\tRG:Z:"); sb_puts(out, cx->p->read_group_id); }
                sb_putc(out, '\n');
            }
        } else if (mp[i].n > 0) {
            mapq_int = (int32_t)(mapq + 5 * (0.2 - (double)(-1 * mp[i].totalScore) / readLen) / 0.2);
            for (int j = 0; j < mp[i].n; j++) {
                const sam_t *s = &mp[i].v[j];
                const char *rn; uint32_t rs;
                intv_info(cx->ix, s->pos, s->posEnd, &rn, &rs);
                sam_record(out, cx, ri, s, s->flag | 256, rn, rs, mapq_int);
                sb_putc(out, '\n');
            }
        }
    }
}

/* ---------- mapSeq for one read (src/LordFAST.cpp:461-580) ---------- */
static void map_one(ctx_t *cx, int t, const char *name, const char *seq, const char *qual_in, sbuf_t *out)
{
    uint32_t readLen = (uint32_t)strlen(seq);
    int isFq = (qual_in && qual_in[0]);
    const char *qual = isFq ? qual_in : "*";
    uint32_t qualLen = isFq ? readLen : 1;
    readinfo_t ri; ri.qName = name; ri.seq = seq; ri.qual = qual; ri.seq_rev = NULL; ri.qual_rev = NULL;

    if ((int)readLen < cx->p->min_read_len) {
        samlist_clear(&cx->mappings[0]);
        print_sam_entry(out, cx, &ri, (int)readLen, 1);
        return;
    }
    char *seq_rev = (char *)malloc((size_t)readLen + 1), *qual_rev = (char *)malloc((size_t)qualLen + 1);
    revcomp(seq, seq_rev, (int)readLen);
    reverse_str(qual, qual_rev, (int)qualLen);
    ri.seq_rev = seq_rev; ri.qual_rev = qual_rev;

    lfo_seed(cx->ix, cx->p, seq, readLen, cx->F, &cx->nF, cx->R, &cx->nR, NULL);
    cx->nWins = 0;
    /* chain_seeds_clasp on a window without seeds leaves the PREVIOUS call's chain in place (src/Chain.cpp:68,92) and alignWin
     * then extends that stale chain.  Inside a read the previous call is this read's previous window: deterministic, restated
     * here by keeping cx->chain between calls.  Across reads it is whatever the reference's thread mapped before -- scheduling
     * dependent under --threads > 1 -- so every read starts without a chain (the first window of a read comes out empty). */
    cx->chainLen = 0;
    top_wins_coarse(cx, readLen, cx->F, cx->nF, 0, t + 1);
    top_wins_coarse(cx, readLen, cx->R, cx->nR, 1, -(t + 1));

    if (cx->nWins == 0) {
        samlist_clear(&cx->mappings[0]);
        print_sam_entry(out, cx, &ri, (int)readLen, 1);
    } else {
        lfo_sort_heap(cx->topWins, (size_t)cx->nWins, sizeof(win_t), win_less, NULL);       /* :528 */
        const float scoreRatio = 4;
        /* with a single candidate the reference compares against a stale list[1] (App. B #1); both
         * branches then align the same window and emit the same record, so we take the coarse one */
        if (cx->nWins == 1 || cx->topWins[0].score >= scoreRatio * cx->topWins[1].score) {
            samlist_clear(&cx->mappings[0]);
            align_win(cx, &cx->topWins[0], seq, seq_rev, readLen, &cx->mappings[0]);
            print_sam_entry(out, cx, &ri, (int)readLen, 1);
        } else {
            float minScore = (float)cx->topWins[0].score / scoreRatio;
            cx->nWins = 0;
            top_wins_fine(cx, readLen, cx->F, cx->nF, 0, t + cx->chunkSize + 1, minScore);
            top_wins_fine(cx, readLen, cx->R, cx->nR, 1, -(t + cx->chunkSize + 1), minScore);
            for (int i = 0; i < cx->nWins; i++) {
                samlist_clear(&cx->mappings[i]);
                align_win(cx, &cx->topWins[i], seq, seq_rev, readLen, &cx->mappings[i]);
            }
            lfo_std_sort(cx->mappings, (size_t)cx->nWins, sizeof(samlist_t), sam_less, NULL);  /* :565 */
            print_sam_entry(out, cx, &ri, (int)readLen, cx->nWins);
        }
    }
    free(seq_rev); free(qual_rev);
}

/* ---------- batch driver: mapSeqMT (src/LordFAST.cpp:295-316) with per-read output slots ---------- */
typedef struct {
    const lfo_index_t *ix; const lfo_params_t *p;
    int n; const char **names, **seqs, **quals;
    char **outs; size_t *lens;
    int next; pthread_mutex_t lock;
} job_t;

static ctx_t *ctx_new(const lfo_index_t *ix, const lfo_params_t *p, int chunkSize)
{
    ctx_t *cx = (ctx_t *)calloc(1, sizeof(ctx_t));
    cx->ix = ix; cx->p = p; cx->chunkSize = chunkSize;
    cx->refWinNum = (uint32_t)ix->l_pac / (uint32_t)p->min_read_len;             /* :130 */
    /* +2: the reference writes one entry past the array when a seed falls in the final partial
     * window of a read-length-sized grid; that entry is never read back */
    cx->winCnt = (wincount_t *)malloc(((size_t)cx->refWinNum + 2) * sizeof(wincount_t));
    for (uint32_t j = 0; j < cx->refWinNum + 2; j++) { cx->winCnt[j].readIdx = 2 * chunkSize + 10; cx->winCnt[j].cnt = 0; }   /* :269-271 */
    size_t cap = (size_t)p->sampling_count * (size_t)p->max_ref_hits;
    cx->F = (lfo_seed_t *)malloc(cap * sizeof(lfo_seed_t));
    cx->R = (lfo_seed_t *)malloc(cap * sizeof(lfo_seed_t));
    cx->sel = (lfo_seed_t *)malloc(cap * sizeof(lfo_seed_t));
    cx->chain = (lfo_seed_t *)malloc(cap * sizeof(lfo_seed_t));
    cx->topWins = (win_t *)calloc((size_t)p->max_map + 1, sizeof(win_t));
    cx->mappings = (samlist_t *)calloc((size_t)p->max_map + 1, sizeof(samlist_t));
    return cx;
}

static void ctx_free(ctx_t *cx)
{
    for (int i = 0; i <= cx->p->max_map; i++) { samlist_clear(&cx->mappings[i]); free(cx->mappings[i].v); }
    free(cx->winCnt); free(cx->F); free(cx->R); free(cx->sel); free(cx->chain); free(cx->topWins); free(cx->mappings);
    free(cx);
}

static void *worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    ctx_t *cx = ctx_new(jb->ix, jb->p, jb->n);
    for (;;) {
        pthread_mutex_lock(&jb->lock);
        int t = jb->next++;
        pthread_mutex_unlock(&jb->lock);
        if (t >= jb->n) break;
        sbuf_t sb; sb_init(&sb);
        map_one(cx, t, jb->names[t], jb->seqs[t], jb->quals ? jb->quals[t] : NULL, &sb);
        jb->outs[t] = sb.s; jb->lens[t] = sb.n;
    }
    ctx_free(cx);
    return NULL;
}

char *lfo_map_batch(const lfo_index_t *ix, const lfo_params_t *p, int n, const char **names, const char **seqs,
                    const char **quals, size_t *out_len)
{
    job_t jb; memset(&jb, 0, sizeof jb);
    jb.ix = ix; jb.p = p; jb.n = n; jb.names = names; jb.seqs = seqs; jb.quals = quals;
    jb.outs = (char **)calloc((size_t)n + 1, sizeof(char *));
    jb.lens = (size_t *)calloc((size_t)n + 1, sizeof(size_t));
    pthread_mutex_init(&jb.lock, NULL);
    int nt = p->threads < 1 ? 1 : (p->threads > 255 ? 255 : p->threads);
    if (nt == 1) worker(&jb);
    else {
        pthread_t th[255];
        pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstacksize(&at, 16u << 20);
        for (int i = 0; i < nt; i++) pthread_create(&th[i], &at, worker, &jb);
        for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
    }
    size_t tot = 0;
    for (int i = 0; i < n; i++) tot += jb.lens[i];
    char *res = (char *)malloc(tot + 1), *w = res;
    for (int i = 0; i < n; i++) { memcpy(w, jb.outs[i], jb.lens[i]); w += jb.lens[i]; free(jb.outs[i]); }
    *w = 0;
    free(jb.outs); free(jb.lens);
    if (out_len) *out_len = tot;
    return res;
}

/* printSamHeader (src/BWT.cpp:668-681) */
char *lfo_sam_header(const lfo_index_t *ix, const lfo_params_t *p, const char *cmdline)
{
    sbuf_t sb; sb_init(&sb);
    sb_puts(&sb, "@HD\tVN:1.5\tSO:unsorted\n");
    for (int i = 0; i < ix->n_seqs; i++) {
        sb_puts(&sb, "@SQ\tSN:"); sb_puts(&sb, ix->contigs[i].name);
        sb_printf(&sb, "\tLN:%d\n", ix->contigs[i].len);
    }
    if (p && p->read_group_id[0] && p->read_group[0]) { sb_puts(&sb, p->read_group); sb_putc(&sb, '\n'); }   /* src/BWT.cpp:676-679 */
    sb_puts(&sb, "@PG\tID:lordfast\tPN:lordfast\tVN:0.0.10\tCL:"); sb_puts(&sb, cmdline ? cmdline : ""); sb_putc(&sb, '\n');
    return sb.s;
}
