/* lf_replay.c -- alignChain_edlib (src/LordFAST.cpp:1765-2258) as a REPLAY on the host, for the few chains that leave the common
 * path lf_walk.hip handles on the device (clip / split triggers, overlapping fragments, the misaligned-MD branch): the host walks
 * the reference's control flow, every alignment it needs is looked up in a per-chain memo, a miss registers a request (a 32-byte
 * descriptor into HBM-resident reads / reference) and the walk goes on speculatively; lf_pipeline.c sends the requests of all
 * chains to the GPU together and replays the incomplete chains.  No DP cell is computed here. */
#include "lf_pipe.h"

/* ================================================================ D: alignChain_edlib as a replay */
typedef struct {
    int ed, end; const uint8_t *ops; uint32_t nops;
    int have;
    int round; uint64_t ops_begin; uint32_t tcons;      /* where the path lives in HBM; reference bases it consumes */
    uint32_t slot, qn; int lazy;
} edres_t;

typedef struct {
    ctx_t *cx; int tid; job_t *job;
    const char *query;        /* read forward or reverse complement */
    uint32_t readLen;
    int missing;              /* edlib results still to come */
    int bail;                 /* a ksw result is missing: the walk goes on as if the test had not fired (nothing it asks for later depends on the result),
                               * so ALL extension requests of a chain leave in one round instead of one round per test */
    int build;                /* build strings (results complete so far) */
} walk_t;

static int key_eq(const rkey_t *a, const rkey_t *b) { return memcmp(a, b, sizeof(rkey_t)) == 0; }

static memo_t *memo_find(job_t *j, const rkey_t *k)
{
    if (j->hint < j->nmemo && key_eq(&j->memo[j->hint].key, k)) return &j->memo[j->hint++];
    for (int i = 0; i < j->nmemo; i++) if (key_eq(&j->memo[i].key, k)) { j->hint = i + 1; return &j->memo[i]; }
    return NULL;
}
static memo_t *memo_add(job_t *j, const rkey_t *k, arena_t *ar)
{
    if (j->nmemo == j->capmemo) { int nc = j->capmemo ? j->capmemo * 2 : 32; j->memo = (memo_t *)ar_grow(ar, j->memo, (size_t)j->capmemo * sizeof(memo_t), (size_t)nc * sizeof(memo_t)); j->capmemo = nc; }
    memo_t *m = &j->memo[j->nmemo++];
    m->key = *k; m->round = -1; m->slot = -1; m->hops = NULL;
    j->hint = j->nmemo;
    return m;
}

/* lf_map_batch_dev: the host sees a read's bases only where it has to (reads shorter than -l, the replay of the rare
 * chains, entries printed on the host): one small D2H copy on demand.  A read is touched by one worker at a time. */
void rd_host_bases(ctx_t *cx, rd_t *rd, arena_t *ar)
{
    if (rd->seq || !cx->d_seqs) return;
    char *b = (char *)ar_alloc(ar, ((size_t)rd->len + 1) * (rd->isFq ? 2 : 1));
    if (rd->len && lfg_fetch(cx->ix->device, b, cx->d_seqs + rd->src_off, rd->len) != LF_OK) memset(b, 'N', rd->len);
    b[rd->len] = 0;
    if (rd->isFq) {
        char *q = b + rd->len + 1;
        if (rd->len && lfg_fetch(cx->ix->device, q, cx->d_quals + rd->src_off, rd->len) != LF_OK) memset(q, '!', rd->len);
        q[rd->len] = 0; rd->qual = q;
    }
    rd->seq = b;
}

/* the walk's query string; the reverse complement of a read is only materialised if a byte-string request needs it */
static const char *walk_query(walk_t *w)
{
    if (!w->query) {
        rd_t *rd = &w->cx->reads[w->job->read];
        rd_host_bases(w->cx, rd, &w->cx->arena[w->tid]);
        if (!rd->seq_rev) { rd->seq_rev = (char *)ar_alloc(&w->cx->arena[w->tid], (size_t)rd->len + 1); revcomp_into(rd->seq, rd->seq_rev, rd->len); }
        w->query = rd->seq_rev;
    }
    return w->query;
}
/* bytes of a request: query segment of the walk's query string (optionally reverse-complemented) */
static void put_query(walk_t *w, const rkey_t *k, char *dst)
{
    const char *src = walk_query(w) + k->qs;
    if (!k->qrc) memcpy(dst, src, k->qn);
    else for (uint32_t i = 0; i < k->qn; i++) dst[i] = rc_char(src[k->qseg - 1 - i]);
}
static void put_target(const walk_t *w, const rkey_t *k, char *dst)
{
    const uint8_t *pac = w->cx->ix->pac;
    if (!k->trc) for (uint32_t i = 0; i < k->tn; i++) dst[i] = "ACGT"[pac_base(pac, k->ts + i)];
    else for (uint32_t i = 0; i < k->tn; i++) dst[i] = "ACGT"[3 - pac_base(pac, k->ts + k->tseg - 1 - i)];
}
static uint8_t code_of(char c)
{   /* _pf_char2int, src/LordFAST.cpp:158-164 */
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

static void stage_ksw(walk_t *w, memo_t *m)
{
    stage_t *s = &w->cx->stages[w->tid];
    const rkey_t *k = &m->key;
    if (s->kn == s->kcap) {
        s->kcap = s->kcap ? s->kcap * 2 : 64;
        s->kqoff = (uint64_t *)realloc(s->kqoff, ((size_t)s->kcap + 1) * 8); s->ktoff = (uint64_t *)realloc(s->ktoff, ((size_t)s->kcap + 1) * 8);
        s->kprm = (int32_t *)realloc(s->kprm, (size_t)s->kcap * 7 * 4); s->kowner = (memo_t **)realloc(s->kowner, (size_t)s->kcap * sizeof(memo_t *));
        if (s->kn == 0) { s->kqoff[0] = 0; s->ktoff[0] = 0; }
    }
    if (s->kqn + k->qn + 1 > s->kqcap) { s->kqcap = (s->kqn + k->qn + 1) * 2; s->kq = (uint8_t *)realloc(s->kq, s->kqcap); }
    if (s->ktn + k->tn + 1 > s->ktcap) { s->ktcap = (s->ktn + k->tn + 1) * 2; s->kt = (uint8_t *)realloc(s->kt, s->ktcap); }
    /* codes: convertChar2int / reverseComplementIntStr (src/LordFAST.cpp:1191-1201): 3 - code, so an N (4)
     * becomes 255 in the reference and indexes past its 5x5 matrix; we score any code > 3 as 0 */
    const char *src = walk_query(w) + k->qs;
    for (uint32_t i = 0; i < k->qn; i++) {
        uint8_t c = k->qrc ? code_of(src[k->qseg - 1 - i]) : code_of(src[i]);
        s->kq[s->kqn + i] = k->qrc ? (uint8_t)(c > 3 ? 4 : 3 - c) : c;
    }
    const uint8_t *pac = w->cx->ix->pac;
    for (uint32_t i = 0; i < k->tn; i++)
        s->kt[s->ktn + i] = (uint8_t)(k->trc ? 3 - pac_base(pac, k->ts + k->tseg - 1 - i) : pac_base(pac, k->ts + i));
    s->kqn += k->qn; s->ktn += k->tn;
    s->kqoff[s->kn + 1] = s->kqn; s->ktoff[s->kn + 1] = s->ktn;
    int32_t *pr = s->kprm + 7 * s->kn;
    if (k->mode == 0) { pr[0] = 0; pr[1] = 1; pr[2] = 0; pr[3] = 1; pr[4] = 40; pr[5] = 40; }        /* ksw_extend :1848,:2180 */
    else { pr[0] = 8; pr[1] = 1; pr[2] = 4; pr[3] = 1; pr[4] = 100; pr[5] = 200; }                    /* ksw_extend2 :1971,:1981 */
    pr[6] = (int32_t)k->qn;                                                                           /* h0 = readAlnLen */
    s->kowner[s->kn] = (memo_t *)(uintptr_t)(((uint64_t)(uintptr_t)(m - w->job->memo)));
    s->kn++;
}

/* request -> descriptor (no bytes are copied: the GPU reads the resident read batch and the 2-bit reference) */
static void stage_edlib_desc(walk_t *w, memo_t *m)
{
    stage_t *s = &w->cx->stages[w->tid];
    const rkey_t *k = &m->key;
    const rd_t *rd = &w->cx->reads[w->job->read];
    if (s->dn == s->dcap) {
        s->dcap = s->dcap ? s->dcap * 2 : 1024;
        s->dd = (lf_aln_desc_t *)realloc(s->dd, (size_t)s->dcap * sizeof(lf_aln_desc_t));
        s->dops = (uint64_t *)realloc(s->dops, (size_t)s->dcap * 8);
        s->downer = (uintptr_t *)realloc(s->downer, (size_t)s->dcap * sizeof(uintptr_t));
    }
    lf_aln_desc_t *d = &s->dd[s->dn];
    memset(d, 0, sizeof *d);
    const int64_t roff = (int64_t)w->cx->cat_off[rd->seed_idx], L = (int64_t)rd->len;
    /* the walk's query string is the read (forward chains) or its reverse complement (reverse chains);
     * a request may itself ask for the reverse complement of a segment: compose into (start, direction, complement).
     * complementing twice is the identity for every byte that can match the upper-case reference */
    const int rev_q = (w->job->isRev ? 1 : 0) ^ (k->qrc ? 1 : 0);
    int64_t qstart;
    if (!w->job->isRev) qstart = k->qrc ? roff + k->qs + k->qseg - 1 : roff + k->qs;
    else qstart = k->qrc ? roff + L - k->qs - k->qseg : roff + L - 1 - k->qs;
    d->qstart = qstart;
    d->tstart = k->trc ? (int64_t)k->ts + k->tseg - 1 : (int64_t)k->ts;
    d->n = k->qn; d->m = k->tn; d->mode = k->mode;
    d->flags = (uint8_t)((rev_q ? (LF_F_QREV | LF_F_QCOMP) : 0) | (k->trc ? (LF_F_TREV | LF_F_TCOMP) : 0) | (w->cx->lazy ? LF_F_LAZYX : 0));
    s->dops[s->dn] = s->dops_total; s->dops_total += (uint64_t)k->qn + k->tn;
    s->downer[s->dn] = (uintptr_t)(m - w->job->memo);
    s->dn++;
    s->ext_bytes += (uint64_t)k->qn + (k->tn + 3) / 4 + k->qn + k->tn;      /* SURVEY 8(d) B_ext */
    s->blk_steps += (uint64_t)((k->qn + 63) / 64) * k->tn;
}

/* edlibAlign(query segment, target segment, mode, PATH) through the memo */
static edres_t need_edlib(walk_t *w, int qrc, uint32_t qs, uint32_t qseg, uint32_t qn, int trc, uint32_t ts, uint32_t tseg, uint32_t tn, int mode)
{
    edres_t r; memset(&r, 0, sizeof r);
    rkey_t k; memset(&k, 0, sizeof k);
    k.type = 0; k.qrc = (uint8_t)qrc; k.trc = (uint8_t)trc; k.mode = (uint8_t)mode;
    k.qs = qs; k.qseg = qseg; k.qn = qn; k.ts = ts; k.tseg = tseg; k.tn = tn;
    memo_t *m = memo_find(w->job, &k);
    if (!m) {
        m = memo_add(w->job, &k, &w->cx->arena[w->tid]);
        /* every request is a descriptor: problems above edlib's traceback switch are cut into leaves by the Hirschberg levels
         * on the device (lf_hirsch.hip, any query length), an empty side is a pure run written by the binning kernel */
        stage_edlib_desc(w, m); jv_push(&w->cx->edd_jobs[w->tid], w->job);
    }
    if (m->round < 0) { w->missing++; w->build = 0; r.ed = 0; r.end = (mode == 0) ? (int)tn - 1 : (int)tn - 1; return r; }
    const ed_round_t *R = &w->cx->ed_rounds[m->round];
    r.have = 1; r.ed = R->ed[m->slot]; r.end = R->end[m->slot]; r.nops = R->ops_len[m->slot];
    r.round = m->round; r.ops_begin = R->ops_off[m->slot] + ((uint64_t)qn + tn - r.nops);     /* end-aligned in its region */
    r.tcons = mode == 0 ? tn : (uint32_t)(r.end + 1);                               /* NW: all of it; SHW: up to the end column */
    r.slot = (uint32_t)m->slot; r.qn = qn; r.lazy = R->lazy;
    r.ops = m->hops ? m->hops + ((uint64_t)qn + tn - r.nops) : (R->ops ? R->ops + r.ops_begin : NULL);
    return r;
}

static int need_ksw(walk_t *w, int set, int qrc, uint32_t qs, uint32_t qseg, int trc, uint32_t ts, uint32_t tseg, int *qle, int *tle)
{
    rkey_t k; memset(&k, 0, sizeof k);
    k.type = 1; k.qrc = (uint8_t)qrc; k.trc = (uint8_t)trc; k.mode = (uint8_t)set;
    k.qs = qs; k.qseg = qseg; k.qn = qseg; k.ts = ts; k.tseg = tseg; k.tn = tseg;
    memo_t *m = memo_find(w->job, &k);
    if (!m) {
        /* LF_KSW_ONE_PER_ROUND=1 (A / B, read per request: a miss is rare): round 4's behaviour, a walk asks for ONE extension and stops asking */
        if (w->bail && lf_env_long("LF_KSW_ONE_PER_ROUND", 0) != 0) return 0;
        m = memo_add(w->job, &k, &w->cx->arena[w->tid]); stage_ksw(w, m); jv_push(&w->cx->ksw_jobs[w->tid], w->job);
    }
    if (m->round < 0) { w->bail = 1; w->build = 0; return 0; }
    const ksw_round_t *R = &w->cx->ksw_rounds[m->round];
    *qle = R->qle[m->slot]; *tle = R->tle[m->slot];
    return 1;
}

/* ---- CIGAR / MD tracks (src/LordFAST.cpp:1570-1763) ---- */
static const char OP2CH[4] = { 'M', 'I', 'D', 'M' };
static void ops_back(track_t *cg, track_t *md, const edres_t *r, const uint8_t *pac, int trc, uint32_t ts, uint32_t tseg)
{   /* edlibCigar_pushback + edlibMD_pushback: target base for deletions / mismatches */
    if (!r->have) return;
    tr_room(cg, 0, r->nops); tr_room(md, 0, r->nops);
    uint32_t ti = 0;
    for (uint32_t i = 0; i < r->nops; i++) {
        const uint8_t op = r->ops[i];
        cg->buf[cg->end++] = OP2CH[op];
        char m;
        if (op == 0) { m = '='; ti++; }
        else if (op == 1) m = '-';
        else { m = "ACGT"[trc ? 3 - pac_base(pac, ts + tseg - 1 - ti) : pac_base(pac, ts + ti)]; ti++; }
        md->buf[md->end++] = m;
    }
}
static void ops_front(track_t *cg, track_t *md, const edres_t *r, const uint8_t *pac, uint32_t ts, uint32_t tseg)
{   /* edlibCigar_pushfront + edlibMD_pushfront: the alignment was computed on reverse complements of
       query prefix and reference [ts, ts+tseg); pushing each op to the front restores forward order and
       the MD base is the complement of the (reverse-complemented) target base = the forward base */
    if (!r->have) return;
    tr_room(cg, r->nops, 0); tr_room(md, r->nops, 0);
    uint32_t ti = 0;
    for (uint32_t i = 0; i < r->nops; i++) {
        const uint8_t op = r->ops[i];
        cg->buf[--cg->beg] = OP2CH[op];
        char m;
        if (op == 0) { m = '='; ti++; }
        else if (op == 1) m = '-';
        else { m = "ACGT"[pac_base(pac, ts + tseg - 1 - ti)]; ti++; }   /* complement(rc target[ti]) */
        md->buf[--md->beg] = m;
    }
}

static char *cigar_string(const track_t *c, arena_t *ar)
{   /* edlibCigar_toString: leading / trailing I runs print as S */
    str_t sb; str_init_ar(&sb, ar, 512);
    char ch = 0; unsigned num = 0; int opn = 0;
    const size_t n = tr_size(c);
    for (size_t i = 0; i < n; i++) {
        const char x = c->buf[c->beg + i];
        if (x != ch) {
            if (ch != 0) { str_putu(&sb, num); str_putc(&sb, (opn == 0 && ch == 'I') ? 'S' : ch); opn++; }
            num = 1; ch = x;
        } else num++;
    }
    if (num) { str_putu(&sb, num); str_putc(&sb, ch == 'I' ? 'S' : ch); }
    return sb.s;
}
static char *md_string(const track_t *md, const track_t *cg, arena_t *ar)
{   /* edlibMD_toString */
    str_t sb; str_init_ar(&sb, ar, 512);
    unsigned num = 0; char last = '=';
    const size_t n = tr_size(md);
    for (size_t i = 0; i < n; i++) {
        const char m = md->buf[md->beg + i], c = cg->buf[cg->beg + i];
        if (m == '=') { num++; last = '='; }
        else if (m == '-') last = 'I';
        else if (c == 'M') { str_putu(&sb, num); num = 0; str_putc(&sb, m); last = 'X'; }
        else if (c == 'D') { if (last != 'D') { str_putu(&sb, num); num = 0; str_putc(&sb, '^'); } str_putc(&sb, m); last = 'D'; }
    }
    str_putu(&sb, num);
    return sb.s;
}

/* ---- alignment builder: CIGAR + MD of one SAM record ------------------------------------------------
 * STREAM mode (default) run-length encodes on the fly: it is fed (cigar char, md char) pairs in final order and
 * keeps the state machines of edlibCigar_toString / edlibMD_toString (src/LordFAST.cpp:1596-1626,1717-1763), so no
 * per-base track is materialised.  Front insertions are legal only while nothing has been appended (that is how the
 * reference uses them: left extension first, or right after a clear); they are stacked and flushed first.
 * TRACK mode keeps the per-base deques of the reference and is used when a walk hits the one branch that misaligns
 * MD and CIGAR (src/LordFAST.cpp:2057, App. B #3). */
typedef struct { int kind; char cg, md; uint32_t n; edres_t r; uint32_t ts, tseg; } fseg_t;
typedef struct {
    int track_mode, need_track, active;    /* track_mode: 0 STREAM (host strings), 1 TRACK (per-base), 2 RECIPE (GPU renders) */
    stage_t *rs; uint64_t item_mark;       /* RECIPE mode: the worker's item list; first item of the open record */
    track_t cg, md;                        /* TRACK mode */
    str_t scg, smd; char ch; unsigned run; int opn; unsigned mdnum; char last; int fed;    /* STREAM mode */
    fseg_t front[8]; int nfront;
    const uint8_t *pac; arena_t *ar; size_t hint;
} alnb_t;

static void ab_reset_stream(alnb_t *b) { str_init_ar(&b->scg, b->ar, b->hint); str_init_ar(&b->smd, b->ar, b->hint); b->ch = 0; b->run = 0; b->opn = 0; b->mdnum = 0; b->last = '='; b->fed = 0; b->nfront = 0; }
static void ab_init(alnb_t *b, int track_mode, int active, const uint8_t *pac, size_t hint, arena_t *ar, stage_t *rs)
{
    memset(b, 0, sizeof *b);
    b->track_mode = track_mode; b->active = active; b->pac = pac; b->ar = ar; b->hint = hint / 2 + 128;
    if (!active) return;
    if (track_mode == 2) { b->rs = rs; b->item_mark = rs->rin; b->last = '='; }
    else if (track_mode) { tr_init(&b->cg, hint); tr_init(&b->md, hint); } else ab_reset_stream(b);
}
static void ab_free(alnb_t *b)
{
    if (!b->active) return;
    if (b->track_mode == 1) { free(b->cg.buf); free(b->md.buf); }
}
/* ---- RECIPE mode: the order of the pieces is all the host records ---- */
static inline lf_ritem_t *rc_item(alnb_t *b)
{
    stage_t *s = b->rs;
    if (s->rin == s->ricap) { s->ricap = s->ricap ? s->ricap * 2 : 4096; s->ri = (lf_ritem_t *)realloc(s->ri, s->ricap * sizeof(lf_ritem_t)); }
    lf_ritem_t *it = &s->ri[s->rin++];
    memset(it, 0, sizeof *it);
    return it;
}
static void rc_run(alnb_t *b, int kind, uint32_t n, uint32_t tpos)
{
    if (!n) return;
    b->fed = 1;
    lf_ritem_t *it = rc_item(b); it->kind = (uint8_t)kind; it->n = n; it->tpos = tpos;
}
static void rc_ops(alnb_t *b, const edres_t *r, int kind, uint32_t tpos)
{
    if (r->round >= LF_MAX_ED_ROUNDS) { b->need_track = 1; return; }      /* beyond the rounds kept in HBM: host fallback */
    if (!r->nops) return;
    b->fed = 1;
    lf_ritem_t *it = rc_item(b); it->kind = (uint8_t)kind; it->n = r->nops; it->tpos = tpos; it->round = (uint8_t)r->round; it->ops_begin = r->ops_begin;
    it->slot = r->slot; it->qn = r->qn; it->tcons = r->tcons; it->lazy = (uint8_t)r->lazy;
}
static inline void st_c(alnb_t *b, char c, uint32_t n)
{
    if (c != b->ch) {
        if (b->ch) { str_putu(&b->scg, b->run); str_putc(&b->scg, (b->opn == 0 && b->ch == 'I') ? 'S' : b->ch); b->opn++; }
        b->run = n; b->ch = c;
    } else b->run += n;
}
static inline void st_md_base(alnb_t *b, char base, int is_del)
{
    if (!is_del) { str_putu(&b->smd, b->mdnum); b->mdnum = 0; str_putc(&b->smd, base); b->last = 'X'; }
    else { if (b->last != 'D') { str_putu(&b->smd, b->mdnum); b->mdnum = 0; str_putc(&b->smd, '^'); } str_putc(&b->smd, base); b->last = 'D'; }
}
static void st_run(alnb_t *b, char cg, char md, uint32_t n)
{
    if (!n) return;
    b->fed = 1;
    st_c(b, cg, n);
    if (md == '=') { b->mdnum += n; b->last = '='; } else b->last = 'I';          /* md is '=' or '-' for runs */
}
/* ops in forward order (target base index grows) */
static void st_ops_fwd(alnb_t *b, const edres_t *r, int trc, uint32_t ts, uint32_t tseg)
{
    const uint8_t *pac = b->pac; uint32_t ti = 0;
    if (r->nops) b->fed = 1;
    for (uint32_t i = 0; i < r->nops; ) {
        const uint8_t op = r->ops[i];
        if (op == 0) { uint32_t j = i + 1; while (j < r->nops && r->ops[j] == 0) j++; st_c(b, 'M', j - i); b->mdnum += j - i; b->last = '='; ti += j - i; i = j; continue; }
        if (op == 1) { st_c(b, 'I', 1); b->last = 'I'; i++; continue; }
        const char base = "ACGT"[trc ? 3 - pac_base(pac, ts + tseg - 1 - ti) : pac_base(pac, ts + ti)];
        st_c(b, op == 2 ? 'D' : 'M', 1); st_md_base(b, base, op == 2); ti++; i++;
    }
}
/* ops that the reference pushes to the FRONT one by one (reversed order); target = reverse complement of
 * [ts, ts+tseg): the MD base is the complement of it, i.e. the forward base (edlibMD_pushfront) */
static void st_ops_rev(alnb_t *b, const edres_t *r, uint32_t ts, uint32_t tseg)
{
    const uint8_t *pac = b->pac;
    uint32_t ti = 0;
    for (uint32_t i = 0; i < r->nops; i++) ti += (r->ops[i] != 1);
    if (r->nops) b->fed = 1;
    for (uint32_t i = r->nops; i-- > 0; ) {
        const uint8_t op = r->ops[i];
        if (op == 0) { ti--; st_c(b, 'M', 1); b->mdnum++; b->last = '='; continue; }
        if (op == 1) { st_c(b, 'I', 1); b->last = 'I'; continue; }
        ti--;
        const char base = "ACGT"[pac_base(pac, ts + tseg - 1 - ti)];
        st_c(b, op == 2 ? 'D' : 'M', 1); st_md_base(b, base, op == 2);
    }
}
static void ab_flush_front(alnb_t *b)
{
    while (b->nfront > 0) {
        const fseg_t *f = &b->front[--b->nfront];
        if (b->track_mode == 2) {
            if (f->kind == 0) rc_run(b, f->cg == 'M' ? LF_RI_RUN_M : LF_RI_RUN_I, f->n, 0);
            else rc_ops(b, &f->r, LF_RI_OPS_REV, f->ts + f->tseg - f->r.tcons);
            continue;
        }
        if (f->kind == 0) st_run(b, f->cg, f->md, f->n); else st_ops_rev(b, &f->r, f->ts, f->tseg);
    }
}
static void ab_back_run(alnb_t *b, char cg, char md, size_t n)
{
    if (!b->active) return;
    if (b->track_mode == 1) { tr_back_n(&b->cg, n, cg); tr_back_n(&b->md, n, md); return; }
    ab_flush_front(b);
    if (b->track_mode == 2) { rc_run(b, cg == 'M' ? LF_RI_RUN_M : LF_RI_RUN_I, (uint32_t)n, 0); return; }
    st_run(b, cg, md, (uint32_t)n);
}
static void ab_front_run(alnb_t *b, char cg, char md, size_t n)
{
    if (!b->active) return;
    if (b->track_mode == 1) { tr_front_n(&b->cg, n, cg); tr_front_n(&b->md, n, md); return; }
    if (b->fed || b->nfront >= 8) { b->need_track = 1; return; }
    fseg_t *f = &b->front[b->nfront++]; f->kind = 0; f->cg = cg; f->md = md; f->n = (uint32_t)n;
}
static void ab_back_ops(alnb_t *b, const edres_t *r, int trc, uint32_t ts, uint32_t tseg)
{
    if (!b->active || !r->have) return;
    if (b->track_mode == 1) { ops_back(&b->cg, &b->md, r, b->pac, trc, ts, tseg); return; }
    ab_flush_front(b);
    if (b->track_mode == 2) { rc_ops(b, r, trc ? LF_RI_OPS_FWD_TRC : LF_RI_OPS_FWD, trc ? ts + tseg - 1 : ts); return; }
    st_ops_fwd(b, r, trc, ts, tseg);
}
static void ab_front_ops(alnb_t *b, const edres_t *r, uint32_t ts, uint32_t tseg)
{
    if (!b->active || !r->have) return;
    if (b->track_mode == 1) { ops_front(&b->cg, &b->md, r, b->pac, ts, tseg); return; }
    if (b->fed || b->nfront >= 8) { b->need_track = 1; return; }
    fseg_t *f = &b->front[b->nfront++]; f->kind = 1; f->r = *r; f->ts = ts; f->tseg = tseg;
}
static void ab_back_del(alnb_t *b, uint32_t ts, uint32_t n)
{   /* pure deletion between two anchors (src/LordFAST.cpp:2126-2134) */
    if (!b->active || !n) return;
    if (b->track_mode == 1) {
        tr_back_n(&b->cg, n, 'D'); tr_room(&b->md, 0, n);
        for (uint32_t j = 0; j < n; j++) b->md.buf[b->md.end++] = "ACGT"[pac_base(b->pac, ts + j)];
        return;
    }
    ab_flush_front(b); b->fed = 1;
    if (b->track_mode == 2) { rc_run(b, LF_RI_DEL, n, ts); return; }
    st_c(b, 'D', n);
    for (uint32_t j = 0; j < n; j++) st_md_base(b, "ACGT"[pac_base(b->pac, ts + j)], 1);
}
static void ab_md_front_only(alnb_t *b, size_t n)
{   /* the reference's misplaced padding: MD at the front while the CIGAR got it at the back */
    if (!b->active) return;
    if (b->track_mode == 1) { tr_front_n(&b->md, n, '-'); return; }
    b->need_track = 1;
}
static void ab_cg_back_only(alnb_t *b, size_t n)
{
    if (!b->active) return;
    if (b->track_mode == 1) { tr_back_n(&b->cg, n, 'I'); return; }
    b->need_track = 1;
}
static void ab_clear(alnb_t *b)
{
    if (!b->active) return;
    if (b->track_mode == 1) { tr_clear(&b->cg); tr_clear(&b->md); return; }
    if (b->track_mode == 2) { b->rs->rin = b->item_mark; b->fed = 0; b->nfront = 0; return; }
    ab_reset_stream(b);
}
/* strings of the record built so far (ownership passes to the caller) */
static void ab_take(alnb_t *b, char **cigar, char **md)
{
    if (b->track_mode == 1) { *cigar = cigar_string(&b->cg, b->ar); *md = md_string(&b->md, &b->cg, b->ar); return; }
    ab_flush_front(b);
    if (b->run) { str_putu(&b->scg, b->run); str_putc(&b->scg, b->ch == 'I' ? 'S' : b->ch); }
    str_putu(&b->smd, b->mdnum);
    *cigar = b->scg.s; *md = b->smd.s;
    ab_reset_stream(b);
}

static void emit_sam(walk_t *w, samlist_t *map, const sam_t *tmp, alnb_t *ab)
{
    if (!w->build || !ab->active || ab->need_track) return;
    if (ab->track_mode == 2) {                       /* close the record: its pieces are items [item_mark, rin) */
        ab_flush_front(ab);
        if (ab->need_track) return;
        stage_t *s = ab->rs;
        if (s->rrn == s->rrcap) { s->rrcap = s->rrcap ? s->rrcap * 2 : 1024; s->rr = (lf_rrecord_t *)realloc(s->rr, (size_t)s->rrcap * sizeof(lf_rrecord_t)); }
        s->rr[s->rrn].item0 = (uint32_t)ab->item_mark; s->rr[s->rrn].nitems = (uint32_t)(s->rin - ab->item_mark);
        samlist_push(map, tmp, NULL, NULL, ab->ar);
        map->v[map->n - 1].rec = s->rrn++; map->v[map->n - 1].rtid = w->tid;
        ab->item_mark = s->rin; ab->fed = 0; ab->nfront = 0;
        return;
    }
    char *c, *m;
    ab_take(ab, &c, &m);
    samlist_push(map, tmp, c, m, ab->ar);
}

/* the walk itself.  Returns 1 when every alignment it needed was available (map is then final). */
static int walk_chain_mode(ctx_t *cx, int tid, job_t *job, samlist_t *map, int track_mode, int active, int *need_track)
{
    const struct lf_index *ix = cx->ix;
    const uint8_t *pac = ix->pac;
    rd_t *rd = &cx->reads[job->read];
    const int isRev = job->isRev;
    const Seed_t *s = job->chain;
    const uint32_t chainLen = job->chainLen;
    walk_t W; memset(&W, 0, sizeof W);
    rd_host_bases(cx, rd, &cx->arena[tid]);
    W.cx = cx; W.tid = tid; W.job = job; W.query = isRev ? rd->seq_rev : rd->seq; W.readLen = rd->len; W.build = 1;
    job->hint = 0;
    const int32_t readLen = (int32_t)rd->len;
    stage_t *const rs = &cx->stages[tid];
    const uint64_t rin0 = rs->rin; const int rrn0 = rs->rrn;
    alnb_t ab; ab_init(&ab, track_mode, active, pac, rd->len, &cx->arena[tid], rs);
    sam_t tmp; memset(&tmp, 0, sizeof tmp);
    uint32_t chrBeg, chrEnd, readAlnStart, refAlnStart, readAlnEnd, refAlnEnd, i;
    int32_t readAlnLen, refAlnLen, editScore = 0;
    int qle = 0, tle = 0;
    samlist_clear(map);

    chr_boundaries(ix, s[0].tPos, s[chainLen - 1].tPos, &chrBeg, &chrEnd);                   /* :1799 */
    tmp.flag = isRev ? 16 : 0; tmp.pos = s[0].tPos; tmp.qStart = s[0].qPos;

    /* ---- before the first anchor (:1820-1899) ---- */
    readAlnLen = (int32_t)s[0].qPos; refAlnLen = readAlnLen + 20;
    if (readAlnLen > 0) {
        if ((int64_t)s[0].tPos - refAlnLen >= (int64_t)chrBeg) {
            refAlnStart = s[0].tPos - (uint32_t)refAlnLen;
            edres_t r = need_edlib(&W, 1, 0, (uint32_t)readAlnLen, (uint32_t)readAlnLen, 1, refAlnStart, (uint32_t)refAlnLen, (uint32_t)refAlnLen, 1);
            int realigned = 0;
            if (r.have && readAlnLen > CLIP_LEN && (1 - ((float)r.ed / readAlnLen)) < CLIP_SIM) {
                if (need_ksw(&W, 0, 1, 0, (uint32_t)readAlnLen, 1, refAlnStart, (uint32_t)refAlnLen, &qle, &tle) && qle > 0 && qle < readAlnLen) {
                    edres_t r2 = need_edlib(&W, 1, 0, (uint32_t)readAlnLen, (uint32_t)qle, 1, refAlnStart, (uint32_t)refAlnLen, (uint32_t)tle, 0);
                    ab_front_ops(&ab, &r2, refAlnStart, (uint32_t)refAlnLen);
                    editScore -= r2.ed;
                    tmp.pos = s[0].tPos - (uint32_t)r2.end - 1;
                    tmp.qStart = s[0].qPos - (uint32_t)qle;
                    ab_front_run(&ab, 'I', '-', (size_t)(readAlnLen - qle));
                    realigned = 1;
                }
            }
            if (!realigned) {
                editScore -= r.ed;
                ab_front_ops(&ab, &r, refAlnStart, (uint32_t)refAlnLen);
                tmp.pos = s[0].tPos - (uint32_t)r.end - 1;
                tmp.qStart = 0;
            }
        } else ab_front_run(&ab, 'I', '-', (size_t)readAlnLen);
    }

    /* ---- between adjacent anchors (:1901-2137) ---- */
    int numAnchorsSoFar = 1;
    for (i = 0; i + 1 < chainLen; i++) {
        ab_back_run(&ab, 'M', '=', s[i].len);
        readAlnStart = s[i].qPos + s[i].len; refAlnStart = s[i].tPos + s[i].len;
        readAlnEnd = s[i + 1].qPos; refAlnEnd = s[i + 1].tPos;
        readAlnLen = (int32_t)(readAlnEnd - readAlnStart); refAlnLen = (int32_t)(refAlnEnd - refAlnStart);
        if (readAlnLen > 0 && refAlnLen > 0) {
            edres_t r = need_edlib(&W, 0, readAlnStart, (uint32_t)readAlnLen, (uint32_t)readAlnLen, 0, refAlnStart, (uint32_t)refAlnLen, (uint32_t)refAlnLen, 0);
            int handled = 0;
            if (r.have && abs(readAlnLen - refAlnLen) >= SPLIT_LEN && (1 - ((float)r.ed / readAlnLen)) < SPLIT_SIM) {
                /* split test: extension from both ends of the gap (:1967-1983) */
                int q1 = 0, t1 = 0, q2 = 0, t2 = 0;
                const int k1 = need_ksw(&W, 1, 0, readAlnStart, (uint32_t)readAlnLen, 0, refAlnStart, (uint32_t)refAlnLen, &q1, &t1);
                const int k2 = need_ksw(&W, 1, 1, readAlnStart, (uint32_t)readAlnLen, 1, refAlnStart, (uint32_t)refAlnLen, &q2, &t2);
                const uint32_t rs_new = readAlnStart + (uint32_t)q1, ts_new = refAlnStart + (uint32_t)t1;
                const uint32_t re_new = readAlnEnd - (uint32_t)q2, te_new = refAlnEnd - (uint32_t)t2;
                const int32_t tl_new = (int32_t)(te_new - ts_new), rl_new = (int32_t)(re_new - rs_new);
                if (k1 && k2 && (rs_new < re_new || ts_new < te_new)) {                      /* :1995 */
                    handled = 1;
                    if (rs_new > readAlnStart || ts_new > refAlnStart) {                     /* first part :1998-2007 */
                        edres_t a = need_edlib(&W, 0, readAlnStart, (uint32_t)readAlnLen, rs_new - readAlnStart, 0, refAlnStart, (uint32_t)refAlnLen, ts_new - refAlnStart, 0);
                        ab_back_ops(&ab, &a, 0, refAlnStart, (uint32_t)refAlnLen);
                        editScore -= a.ed;
                    }
                    ab_back_run(&ab, 'I', '-', (size_t)((uint32_t)readLen - rs_new));
                    tmp.posEnd = ts_new; tmp.qEnd = rs_new; tmp.nmCount = editScore;
                    if (numAnchorsSoFar > 1) emit_sam(&W, map, &tmp, &ab);
                    ab_clear(&ab); editScore = 0;
                    if (rs_new < re_new && ts_new < te_new) {                                /* middle part :2033-2077 */
                        edres_t f = need_edlib(&W, 0, rs_new, (uint32_t)rl_new, (uint32_t)rl_new, 0, ts_new, (uint32_t)tl_new, (uint32_t)tl_new, 0);
                        edres_t v = need_edlib(&W, 1, rs_new, (uint32_t)rl_new, (uint32_t)rl_new, 0, ts_new, (uint32_t)tl_new, (uint32_t)tl_new, 0);
                        if (f.have && v.have && (1 - ((double)v.ed / rl_new)) > (1 - ((double)f.ed / rl_new)) && (1 - ((double)v.ed / rl_new)) > REVERSE_SIM) {
                            tmp.flag = isRev ? 0 : 16;
                            tmp.pos = ts_new; tmp.qStart = rs_new; tmp.posEnd = te_new; tmp.qEnd = re_new;
                            ab_back_run(&ab, 'I', '-', rs_new);
                            ab_back_ops(&ab, &v, 0, ts_new, (uint32_t)tl_new);
                            editScore -= v.ed;
                            ab_cg_back_only(&ab, (size_t)((uint32_t)readLen - re_new));
                            ab_md_front_only(&ab, (size_t)((uint32_t)readLen - re_new));          /* sic :2057 (App. B #3) */
                            tmp.nmCount = editScore;
                            emit_sam(&W, map, &tmp, &ab);
                            ab_clear(&ab); editScore = 0;
                        }
                    }
                    if (re_new < readAlnEnd || te_new < refAlnEnd) {                          /* second part :2079-2090 */
                        edres_t b = need_edlib(&W, 1, readAlnStart, (uint32_t)readAlnLen, readAlnEnd - re_new, 1, refAlnStart, (uint32_t)refAlnLen, refAlnEnd - te_new, 0);
                        ab_front_ops(&ab, &b, refAlnStart, (uint32_t)refAlnLen);
                        editScore -= b.ed;
                    }
                    ab_front_run(&ab, 'I', '-', re_new);
                    tmp.flag = isRev ? 16 : 0; tmp.pos = te_new; tmp.qStart = re_new;
                    numAnchorsSoFar = 0;
                }
            }
            if (!handled) { editScore -= r.ed; ab_back_ops(&ab, &r, 0, refAlnStart, (uint32_t)refAlnLen); }
        } else if (readAlnLen > 0) {
            ab_back_run(&ab, 'I', '-', (size_t)readAlnLen);
            editScore -= readAlnLen;
        } else {
            if (refAlnLen > 0) ab_back_del(&ab, refAlnStart, (uint32_t)refAlnLen);
            editScore -= refAlnLen;
        }
        numAnchorsSoFar++;
    }

    /* ---- last anchor and the tail (:2149-2230) ---- */
    ab_back_run(&ab, 'M', '=', s[i].len);
    tmp.posEnd = s[i].tPos + s[i].len - 1; tmp.qEnd = s[i].qPos + s[i].len - 1;
    readAlnStart = s[i].qPos + s[i].len;
    readAlnLen = readLen - (int32_t)readAlnStart; refAlnLen = readAlnLen + 20;
    if (readAlnLen > 0) {
        if (s[i].tPos + s[i].len + (uint32_t)refAlnLen - 1 <= chrEnd) {
            refAlnStart = s[i].tPos + s[i].len;
            edres_t r = need_edlib(&W, 0, readAlnStart, (uint32_t)readAlnLen, (uint32_t)readAlnLen, 0, refAlnStart, (uint32_t)refAlnLen, (uint32_t)refAlnLen, 1);
            int realigned = 0;
            if (r.have && readAlnLen > CLIP_LEN && (1 - ((float)r.ed / readAlnLen)) < CLIP_SIM) {
                if (need_ksw(&W, 0, 0, readAlnStart, (uint32_t)readAlnLen, 0, refAlnStart, (uint32_t)refAlnLen, &qle, &tle) && qle > 0 && qle < readAlnLen) {
                    edres_t r2 = need_edlib(&W, 0, readAlnStart, (uint32_t)readAlnLen, (uint32_t)qle, 0, refAlnStart, (uint32_t)refAlnLen, (uint32_t)tle, 0);
                    ab_back_ops(&ab, &r2, 0, refAlnStart, (uint32_t)refAlnLen);
                    editScore -= r2.ed;
                    tmp.posEnd = refAlnStart + (uint32_t)r2.end;
                    tmp.qEnd = readAlnStart + (uint32_t)qle;
                    ab_back_run(&ab, 'I', '-', (size_t)(readAlnLen - qle));
                    realigned = 1;
                }
            }
            if (!realigned) {
                editScore -= r.ed;
                ab_back_ops(&ab, &r, 0, refAlnStart, (uint32_t)refAlnLen);
                tmp.posEnd = refAlnStart + (uint32_t)r.end;
                tmp.qEnd = (uint32_t)readLen;
            }
        } else { ab_back_run(&ab, 'I', '-', (size_t)readAlnLen); }
    }
    tmp.nmCount = editScore;
    emit_sam(&W, map, &tmp, &ab);
    *need_track = ab.need_track;
    ab_free(&ab);
    job->complete = (W.missing == 0 && !W.bail);
    if (!job->complete || ab.need_track) { samlist_clear(map); rs->rin = rin0; rs->rrn = rrn0; }
    else if (track_mode == 2 && active) rs->rin = ab.item_mark;          /* pieces after the last record are dropped */
    return job->complete;
}

/* per-base fallback only: a lazy path (op 0 on every diagonal move) copied back from HBM gets its mismatches here,
 * by the comparison the edlib kernels make: raw bytes of the request's query and target strings */
static void resolve_lazy_ops(ctx_t *cx, int tid, job_t *job, memo_t *m, uint32_t nops)
{
    const rkey_t *k = &m->key;
    rd_t *rd = &cx->reads[job->read];
    walk_t W; memset(&W, 0, sizeof W);
    rd_host_bases(cx, rd, &cx->arena[tid]);
    W.cx = cx; W.tid = tid; W.job = job; W.query = job->isRev ? rd->seq_rev : rd->seq; W.readLen = rd->len;
    char *q = (char *)ar_alloc(&cx->arena[tid], (size_t)k->qn + k->tn + 2), *t = q + k->qn + 1;
    put_query(&W, k, q); put_target(&W, k, t);
    uint8_t *ops = m->hops + ((size_t)k->qn + k->tn - nops);
    uint32_t qi = 0, ti = 0;
    for (uint32_t i = 0; i < nops; i++) {
        const uint8_t op = ops[i];
        if (op == 1) { qi++; continue; }
        if (op == 2) { ti++; continue; }
        if (op == 0 && q[qi] != t[ti]) ops[i] = 3;
        qi++; ti++;
    }
}

int walk_chain(ctx_t *cx, int tid, job_t *job, samlist_t *map)
{
    int need_track = 0;
    const int active = job->nmemo > 0;        /* a first walk has no results yet: it only registers requests */
    const int mode = cx->host_cigar ? 0 : 2;
    int done = walk_chain_mode(cx, tid, job, map, mode, active, &need_track);
    if (done && !active) done = walk_chain_mode(cx, tid, job, map, mode, 1, &need_track);   /* chain without any alignment */
    if (done && need_track) {                                                               /* rare: per-base tracks on the host */
        for (int k = 0; k < job->nmemo; k++) {          /* bring this job's edit paths back from HBM */
            memo_t *m = &job->memo[k];
            if (m->key.type != 0 || m->round < 0 || m->hops) continue;
            const ed_round_t *R = &cx->ed_rounds[m->round];
            if (!R->lazy && (R->ops || !R->d_ops)) continue;
            const size_t region = (size_t)m->key.qn + m->key.tn;
            m->hops = (uint8_t *)ar_alloc(&cx->arena[tid], region + 1);
            if (R->ops) memcpy(m->hops, R->ops + R->ops_off[m->slot], region);
            else if (lfg_fetch(cx->ix->device, m->hops, R->d_ops + R->ops_off[m->slot], region) != LF_OK) { m->hops = NULL; return 0; }
            if (R->lazy) resolve_lazy_ops(cx, tid, job, m, R->ops_len[m->slot]);
        }
        done = walk_chain_mode(cx, tid, job, map, 1, 1, &need_track);
    }
    return done;
}


/* alignWin's scoring tail (src/LordFAST.cpp:1063-1090,1148-1175) */
void score_mapping(const lf_params_t *p, samlist_t *map, int isReverse, uint32_t rLen, uint32_t chainLen)
{
    if (chainLen > 1) {
        map->totalScore = 0;
        for (int i = 0; i < map->n; i++) {
            map->v[i].alnScore = (int32_t)((uint32_t)map->v[i].nmCount + (map->v[i].qEnd - map->v[i].qStart));
            map->totalScore += map->v[i].nmCount;
        }
        const double gp = isReverse ? p->gap_penalty : 0.15;                                  /* :1077 vs :1162 */
        for (int i = 0; i + 1 < map->n; i++) {
            int64_t a = (int64_t)map->v[i + 1].pos - (int64_t)map->v[i].posEnd, b = (int64_t)map->v[i + 1].qStart - (int64_t)map->v[i].qEnd;
            uint32_t diff = (uint32_t)((a < 0 ? -a : a) + (b < 0 ? -b : b));
            map->totalScore = (int32_t)((double)map->totalScore - gp * (double)diff);
        }
        map->totalScore = (int32_t)((uint32_t)map->totalScore - map->v[0].qStart);
        map->totalScore = (int32_t)((uint32_t)map->totalScore - (rLen - map->v[map->n - 1].qEnd));
    } else map->totalScore = (int32_t)((uint32_t)-2 * rLen);
}

