/*
 * lf_pipeline.c -- lf_map_batch: mapSeq (src/LordFAST.cpp:461-580) for a whole read batch.
 *
 * The reference runs one read at a time through seed -> vote -> chain -> extend.  Here every stage is a
 * batch over all reads of a chunk so that each HIP kernel sees 10^5..10^7 independent work items:
 *
 *   A  GPU   seeds of every read                                   lf_seed.hip
 *   B  GPU   window vote (sparse, same counts/ties as the dense tagged array), local maxima, top-N heap,
 *            coarse/fine decision, seed selection + std::sort order per candidate window   lf_vote.hip
 *   C  GPU   dp-n2 chains of all candidate windows                 lf_chain_kernel.h
 *      host  fine mode: top-N heap replay on the chain scores
 *   D  GPU   alignments.  alignChain_edlib (src/LordFAST.cpp:1765-2258) is data dependent (clip test,
 *            split test), so it is written as a REPLAY: the host walks the reference's control flow,
 *            every alignment it needs is looked up in a memo; a miss registers a request and the walk
 *            continues speculatively on the common path.  Requests of all chains go to the GPU together
 *            (lf_align.hip), then incomplete chains are replayed.  ~99 % finish after one GPU round.
 *   D' GPU   CIGAR / MD text of every record from the paths left in HBM (src/LordFAST.cpp:1570-1763)   lf_render.hip
 *   E  host  MAPQ, SAM line assembly (src/LordFAST.cpp:318-459)
 *
 * The host never computes a DP cell: no CPU alignment, chaining or FM-index code exists in this library.
 * lf_debug_crosscheck() can switch stages B, D (walk), D' and E to host implementations that work on data copied back
 * from the device: diagnostic cross-checks for the tests, never selected automatically and not selectable by environment.
 */
#include "lf_pipe.h"
#include "lf_batch.h"

void top_push(win_t *l, int *n, int maxWin, uint32_t i, uint32_t L, float score, int isRev, int req)
{   /* src/LordFAST.cpp:634-654 */
    if (*n < maxWin) {
        win_t *b = &l[*n];
        b->tStart = i * L; b->tEnd = (i + 2) * L - 1; b->score = score; b->isReverse = (uint8_t)isRev; b->req = req;
        (*n)++;
        winh_push_heap(l, *n);
    } else if (score > l[0].score) {
        winh_pop_heap(l, *n);
        win_t *b = &l[*n - 1];
        b->tStart = i * L; b->tEnd = (i + 2) * L - 1; b->score = score; b->isReverse = (uint8_t)isRev; b->req = req;
        winh_push_heap(l, *n);
    }
}

/* device path: the read's mode, windows and fine-mode candidates as lf_vote_select_kernel decided them */
static void phase_select(ctx_t *cx, int tid, int ri)
{
    rd_t *r = &cx->reads[ri];
    const lf_params_t *p = cx->p;
    r->vote_tid = tid;
    if ((int)r->len < p->min_read_len) { r->mode = 0; return; }
    const lfg_vc_t *vc = &cx->vc;
    const int k = r->seed_idx;
    const uint32_t L = r->len;
    r->wins = (win_t *)ar_zalloc(&cx->arena[tid], ((size_t)p->max_map + 1) * sizeof(win_t));
    r->nWins = 0;
    r->mode = vc->mode[k];
    if (r->mode == 2) {
        const int rq = (int)vc->req0[k];
        const uint32_t w = vc->req_win[rq], id = w & 0x7fffffffu;
        win_t *b = &r->wins[0];
        b->tStart = id * L; b->tEnd = (id + 2) * L - 1; b->score = vc->vscore[k]; b->isReverse = (uint8_t)(w >> 31); b->req = rq;
        r->nWins = 1;
    } else if (r->mode == 3) {
        const uint32_t nc = vc->nreq[k];
        r->cands = (struct cand *)ar_alloc(&cx->arena[tid], ((size_t)nc + 1) * sizeof(struct cand));
        r->ncand = (int)nc; r->capcand = (int)nc + 1;
        for (uint32_t c = 0; c < nc; c++) {
            const int rq = (int)(vc->req0[k] + c);
            const uint32_t w = vc->req_win[rq];
            r->cands[c].win = w & 0x7fffffffu; r->cands[c].isRev = (uint8_t)(w >> 31); r->cands[c].req = rq;
        }
    }
}

/* fine mode, after the chain kernel: the heap of src/LordFAST.cpp:879-901 replayed on the chain scores */
void phase_fine_select(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (r->mode != 3) return;
    for (int c = 0; c < r->ncand; c++)
        top_push(r->wins, &r->nWins, cx->p->max_map, r->cands[c].win, r->len, cx->chain_score[r->cands[c].req], r->cands[c].isRev, r->cands[c].req);
}

/* ================================================================ phases driven by lf_map_batch */
static void phase_prepare(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    memset(&r->out, 0, sizeof r->out);
    if ((int)r->len < cx->p->min_read_len) return;
    r->seq_rev = NULL; r->qual_rev = NULL;     /* built on demand: walk_query (rare byte-string requests); SAM prints them in place */
}

static void phase_concat(ctx_t *cx, int tid, int k)
{
    (void)tid;
    const rd_t *r = &cx->reads[cx->seed_map[k]];
    lf_copy_stream(cx->cat + cx->cat_off[k], r->seq, r->len);      /* pinned staging: written once, read by the copy engine only */
}

static void phase_pack(ctx_t *cx, int tid, int k)
{
    (void)tid;
    const rd_t *r = &cx->reads[cx->seed_map[k]];
    if (!lf_pack_read(cx->pk_planes, cx->pk_qw, cx->cat_off[k], r->seq, r->len, cx->pk_xpos, cx->pk_xbyte, cx->pk_xcap, &cx->pk_xn)) cx->pk_overflow = 1;
}

void phase_make_jobs(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (r->mode < 2) return;
    r->jobs = (job_t *)ar_zalloc(&cx->arena[tid], ((size_t)r->nWins + 1) * sizeof(job_t));
    r->maps = (samlist_t *)ar_zalloc(&cx->arena[tid], ((size_t)cx->p->max_map + 1) * sizeof(samlist_t));
    /* clasp, fine mode: chain_seeds_clasp called for a window WITHOUT seeds (all of them in the neighbouring contig) sets the
     * score to -1 and leaves the previous call's chain in place (src/Chain.cpp:68,92); alignWin (src/LordFAST.cpp:1028-1046)
     * then extends that stale chain -- the read's previous window, or the last candidate calcChainScore looked at -- on the
     * window's own strand, and its record is printed once more as a secondary.  Restated: such a window takes the request of
     * the most recent chain call of this read that had seeds (none before it in the read: thread history in the reference,
     * empty here; positions above 2 * 10^9, where the reference also adds its shift to the stale chain once more: not restated). */
    int last_req = -1;
    if (cx->p->chain_alg == 1 && r->mode == 3 && !cx->host_vote)
        for (int c = 0; c < r->ncand; c++) if (cx->chain_len[r->cands[c].req] > 0) last_req = r->cands[c].req;
    for (int w = 0; w < r->nWins; w++) {
        job_t *j = &r->jobs[w];
        int rq = r->wins[w].req;
        if (cx->p->chain_alg == 1 && r->mode == 3 && !cx->host_vote) {
            if (cx->chain_len[rq] == 0 && last_req >= 0) rq = last_req;
            else if (cx->chain_len[rq] > 0) last_req = rq;
            else __atomic_fetch_add(&cx->st->n_stale_first_windows, 1, __ATOMIC_RELAXED);      /* no chain of this read to fall back on: the one corner that stays divergent */
        } else if (cx->p->chain_alg == 1 && r->mode == 2 && !cx->host_vote && cx->chain_len[rq] == 0)
            __atomic_fetch_add(&cx->st->n_stale_first_windows, 1, __ATOMIC_RELAXED);          /* the same corner in coarse mode: the read's ONE window has no seeds of its own contig */
        j->req = rq;
        j->read = ri; j->widx = w; j->isRev = r->wins[w].isReverse;
        j->chainLen = cx->chain_len[rq];
        /* the chain is only READ on the host, and only by the replay of the few chains that leave the common path (lf_replay.c): it stays
         * where the chain stage left it (the lane's pinned slot, valid until this lane's next lfg_vote_chain) -- no copy per job */
        if (!cx->host_vote) j->chain = cx->vc.chain_seeds ? cx->vc.chain_seeds + cx->vc.chain_off[rq] : NULL;      /* NULL: fetched with the open list, if the job ends up on it */
        else {
            j->chain = (Seed_t *)ar_alloc(&cx->arena[tid], ((size_t)j->chainLen + 1) * sizeof(Seed_t));
            const creq_t *cq = &cx->creq[rq];
            for (uint32_t k = 0; k < j->chainLen; k++) j->chain[k] = cx->cseeds[cq->off + cx->chain_idx[cq->off + k]];
        }
        j->complete = (j->chainLen <= 1);                   /* nothing to extend: totalScore = -2L (:1089) */
    }
}

static int cmp_int(const void *a, const void *b) { const int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }
static void phase_select_jobs(ctx_t *cx, int tid, int ri) { phase_select(cx, tid, ri); phase_fine_select(cx, tid, ri); phase_make_jobs(cx, tid, ri); }

/* the device walk's records (lfg_walk_emit) become the jobs' mappings; the chains it left to the host are collected */
static void phase_take_wrec(ctx_t *cx, int tid, int k)
{
    job_t *j = cx->wj_owner[k];
    const lf_wrec_t *wr = (const lf_wrec_t *)cx->wj_rec + k;
    if (wr->rare) { cx->open[__atomic_fetch_add(&cx->n_rare, 1, __ATOMIC_RELAXED)] = k; return; }
    rd_t *r = &cx->reads[j->read];
    samlist_t *map = &r->maps[j->widx];
    sam_t tmp; memset(&tmp, 0, sizeof tmp);
    tmp.flag = j->isRev ? 16 : 0; tmp.pos = wr->pos; tmp.posEnd = wr->posEnd; tmp.qStart = wr->qStart; tmp.qEnd = wr->qEnd; tmp.nmCount = wr->nm;
    samlist_clear(map);
    samlist_push(map, &tmp, NULL, NULL, &cx->arena[tid]);
    map->v[0].rec = k; map->v[0].rtid = -2;                 /* record k of the device-planned recipe */
    j->complete = 1;
}
static void phase_walk_open(ctx_t *cx, int tid, int k);

static void phase_walk(ctx_t *cx, int tid, int ri)
{
    rd_t *r = &cx->reads[ri];
    if (r->mode < 2) return;
    for (int w = 0; w < r->nWins; w++) {
        job_t *j = &r->jobs[w];
        if (j->complete) continue;
        walk_chain(cx, tid, j, &r->maps[w]);
    }
}

static void phase_walk_open(ctx_t *cx, int tid, int k) { phase_walk(cx, tid, cx->open[k]); }

static void phase_merge_desc(ctx_t *cx, int tid, int t)
{
    (void)tid;
    stage_t *s = &cx->stages[t];
    int g = cx->mg_gbase[t];
    const uint64_t ob = cx->mg_qbase[t];
    for (int k = 0; k < s->dn; k++, g++) {
        cx->mg_desc[g] = s->dd[k];
        cx->mg_R->ops_off[g] = ob + s->dops[k];
        memo_t *m = &cx->edd_jobs[t].job[k]->memo[s->downer[k]];
        m->round = cx->mg_round; m->slot = g;
    }
    s->dn = 0; s->dops_total = 0; cx->edd_jobs[t].n = 0;
}


/* E is three passes so that the SAM text is formatted straight into its final place:
 * score + order the mappings; print in COUNT mode (exact record sizes); print again into the final buffer */
static void phase_sam_score(ctx_t *cx, int tid, int ri)
{
    (void)tid;
    rd_t *r = &cx->reads[ri];
    if (r->mode < 2) {
        r->maps = (samlist_t *)ar_zalloc(&cx->arena[tid], 2 * sizeof(samlist_t));
        if (cx->stage_sink) { int32_t *o = (int32_t *)malloc(8); o[0] = r->mode; o[1] = 0; cx->stage_sink[cx->stage_i0 + ri] = o; }
        return;
    }
    for (int w = 0; w < r->nWins; w++) score_mapping(cx->p, &r->maps[w], r->wins[w].isReverse, r->len, r->jobs[w].chainLen);
    if (cx->stage_sink) {
        /* the stage view (lf_map_stages_batch): what mapSeq holds between alignWin and the sort / print (src/LordFAST.cpp:547, :561):
         * { mode, n_wins, then per window tStart, tEnd, isReverse, score bits, totalScore, n_records, 7 ints per record } */
        size_t words = 2;
        for (int w = 0; w < r->nWins; w++) words += 6 + 7 * (size_t)r->maps[w].n;
        int32_t *o = (int32_t *)malloc(words * 4); size_t k = 0;
        o[k++] = r->mode; o[k++] = r->nWins;
        for (int w = 0; w < r->nWins; w++) {
            const win_t *wn = &r->wins[w]; const samlist_t *m = &r->maps[w];
            o[k++] = (int32_t)wn->tStart; o[k++] = (int32_t)wn->tEnd; o[k++] = wn->isReverse; memcpy(&o[k++], &wn->score, 4); o[k++] = m->totalScore; o[k++] = m->n;
            for (int j = 0; j < m->n; j++) { const sam_t *x = &m->v[j]; o[k++] = (int32_t)x->pos; o[k++] = (int32_t)x->posEnd; o[k++] = (int32_t)x->qStart; o[k++] = (int32_t)x->qEnd; o[k++] = x->flag; o[k++] = x->alnScore; o[k++] = x->nmCount; }
        }
        cx->stage_sink[cx->stage_i0 + ri] = o;
    }
    if (r->mode == 3) samsort_sort(r->maps, r->nWins);                     /* std::sort(compareSam) :565 */
}
/* what the Hirschberg levels of the lane's alignment calls left in lane values 4 .. 6 (lf_align.hip): longest root, nodes by kind of sweep */
static void take_hirsch_stats(int device, lf_stats_t *st)
{
    const uint64_t mx = lfg_lane_value(device, 4);
    if (mx > st->hirsch_max_rows) st->hirsch_max_rows = mx;
    st->hirsch_banded_nodes += lfg_lane_value(device, 5); st->hirsch_unbanded_nodes += lfg_lane_value(device, 6);
    lfg_lane_set_value(device, 4, 0); lfg_lane_set_value(device, 5, 0); lfg_lane_set_value(device, 6, 0);
}
static inline void tmark(ctx_t *cx, const char *label)
{
    if (cx->lane >= 0 && cx->lane < LF_MAX_LANES) g_lane_mark[cx->lane] = label;
    if (cx->timing && cx->n_marks < 96) { cx->marks[cx->n_marks].label = label; cx->marks[cx->n_marks].t = now_ms(); cx->n_marks++; }
}
static void tmark_dump(ctx_t *cx, double t_begin)
{
    if (!cx->timing) return;
    char line[4096]; int o = 0; double prev = t_begin;
    for (int k = 0; k < cx->n_marks && o < 3900; k++) { o += snprintf(line + o, sizeof line - (size_t)o, " %s %.1f", cx->marks[k].label, cx->marks[k].t - prev); prev = cx->marks[k].t; }
    fprintf(stderr, "[lf] lane %d timeline (ms per step, start t=%.1f):%s\n", cx->lane, t_begin, line);
}

/* ---------------------------------------------------------------- one chunk of reads through all stages */
int map_chunk(ctx_t *cx)
{
    const int n = cx->n_reads, nt = cx->n_threads;
    lf_stats_t *st = cx->st;
    int rc = LF_OK;
    double t0 = now_ms(), t1;
    const int timing = lf_env_set("LF_TIMING"), timing0 = timing;
    cx->timing = timing; cx->n_marks = 0;
    const double t_begin = t0;

    parallel_for(cx, n, phase_prepare);
    tmark(cx, "prepare");
    if (lf_env_set("LF_TIMING")) fprintf(stderr, "[lf] prepare %.1f ms\n", now_ms() - t0);

    /* ---- A: seeds ---- */
    lfg_hits_t hits; memset(&hits, 0, sizeof hits);
    {
        int *map = (int *)malloc((size_t)n * sizeof(int)); int m = 0;
        uint64_t bases = 0;
        for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) { cx->reads[i].seed_idx = m; map[m++] = i; bases += cx->reads[i].len; }
        if (m) {
            char *cat = cx->d_seqs ? NULL : (char *)lfg_pin_slot(LF_PS_READS, bases + 64);
            uint64_t *off = (uint64_t *)lfg_pin_slot(LF_PS_READOFF, ((size_t)m + 1) * 8 * (cx->d_seqs ? 2 : 1));
            if ((!cat && !cx->d_seqs) || !off) { free(map); return LF_ERR_NOMEM; }
            uint64_t o = 0;
            for (int k = 0; k < m; k++) { off[k] = o; o += cx->reads[map[k]].len; }
            off[m] = o;
            cx->seed_map = map; cx->cat = cat; cx->cat_off = off;
            double tc0 = now_ms();
            if (cx->d_seqs) {      /* bases already in HBM: a device gather replaces the host concatenation + H2D copy */
                uint64_t *so = off + m + 1;
                for (int k = 0; k < m; k++) so[k] = cx->reads[map[k]].src_off;
                rc = lfg_seed_src(cx->ix, cx->p, m, NULL, cx->d_seqs, so, off, cx->host_vote, &hits);
            } else {
            /* the lanes of a step all start with this copy: one at a time, with every pool thread on it, so that the first lane's
             * bases are ready (and on their way, see lfg_seed_src) after 1 / 8 of the time instead of all lanes' after all of it */
            /* (one turn per DEVICE since round 6, like the upload turn of lf_seed.hip: the lanes of different devices in one process -- lf_map_batch_multi -- do not
             * wait for each other's packing; a turn is never held across a GPU wait) */
            static pthread_mutex_t concat_turns[16] = { [0 ... 15] = PTHREAD_MUTEX_INITIALIZER };
            pthread_mutex_t *concat_turn = &concat_turns[cx->ix->device & 15];
            /* Packed upload: the pool threads turn the reads into the three bit planes the alignment kernels work on anyway
             * (lo / hi / valid: 3 / 8 of the bytes; the device rebuilds the bytes for the seed search and the SAM writer) plus a
             * list of the bytes that are not upper-case ACGT -- the link carries 0.57 instead of 1.53 GB per 100 k reads, and the
             * lanes of a step start 1.3 instead of 3.4 ms apart.  A chunk with more than one such byte in 64 (lower-case reads)
             * goes up as bytes.  LF_UPLOAD_PACKED=0: always bytes. */
            const int packed_on = lf_env_long("LF_UPLOAD_PACKED", 1) != 0;
            int packed = 0;
            if (packed_on && !cx->host_vote && cx->pre && cx->pre->min_read_len == cx->p->min_read_len) {
                /* the batch was packed when it was made (lf_batch.h): this chunk's planes are a bit range of the batch's -- nothing to do on the host but
                 * to say where it is; the device shifts it into place (lf_seed.hip) */
                const lf_prepack_t *P = cx->pre;
                const uint64_t o0 = P->boff[cx->pre_i0];
                lf_packed_src_t pk; memset(&pk, 0, sizeof pk);
                pk.planes = P->planes; pk.qw = (bases + 63) / 64 + 2; pk.src_qw = P->QW; pk.word0 = o0 >> 6; pk.shift = (uint32_t)(o0 & 63); pk.exc_base = o0;
                uint64_t e0 = 0, e1 = P->n_exc;
                { uint64_t lo_ = 0, hi_ = P->n_exc; while (lo_ < hi_) { const uint64_t md = (lo_ + hi_) >> 1; if (P->exc_pos[md] < o0) lo_ = md + 1; else hi_ = md; } e0 = lo_; }
                { uint64_t lo_ = e0, hi_ = P->n_exc; while (lo_ < hi_) { const uint64_t md = (lo_ + hi_) >> 1; if (P->exc_pos[md] < o0 + bases) lo_ = md + 1; else hi_ = md; } e1 = lo_; }
                pk.exc_pos = P->exc_pos + e0; pk.exc_byte = P->exc_byte + e0; pk.n_exc = e1 - e0;
                if (P->boff[cx->pre_i0 + n] - o0 == bases) {
                    tmark(cx, "prepacked");
                    rc = lfg_seed_packed(cx->ix, cx->p, m, &pk, off, cx->host_vote, &hits);
                    packed = 1;
                }
            }
            if (!packed && packed_on && !cx->host_vote) {
                const uint64_t qw = (bases + 63) / 64 + 2;                  /* lf_plane_words (lf_rsweep.h) */
                const uint64_t xcap = bases / 64 + 1024;
                uint64_t *xpos = (uint64_t *)lfg_pin_slot(LF_PS_EXC_POS, xcap * 8); uint8_t *xbyte = (uint8_t *)lfg_pin_slot(LF_PS_EXC_BYTE, xcap);
                if (xpos && xbyte && 3 * qw * 8 <= bases + 64) {            /* the planes take the staging buffer's place */
                    uint64_t *planes = (uint64_t *)cat;
                    cx->pk_planes = planes; cx->pk_qw = qw; cx->pk_xpos = xpos; cx->pk_xbyte = xbyte; cx->pk_xcap = xcap; cx->pk_xn = 0; cx->pk_overflow = 0;
                    pthread_mutex_lock(concat_turn);
                    /* the words that hold a read boundary (two reads OR their bits in) and the slack behind the last base */
                    for (int x = 0; x < 3; x++) {
                        uint64_t *P = planes + (size_t)x * qw;
                        for (int k = 0; k <= m; k++) P[off[k] >> 6] = 0;
                        for (uint64_t w = off[m] >> 6; w < qw; w++) P[w] = 0;
                    }
                    parallel_for(cx, m, phase_pack);
                    pthread_mutex_unlock(concat_turn);
                    tmark(cx, "pack");
                    if (!cx->pk_overflow) {
                        lf_packed_src_t pk; memset(&pk, 0, sizeof pk); pk.planes = planes; pk.qw = qw; pk.exc_pos = xpos; pk.exc_byte = xbyte; pk.n_exc = cx->pk_xn;
                        if (lf_env_set("LF_TIMING")) fprintf(stderr, "[lf] pack %.1f ms, %llu bytes outside ACGT\n", now_ms() - tc0, (unsigned long long)pk.n_exc);
                        tc0 = now_ms();
                        rc = lfg_seed_packed(cx->ix, cx->p, m, &pk, off, cx->host_vote, &hits);
                        packed = 1;
                    }
                }
            }
            if (!packed) {
            { pthread_mutex_lock(concat_turn); parallel_for(cx, m, phase_concat); pthread_mutex_unlock(concat_turn); }
            tmark(cx, "concat");
            if (lf_env_set("LF_TIMING")) fprintf(stderr, "[lf] concat %.1f ms\n", now_ms() - tc0);
            tc0 = now_ms();
            rc = lfg_seed(cx->ix, cx->p, m, cat, off, cx->host_vote, &hits);
            }
            }
            tmark(cx, "SEED");
            if (lf_env_set("LF_TIMING")) fprintf(stderr, "[lf] lfg_seed %.1f ms (search %.1f locate %.1f), %llu hits\n", now_ms() - tc0, hits.ms_search, hits.ms_locate, (unsigned long long)hits.n_hits);
            if (rc != LF_OK) { free(map); return rc; }
            /* the vote stage sorts 2 keys per hit with 31-bit indices: a chunk above the limit is cut in two by the
             * caller and mapped again (the reference has no such limit; it must stay an implementation detail) */
            if (hits.n_hits >= cx->max_chunk_hits && n > 1) { free(map); cx->seed_map = NULL; return LF_RC_SPLIT; }
            st->n_seeds += hits.n_hits; st->n_cache += hits.counters[0]; st->n_occblk += hits.counters[1]; st->n_sa += hits.counters[2]; st->n_readbytes += hits.counters[3];
            st->ms_k_search += hits.ms_search; st->ms_k_accept += hits.ms_accept; st->ms_k_locate += hits.ms_locate;
            st->search_launches++; st->locate_launches++;
        }
        free(map); cx->seed_map = NULL;
        cx->hits = &hits;
    }
    double tstage[8] = { 0 };
    t1 = now_ms(); st->ms_seed += t1 - t0; tstage[0] = t1 - t0; t0 = t1;

    const int host_vote = cx->host_vote;
    if (!host_vote) {
        /* ---- B + C on the device: votes -> candidate windows -> sorted requests -> chains; only chains come back ---- */
        uint32_t max_len = 0;
        for (int i = 0; i < n; i++) if (cx->reads[i].len > max_len) max_len = cx->reads[i].len;
        int m = 0;
        for (int i = 0; i < n; i++) if ((int)cx->reads[i].len >= cx->p->min_read_len) m++;
        rc = lfg_vote_chain(cx->ix, cx->p, m, hits.n_hits, max_len, &cx->vc);
        if (timing0) fprintf(stderr, "[lf] lfg_vote_chain %.1f ms (vote %.1f chain %.1f), %d requests, %llu request seeds, %llu chain seeds\n", now_ms() - t0,
                             cx->vc.ms_vote, cx->vc.ms_chain, cx->vc.n_req, (unsigned long long)cx->vc.n_req_seeds, (unsigned long long)cx->vc.n_chain_seeds);
        tmark(cx, "VOTECHAIN");
        if (rc != LF_OK) return rc;
        cx->n_creq = cx->vc.n_req; cx->chain_len = cx->vc.chain_len; cx->chain_score = cx->vc.chain_score;
        /* every chain is walked on the host in the cross-check modes: all of them come back; otherwise only the ones the device walk leaves open (extend) */
        if (cx->host_cigar || (g_crosscheck & LF_XC_HOST_WALK)) { rc = lfg_vote_fetch_chains(cx->ix, &cx->vc); if (rc != LF_OK) return rc; }
        st->ms_k_vote += cx->vc.ms_vote; st->ms_k_chain += cx->vc.ms_chain; st->n_chain_problems += (uint64_t)cx->vc.n_req;
        st->n_req_seeds += cx->vc.n_req_seeds; st->n_tie_requests += cx->vc.n_tie_req;
        t1 = now_ms(); st->ms_vote += t1 - t0; tstage[1] = t1 - t0; t0 = t1;
        parallel_for(cx, n, phase_select_jobs);            /* selection, fine-mode heap and the jobs of a read: one pass over the reads */
        tmark(cx, "select+jobs");
        t1 = now_ms(); st->ms_chain += t1 - t0; tstage[2] = t1 - t0; t0 = t1;
        goto extend;
    }
    /* ---- B + C through the host cross-check (lf_debug_crosscheck bit 0): vote, selection, std::sort on the host, chains on the device ---- */
    rc = g_xc.vote_chain ? g_xc.vote_chain(cx) : LF_ERR_ARG;
    if (rc != LF_OK) return rc;
    t1 = now_ms(); st->ms_vote += t1 - t0; tstage[1] = t1 - t0; t0 = t1;
    parallel_for(cx, n, phase_fine_select);
    parallel_for(cx, n, phase_make_jobs);
    t1 = now_ms(); st->ms_chain += t1 - t0; tstage[2] = t1 - t0; t0 = t1;

extend:
    /* ---- D: extension rounds ---- */
    cx->stages = (stage_t *)calloc((size_t)nt, sizeof(stage_t));
    cx->ed_jobs = (jobvec_t *)calloc((size_t)nt, sizeof(jobvec_t));
    cx->ksw_jobs = (jobvec_t *)calloc((size_t)nt, sizeof(jobvec_t));
    cx->edd_jobs = (jobvec_t *)calloc((size_t)nt, sizeof(jobvec_t));
    /* ---- D0: the common path of alignChain_edlib on the device (lf_walk.hip): pieces -> descriptors -> alignments ->
     * clip / split triggers -> record fields + CIGAR recipe, all in HBM.  Chains that leave the common path (and everything
     * when a host cross-check mode is on) are replayed by the host walk below, which then finds them incomplete. ---- */
    cx->n_dev_recs = 0; cx->n_dev_items = 0; cx->d_dev_recs = NULL; cx->d_dev_items = NULL;
    free(cx->open); cx->open = NULL; cx->n_open = 0;
    if (!host_vote && !cx->host_cigar && !(g_crosscheck & LF_XC_HOST_WALK)) {
        int nj = 0;
        for (int i = 0; i < n; i++) { const rd_t *r = &cx->reads[i]; if (r->mode >= 2) for (int w = 0; w < r->nWins; w++) nj += r->jobs[w].chainLen > 1; }
        if (nj > 0) {
            lf_wjob_t *wj = (lf_wjob_t *)lfg_pin_slot(LF_PS_WALK0 + 2, (size_t)nj * sizeof(lf_wjob_t));
            job_t **owner = (job_t **)malloc((size_t)nj * sizeof(job_t *));
            if (!wj || !owner) { free(owner); return LF_ERR_NOMEM; }
            int k = 0;
            for (int i = 0; i < n; i++) {
                rd_t *r = &cx->reads[i];
                if (r->mode < 2) continue;
                for (int w = 0; w < r->nWins; w++) {
                    job_t *j = &r->jobs[w];
                    if (j->chainLen <= 1) continue;
                    wj[k].req = (uint32_t)j->req; wj[k].read = (uint32_t)r->seed_idx; wj[k].chain_len = j->chainLen; wj[k].is_rev = (uint8_t)j->isRev;
                    wj[k].pad[0] = wj[k].pad[1] = wj[k].pad[2] = 0;
                    owner[k++] = j;
                }
            }
            lfg_walk_t W;
            rc = lfg_walk_plan(cx->ix, nj, wj, &cx->vc, cx->lazy, &W);
            tmark(cx, "WALKPLAN");
            if (rc != LF_OK) { free(owner); return rc; }
            void *dops = NULL, *d_ed = NULL, *d_end = NULL, *d_len = NULL; float ms = 0;
            if (W.n_desc) rc = lfg_edlib_desc_dev(cx->ix, (int)W.n_desc, W.d_desc, W.d_opsoff, W.ops_total, &W.hc, LF_DS_RND0 + 0, &dops, &d_ed, &d_end, &d_len, &ms);
            tmark(cx, "EDLIB0");
            if (rc != LF_OK) { free(owner); return rc; }
            lf_wrec_t *wrec = NULL;
            rc = lfg_walk_emit(cx->ix, &cx->vc, cx->lazy, &W, d_ed, d_end, d_len, &wrec);
            tmark(cx, "WALKEMIT");
            if (rc != LF_OK) { free(owner); return rc; }
            {   /* extension round 0 lives on the device only */
                ed_round_t R; memset(&R, 0, sizeof R);
                R.n = (int)W.n_desc; R.pinned = 1; R.ops_bytes = W.ops_total; R.d_ops = (uint8_t *)dops; R.d_desc = W.d_desc; R.lazy = cx->lazy;
                cx->ed_rounds = (ed_round_t *)realloc(cx->ed_rounds, ((size_t)cx->n_ed_rounds + 1) * sizeof(ed_round_t));
                cx->ed_rounds[cx->n_ed_rounds++] = R;
            }
            if (W.n_desc) ms = lfg_edlib_round_ms();           /* (the round returned without waiting: its events are complete now, behind lfg_walk_emit's wait) */
            st->ms_k_edlib += ms; st->n_edlib_problems += W.n_desc; st->edlib_launches += 1; st->ops_bytes += W.ops_total;
            if (W.n_desc) { float bd[4]; lfg_edlib_breakdown(bd); st->ms_k_rsweep += bd[0]; st->ms_k_tb += bd[1]; st->ms_k_hirsch += bd[2]; st->ms_k_bin += bd[3]; }
            st->ext_bytes += W.ext_bytes; st->dp_block_steps += W.block_steps;
            st->hirsch_bytes += 2 * W.hc.sum_n + W.hc.sum_m + (W.hc.sum_m + 3) / 4;
            take_hirsch_stats(cx->ix->device, st);
            /* (the pool: one sam_t per job, 100 k of them per step) */
            cx->open = (int *)malloc(((size_t)nj + 1) * sizeof(int));
            if (!cx->open) { free(owner); return LF_ERR_NOMEM; }
            cx->wj_owner = owner; cx->wj_rec = wrec; cx->n_rare = 0;
            parallel_for(cx, nj, phase_take_wrec);
            const int n_rare = cx->n_rare;
            /* rare jobs -> the reads that hold them, each once, in read order (the replay's requests are staged in this order) */
            for (k = 0; k < n_rare; k++) cx->open[k] = owner[cx->open[k]]->read;
            qsort(cx->open, (size_t)n_rare, sizeof(int), cmp_int);
            { int u = 0; for (k = 0; k < n_rare; k++) if (u == 0 || cx->open[u - 1] != cx->open[k]) cx->open[u++] = cx->open[k]; cx->n_open = u; }
            cx->wj_owner = NULL; cx->wj_rec = NULL;
            if (timing) fprintf(stderr, "[lf] device walk: %d jobs, %llu pieces aligned, %d jobs (%.1f %%) left to the host replay\n", nj, (unsigned long long)W.n_desc, n_rare, 100.0 * n_rare / nj);
            cx->n_dev_recs = nj; cx->n_dev_items = W.n_items; cx->d_dev_recs = W.d_recs; cx->d_dev_items = W.d_items;
            free(owner);
        }
    }
    if (cx->open ? cx->n_open > 0 : cx->d_seqs != NULL) {
        /* What the host replay reads, fetched together with ONE gather kernel, one copy, one wait: the chains of the jobs the device walk left open
         * (the chain stage keeps all chains in HBM; round 5 still copied every chain of every chunk back -- 64 MB per 100 k reads -- for the one or
         * two chains per chunk the host ever looks at) and, for lf_map_batch_dev, those reads' bases (rd_host_bases would get them one by one, each
         * copy behind a lock and a wait).  Without an open list (cross-check paths: every chain is walked on the host, all chains were copied back
         * after the chain stage) only the bases. */
        const int nscan = cx->open ? cx->n_open : n;
        int nf = 0; size_t tot = 0;
        char *buf = NULL; uint64_t *hoff = NULL; const void **src = NULL; size_t *nb = NULL; uint8_t *term = NULL;
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) {
                if (!nf) break;
                buf = (char *)lfg_pin_slot(LF_PS_HOSTBASES, tot + 64);
                hoff = (uint64_t *)malloc((size_t)nf * sizeof(uint64_t)); src = (const void **)malloc((size_t)nf * sizeof(void *)); nb = (size_t *)malloc((size_t)nf * sizeof(size_t));
                term = (uint8_t *)malloc((size_t)nf);
                if (!buf || !hoff || !src || !nb || !term) { free(hoff); free(src); free(nb); free(term); return LF_ERR_NOMEM; }
            }
            int k = 0; size_t o = 0;
            for (int x = 0; x < nscan; x++) {
                rd_t *r = &cx->reads[cx->open ? cx->open[x] : x];
                if (r->mode < 2) continue;
                int open = 0;
                for (int w = 0; w < r->nWins; w++) open |= !r->jobs[w].complete;
                if (!open) continue;
                if (cx->d_seqs && !r->seq) {                /* (a read is visited once per pass: still NULL in the second) */
                    if (pass == 1) { hoff[k] = o; src[k] = cx->d_seqs + r->src_off; nb[k] = r->len; term[k] = 1; r->seq = buf + o; }
                    k++; o += (size_t)r->len + 1;
                    if (r->isFq) { if (pass == 1) { hoff[k] = o; src[k] = cx->d_quals + r->src_off; nb[k] = r->len; term[k] = 1; r->qual = buf + o; } k++; o += (size_t)r->len + 1; }
                }
                if (cx->open && !cx->vc.chain_seeds) for (int w = 0; w < r->nWins; w++) {
                    job_t *j = &r->jobs[w];
                    if (j->complete || j->chainLen == 0) continue;
                    o = (o + 7) & ~(size_t)7;
                    if (pass == 1) { hoff[k] = o; src[k] = (const Seed_t *)cx->vc.d_chain_seeds + cx->vc.chain_off[j->req]; nb[k] = (size_t)j->chainLen * sizeof(Seed_t); term[k] = 0; j->chain = (Seed_t *)(buf + o); }
                    k++; o += (size_t)j->chainLen * sizeof(Seed_t);
                }
            }
            if (pass == 0) { nf = k; tot = o; continue; }
            rc = lfg_fetch_gather(cx->ix->device, k, buf, hoff, src, nb, o);
            /* (the copy brings the staging buffer's gaps along: the strings' terminators are written after it) */
            for (int i = 0; i < k; i++) if (term[i]) buf[hoff[i] + nb[i]] = 0;
            free(hoff); free(src); free(nb); free(term);
            if (rc != LF_OK) return rc;
            tmark(cx, "fetch");
        }
    }
    for (int round = 0; round < 64; round++) {
        double tw0 = now_ms();
        if (cx->open) {
            /* only the reads that still have an open job; the list shrinks as their jobs complete */
            if (cx->n_open) parallel_for(cx, cx->n_open, phase_walk_open);
            int u = 0;
            for (int x = 0; x < cx->n_open; x++) {
                const rd_t *r = &cx->reads[cx->open[x]];
                int open = 0;
                for (int w = 0; w < r->nWins; w++) open |= !r->jobs[w].complete;
                if (open) cx->open[u++] = cx->open[x];
            }
            cx->n_open = u;
        } else parallel_for(cx, n, phase_walk);
        tmark(cx, "walk");
        if (timing) fprintf(stderr, "[lf] round %d walk %.1f ms\n", round, now_ms() - tw0);
        int nk = 0, nd = 0;
        for (int t = 0; t < nt; t++) { nk += cx->stages[t].kn; nd += cx->stages[t].dn; }
        if (nk == 0 && nd == 0) break;
        if (nd) {
            /* descriptor requests: nothing but 32-byte descriptors goes to the GPU */
            uint64_t *obase = (uint64_t *)malloc((size_t)nt * 8); int *gbase = (int *)malloc((size_t)nt * sizeof(int));
            uint64_t ops_total = 0;
            { int g0 = 0; for (int t = 0; t < nt; t++) { obase[t] = ops_total; gbase[t] = g0; ops_total += cx->stages[t].dops_total; g0 += cx->stages[t].dn; st->ext_bytes += cx->stages[t].ext_bytes; cx->stages[t].ext_bytes = 0; st->dp_block_steps += cx->stages[t].blk_steps; cx->stages[t].blk_steps = 0; } }
            const int ridx = cx->n_ed_rounds;
            const int pin = ridx < 16;
            ed_round_t R; memset(&R, 0, sizeof R);
            R.n = nd; R.pinned = pin; R.ops_bytes = ops_total;
            /* the edit paths stay in HBM (slot of this round) unless the host builds the strings itself */
            const int host_ops = cx->host_cigar || ridx >= LF_MAX_ED_ROUNDS;
            if (pin) {
                R.ed = (int32_t *)lfg_pin_slot(LF_PS_ROUND0 + 4 * ridx, (size_t)nd * 4); R.end = (int32_t *)lfg_pin_slot(LF_PS_ROUND0 + 4 * ridx + 1, (size_t)nd * 4);
                R.ops_len = (uint32_t *)lfg_pin_slot(LF_PS_ROUND0 + 4 * ridx + 2, (size_t)nd * 4);
                if (host_ops) R.ops = (uint8_t *)lfg_pin_slot(LF_PS_ROUND0 + 4 * ridx + 3, ops_total + 1);
            } else {
                R.ed = (int32_t *)malloc((size_t)nd * 4); R.end = (int32_t *)malloc((size_t)nd * 4);
                R.ops_len = (uint32_t *)malloc((size_t)nd * 4); R.ops = (uint8_t *)malloc(ops_total + 1);
            }
            R.ops_off = (uint64_t *)malloc((size_t)nd * 8);
            lf_aln_desc_t *desc = (lf_aln_desc_t *)lfg_pin_slot(LF_PS_ALN_PROB, (size_t)nd * sizeof(lf_aln_desc_t));
            if (!desc || !R.ed || !R.end || !R.ops_len || (host_ops && !R.ops)) return LF_ERR_NOMEM;
            cx->mg_desc = desc; cx->mg_R = &R; cx->mg_qbase = obase; cx->mg_gbase = gbase; cx->mg_round = ridx;
            double tm0 = now_ms();
            parallel_for(cx, nt, phase_merge_desc);
            tmark(cx, "merge");
            free(obase); free(gbase);
            float ms = 0;
            void *dops = NULL, *ddesc = NULL;
            rc = lfg_edlib_desc(cx->ix, nd, desc, R.ops_off, ops_total, R.ed, R.end, R.ops, R.ops_len,
                                LF_DS_RND0 + 2 * (ridx < LF_MAX_ED_ROUNDS ? ridx : 0), &dops, &ddesc, &ms);
            if (!host_ops) { R.d_ops = (uint8_t *)dops; R.d_desc = ddesc; }
            R.lazy = cx->lazy;
            tmark(cx, "EDLIB");
            if (timing) fprintf(stderr, "[lf] round %d: %d descriptor problems, merge+solve %.1f ms (kernels %.1f ms), ops %.1f MB\n", round, nd, now_ms() - tm0, ms, ops_total / 1e6);
            cx->ed_rounds = (ed_round_t *)realloc(cx->ed_rounds, ((size_t)cx->n_ed_rounds + 1) * sizeof(ed_round_t));
            cx->ed_rounds[cx->n_ed_rounds++] = R;
            if (rc != LF_OK) return rc;
            st->ms_k_edlib += ms; st->n_edlib_problems += (uint64_t)nd; st->edlib_launches += 1; st->ops_bytes += ops_total;
            for (int i = 0; i < nd; i++) if (desc[i].n && desc[i].m && 20ull * ((desc[i].n + 63) / 64) * desc[i].m + 8ull * desc[i].m >= 1024 * 1024)      /* edlib's traceback switch (lib/edlib/edlib.cpp:1117-1119) */
                st->hirsch_bytes += 2ull * desc[i].n + desc[i].m + (desc[i].m + 3) / 4;
            { float bd[4]; lfg_edlib_breakdown(bd); st->ms_k_rsweep += bd[0]; st->ms_k_tb += bd[1]; st->ms_k_hirsch += bd[2]; st->ms_k_bin += bd[3]; }
            take_hirsch_stats(cx->ix->device, st);
        }
        if (nk) {
            uint64_t qn = 0, tn = 0;
            for (int t = 0; t < nt; t++) { qn += cx->stages[t].kqn; tn += cx->stages[t].ktn; }
            uint8_t *qb = (uint8_t *)malloc(qn + 1), *tb = (uint8_t *)malloc(tn + 1);
            uint64_t *qoff = (uint64_t *)malloc(((size_t)nk + 1) * 8), *toff = (uint64_t *)malloc(((size_t)nk + 1) * 8);
            int32_t *prm = (int32_t *)malloc((size_t)nk * 7 * 4);
            ksw_round_t R; memset(&R, 0, sizeof R);
            R.n = nk; R.score = (int32_t *)malloc((size_t)nk * 4); R.qle = (int32_t *)malloc((size_t)nk * 4); R.tle = (int32_t *)malloc((size_t)nk * 4);
            const int ridx = cx->n_ksw_rounds;
            int g = 0; uint64_t qo = 0, to = 0;
            for (int t = 0; t < nt; t++) {
                stage_t *s = &cx->stages[t];
                memcpy(qb + qo, s->kq, s->kqn); memcpy(tb + to, s->kt, s->ktn);
                for (int k = 0; k < s->kn; k++, g++) {
                    qoff[g] = qo + s->kqoff[k]; toff[g] = to + s->ktoff[k];
                    memcpy(prm + 7 * g, s->kprm + 7 * k, 28);
                    memo_t *m = &cx->ksw_jobs[t].job[k]->memo[(uintptr_t)s->kowner[k]];
                    m->round = ridx; m->slot = g;
                }
                qo += s->kqn; to += s->ktn;
                s->kn = 0; s->kqn = 0; s->ktn = 0; cx->ksw_jobs[t].n = 0;
            }
            qoff[nk] = qo; toff[nk] = to;
            float ms = 0;
            rc = lfg_ksw(cx->ix->device, nk, qb, qoff, tb, toff, prm, R.score, R.qle, R.tle, &ms);
            tmark(cx, "KSW");
            free(qb); free(tb); free(qoff); free(toff); free(prm);
            cx->ksw_rounds = (ksw_round_t *)realloc(cx->ksw_rounds, ((size_t)cx->n_ksw_rounds + 1) * sizeof(ksw_round_t));
            cx->ksw_rounds[cx->n_ksw_rounds++] = R;
            if (rc != LF_OK) return rc;
            st->ms_k_ksw += ms; st->n_ksw_problems += (uint64_t)nk; st->ksw_bytes += qn + tn + 12ull * (uint64_t)nk;
        }
    }
    t1 = now_ms(); st->ms_extend += t1 - t0; tstage[3] = t1 - t0; t0 = t1;

    /* ---- D': CIGAR / MD text of every record, rendered on the GPU from the paths in HBM ---- */
    if (!cx->host_cigar) {
        uint64_t n_items = 0; int n_recs = 0;
        cx->rrbase = (int *)malloc((size_t)nt * sizeof(int));
        uint64_t *ibase = (uint64_t *)malloc((size_t)nt * 8);
        for (int t = 0; t < nt; t++) { cx->rrbase[t] = cx->n_dev_recs + n_recs; ibase[t] = cx->n_dev_items + n_items; n_recs += cx->stages[t].rrn; n_items += cx->stages[t].rin; }
        if (n_recs + cx->n_dev_recs) {
            if (n_items + cx->n_dev_items >= 0xffffffffull) { lf_set_error("lf_map_batch: too many CIGAR pieces in one chunk"); free(ibase); return LF_ERR_ARG; }
            lf_ritem_t *items = (lf_ritem_t *)lfg_pin_slot(LF_PS_RENDER0 + 3, (n_items + 1) * sizeof(lf_ritem_t));
            lf_rrecord_t *recs = (lf_rrecord_t *)lfg_pin_slot(LF_PS_RENDER0 + 4, ((size_t)n_recs + 1) * sizeof(lf_rrecord_t));
            if (!items || !recs) { free(ibase); return LF_ERR_NOMEM; }
            for (int t = 0; t < nt; t++) {
                const stage_t *s = &cx->stages[t];
                if (s->rin) memcpy(items + (ibase[t] - cx->n_dev_items), s->ri, s->rin * sizeof(lf_ritem_t));
                for (int k = 0; k < s->rrn; k++) { lf_rrecord_t q = s->rr[k]; q.item0 += (uint32_t)ibase[t]; recs[cx->rrbase[t] - cx->n_dev_recs + k] = q; }
            }
            /* rounds whose paths were computed through the host (Hirschberg-size problems): put them into HBM too */
            const void *round_ops[LF_MAX_ED_ROUNDS], *round_desc[LF_MAX_ED_ROUNDS]; memset(round_ops, 0, sizeof round_ops); memset(round_desc, 0, sizeof round_desc);
            for (int k = 0; k < cx->n_ed_rounds && k < LF_MAX_ED_ROUNDS; k++) {
                ed_round_t *Rk = &cx->ed_rounds[k];
                if (!Rk->d_ops && Rk->ops && Rk->ops_bytes) {
                    void *dp = lfg_dev_slot(cx->ix->device, LF_DS_RND0 + 2 * k, Rk->ops_bytes + 64);
                    if (!dp) { free(ibase); return LF_ERR_NOMEM; }
                    rc = lfg_upload(cx->ix->device, dp, Rk->ops, Rk->ops_bytes);
                    if (rc != LF_OK) { free(ibase); return rc; }
                    Rk->d_ops = (uint8_t *)dp;
                }
                round_ops[k] = Rk->d_ops; round_desc[k] = Rk->d_desc;
            }
            float ms = 0; uint64_t tbytes = 0;
            tmark(cx, "recipe");
            rc = lfg_render(cx->ix, cx->n_dev_recs, cx->d_dev_recs, cx->n_dev_items, cx->d_dev_items, n_recs, recs, n_items, items, round_ops, round_desc, &cx->rtext, &cx->roffs, &tbytes, &ms,
                            cx->dev_sam ? &cx->rtext_dev : NULL, &cx->rlens);
            tmark(cx, "RENDER");
            if (timing) fprintf(stderr, "[lf] render: %d records, %llu pieces, %.1f MB text, kernels %.1f ms, total %.1f ms\n", n_recs, (unsigned long long)n_items, tbytes / 1e6, ms, now_ms() - t0);
            if (rc != LF_OK) { free(ibase); return rc; }
            st->ms_k_render += ms; st->render_bytes += tbytes; st->render_launches += 1;
            if (!cx->dev_sam) parallel_for(cx, n, g_xc.bind_text);
        }
        free(ibase);
    }
    t1 = now_ms(); st->ms_render += t1 - t0; tstage[4] = t1 - t0; t0 = t1;

    /* ---- E: SAM (score + count here; the text is written by lf_map_batch straight into the output buffer) ---- */
    parallel_for(cx, n, phase_sam_score);
    cx->out_base = NULL;
    if (cx->dev_sam) {
        rc = sam_stage_dev(cx);
        if (rc != LF_OK) return rc;
    } else parallel_for(cx, n, g_xc.sam_print);
    t1 = now_ms(); st->ms_sam += t1 - t0; tstage[5] = t1 - t0;
    tmark(cx, "samcount");
    tmark_dump(cx, t_begin);
    if (timing) fprintf(stderr, "[lf] lane %d chunk of %d reads: seed %.1f vote %.1f chain %.1f extend %.1f render %.1f sam-count %.1f ms (t=%.1f)\n",
                        cx->lane, n, tstage[0], tstage[1], tstage[2], tstage[3], tstage[4], tstage[5], now_ms());
    return LF_OK;
}

void chunk_free(ctx_t *cx)
{
    for (int t = 0; t < cx->n_threads; t++) ar_reset(&cx->arena[t]);      /* every per-read object at once */
    free(cx->creq); free(cx->cseeds); free(cx->chain_idx);
    free(cx->open); cx->open = NULL; cx->n_open = 0;
    if (cx->host_vote) { free(cx->chain_len); free(cx->chain_score); }          /* device path: pinned slots of lfg_vote_chain */
    for (int k = 0; k < cx->n_ed_rounds; k++) { ed_round_t *R = &cx->ed_rounds[k]; if (!R->pinned) { free(R->ed); free(R->end); free(R->ops_len); free(R->ops); } free(R->ops_off); }
    for (int k = 0; k < cx->n_ksw_rounds; k++) { ksw_round_t *R = &cx->ksw_rounds[k]; free(R->score); free(R->qle); free(R->tle); }
    free(cx->ed_rounds); free(cx->ksw_rounds);
    if (cx->stages) {
        for (int t = 0; t < cx->n_threads; t++) {
            stage_t *s = &cx->stages[t];
            free(s->qb); free(s->tb); free(s->qoff); free(s->toff); free(s->mode); free(s->owner);
            free(s->kq); free(s->kt); free(s->kqoff); free(s->ktoff); free(s->kprm); free(s->kowner);
            free(s->dd); free(s->dops); free(s->downer); free(s->ri); free(s->rr);
            free(cx->ed_jobs[t].job); free(cx->ksw_jobs[t].job); free(cx->edd_jobs[t].job);
        }
        free(cx->stages); free(cx->ed_jobs); free(cx->ksw_jobs); free(cx->edd_jobs); cx->ed_jobs = cx->ksw_jobs = cx->edd_jobs = NULL;
    }
    cx->creq = NULL; cx->cseeds = NULL; cx->chain_idx = NULL; cx->chain_len = NULL; cx->chain_score = NULL;
    cx->ed_rounds = NULL; cx->ksw_rounds = NULL; cx->n_ed_rounds = cx->n_ksw_rounds = 0; cx->stages = NULL;
    free(cx->rrbase); cx->rrbase = NULL; cx->rtext = NULL; cx->roffs = NULL;
}

