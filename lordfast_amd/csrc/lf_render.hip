/*
 * lf_render.hip -- CIGAR and MD:Z text of every SAM record, written on the GPU from the alignment paths that the
 * edlib kernels left in HBM.
 *
 * The reference builds both strings base by base while it walks a chain (alignChain_edlib, src/LordFAST.cpp:1570-1763
 * for the per-base tracks and :1799-2230 for the order of the pieces).  Here the host's chain walk only records the
 * ORDER of the pieces as 24-byte items (anchor run, clipped run, edlib path forward / reversed, pure deletion); one
 * wavefront per record then turns the pieces into text:
 *
 *   CIGAR  run-length encoding of M / I / D over the concatenated pieces; a leading or trailing I is printed as S
 *          (src/LordFAST.cpp:1691-1708)
 *   MD     <matches><mismatched reference base> ... ^<deleted reference bases> ...; an insertion breaks a deletion run
 *          (src/LordFAST.cpp:1717-1760)
 *
 * Per 64-element tile the lanes classify their element (register tables), find run starts and match-counter flushes as
 * comparison masks, get the run lengths from the position of the previous set bit and the match counts from a difference of
 * prefix counts (v_mbcnt), and place the tokens of BOTH strings with one DPP prefix sum; only the open run, the match counter
 * and the output cursors are carried from tile to tile.  On the product path the kernel runs ONCE, into per-record regions
 * sized by an upper bound (lf_render_caps_kernel -> one scan); the count + write form remains for packed text.  The edit
 * paths never leave the device, and they arrive with their mismatches marked (the traceback kernels classify diagonal moves).
 *
 * Traffic per record: ops bytes read once, text written once, a few 2-bit reference bases.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include "lf_internal.h"
#include "lf_gpu_common.h"
#include "lf_scan.h"

struct lf_rrounds { const uint8_t *ops[LF_MAX_ED_ROUNDS]; };

/* decimal digits of v followed by up to two characters (any v: the scalar path and the rare long runs) */
__device__ __forceinline__ int lf_ndigits(uint32_t v)
{
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6
         : v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
__device__ __forceinline__ void lf_put_token(char *dst, uint32_t v, int nd, char c1, char c2)
{
    for (int k = nd - 1; k >= 0; k--) { dst[k] = (char)('0' + v % 10u); v /= 10u; }
    dst[nd] = c1;
    if (c2) dst[nd + 1] = c2;
}
/* the tile path: run lengths and match counts below 10 000 (a wave-uniform test sends anything longer to the functions above).
 * nd = 1 .. 4 without a branch; the token is built in a register -- v / 10 as a 24-bit multiply and a shift (exact below
 * 81 920) -- and leaves as byte stores under one comparison each. */
__device__ __forceinline__ int lf_ndigits4(uint32_t v) { return 1 + (int)(v >= 10u) + (int)(v >= 100u) + (int)(v >= 1000u); }
__device__ __forceinline__ void lf_put_token4(char *dst, uint32_t v, int nd /* 0 .. 4 */, uint32_t c1, uint32_t c2 /* 0: none */)
{
    const uint32_t q1 = __umul24(v, 52429u) >> 19, q2 = __umul24(q1, 52429u) >> 19, q3 = __umul24(q2, 52429u) >> 19;
    const uint32_t d3 = v - 10u * q1, d2 = q1 - 10u * q2, d1 = q2 - 10u * q3;      /* q3 = thousands */
    const uint64_t dig = (uint64_t)(0x30303030u + (q3 | (d1 << 8) | (d2 << 16) | (d3 << 24))) >> ((4 - nd) << 3);      /* first digit in byte 0; nd 0: none */
    const uint64_t tok = dig | ((uint64_t)(c1 | (c2 << 8)) << (nd << 3));
    const int len = nd + 1 + (c2 ? 1 : 0);
    const uint32_t lo = (uint32_t)tok, hi = (uint32_t)(tok >> 32);
    dst[0] = (char)lo;
    if (len > 1) dst[1] = (char)(lo >> 8);
    if (len > 2) dst[2] = (char)(lo >> 16);
    if (len > 3) dst[3] = (char)(lo >> 24);
    if (len > 4) dst[4] = (char)hi;
    if (len > 5) dst[5] = (char)(hi >> 8);
}
/* DPP form (VALU, no LDS crossbar) of the inclusive prefix sum over the wavefront */
__device__ __forceinline__ uint32_t lf_wave_incl_sum(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      /* row_shr:1 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      /* row_shr:2 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);      /* row_shr:4 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);      /* row_shr:8 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     /* row_bcast:15 -> rows 1, 3 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     /* row_bcast:31 -> rows 2, 3 */
    return (uint32_t)x;
}
/* set bits of m below this lane */
__device__ __forceinline__ uint32_t lf_mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ char lf_pac_char(const uint8_t *pac, uint32_t pos, int comp)
{
    int b = (pac[pos >> 2] >> ((~pos & 3u) << 1)) & 3;
    if (comp) b = 3 - b;
    return (char)(0x54474341u >> (b << 3));          /* "ACGT"[b] without a table load */
}

/* MODE 0: counting pass (exact sizes)   1: writing pass behind it (packed text: the host-side consumers)
 *      2: SINGLE pass -- writes into per-record regions sized by an upper bound (lf_render_caps_kernel) and reports the exact
 *         lengths; the SAM writer copies the text out of the regions anyway, so nothing is gained by packing it first.
 * Every path arrives with its mismatches marked (op 3): the edlib traceback kernels classify the diagonal moves themselves.
 *
 * Two levels.  85 % of a record's elements are matches, which print nothing by themselves: a 64-op tile of a path is only
 * CLASSIFIED -- its mismatches, insertions and deletions become EVENTS (type, the number of matches in front of it, the
 * reference base of a mismatch / deletion) in an LDS queue, anchor runs just add to the count of pending matches -- and the
 * expensive part (run lengths, match counts, digits, token placement, stores) runs once per 64 EVENTS, i.e. once per ~430
 * elements.  An event stands for two segments of the CIGAR's letter stream: 'M' x eq (if eq > 0), then its own letter x len
 * (mismatch: 'M' x 1).  Run lengths follow R_i = carry_i * R_(i-1) + add_i -- an affine recurrence, composed over the
 * wavefront by one DPP scan of (carry, add) pairs; match counts are differences of a prefix sum of eq. */
#define LF_RM_COUNT  0
#define LF_RM_WRITE  1
#define LF_RM_SINGLE 2
#define EV_X 0
#define EV_I 1
#define EV_D 2
/* inclusive scan of affine maps x -> c x + a (c in {0, 1}) over the wavefront: lane i ends with the composition of lanes 0 .. i */
__device__ __forceinline__ void lf_wave_affine_scan(uint32_t &c, uint32_t &a)
{
#define LF_AFF_STEP(CTRL, RM) { \
        const uint32_t tc = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)c, CTRL, RM, 0xf, false); \
        const uint32_t ta = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, CTRL, RM, 0xf, false); \
        a += c ? ta : 0u; c &= tc; }
    LF_AFF_STEP(0x111, 0xf) LF_AFF_STEP(0x112, 0xf) LF_AFF_STEP(0x114, 0xf) LF_AFF_STEP(0x118, 0xf)
    LF_AFF_STEP(0x142, 0xa) LF_AFF_STEP(0x143, 0xc)
#undef LF_AFF_STEP
}
template <int MODE>
__global__ void __launch_bounds__(64)
lf_render_kernel(const lf_rrecord_t *__restrict__ recs, int n_recs, const lf_ritem_t *__restrict__ items, lf_rrounds R,
                 const uint8_t *__restrict__ pac, uint32_t *__restrict__ lens /* 2 per record */,
                 const uint64_t *__restrict__ offs /* 2 per record (WRITE) */, char *__restrict__ text)
{
    constexpr bool WRITE = MODE != LF_RM_COUNT;
    __shared__ uint2 s_ev[128];                      /* the event queue: x = matches in front, y = len << 10 | base << 2 | type */
    const int rec = blockIdx.x;
    if (rec >= n_recs) return;
    const int lane = threadIdx.x;
    const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const lf_rrecord_t rr = recs[rec];
    char *cg = WRITE ? text + offs[2 * rec] : nullptr, *md = WRITE ? text + offs[2 * rec + 1] : nullptr;
    /* carried state (wave-uniform) */
    int c_ch = 0; uint32_t c_run = 0; int c_first = 1; uint32_t c_out = 0;
    uint32_t m_num = 0; int m_last = -1 /* type of the last event, -1: matches (or nothing) */; uint32_t m_out = 0;
    uint32_t pending = 0;                            /* matches seen since the last event */
    int n_ev = 0;                                    /* events waiting in s_ev */

    /* close the open CIGAR run and open a new one (wave-uniform path, lane 0 writes) */
    auto cigar_run = [&](int ch, uint32_t n) {
        if (ch == c_ch) { c_run += n; return; }
        if (c_ch) {
            const int nd = lf_ndigits(c_run);
            if (WRITE && lane == 0) lf_put_token(cg + c_out, c_run, nd, (c_first && c_ch == 'I') ? 'S' : (char)c_ch, 0);
            c_out += nd + 1; c_first = 0;
        }
        c_ch = ch; c_run = n;
    };

    /* ---- the first `cnt` events of the queue -> text ---- */
    auto flush_events = [&](int cnt) {
        const bool act = lane < cnt;
        const uint2 ev = s_ev[lane];
        const uint32_t eq = act ? ev.x : 0u, ty = ev.y & 3u, len = act ? ev.y >> 10 : 0u, bch = (ev.y >> 2) & 0xffu;
        const bool hasA = eq > 0;
        const int L = !act ? 0 : ty == EV_I ? 'I' : ty == EV_D ? 'D' : 'M';
        /* ---- CIGAR ---- */
        const int pL0 = __builtin_amdgcn_update_dpp(c_ch, L, 0x138, 0xf, 0xf, false);        /* letter of the previous event (lane 0: the open run's) */
        const int pL = pL0;
        const int lbB = hasA ? 'M' : pL;                                                       /* letter in front of the event's own segment */
        const bool tokA = hasA && pL != 'M' && pL != 0;                                        /* the matches close the previous run */
        const bool tokB = act && L != lbB && lbB != 0;                                         /* the event closes the run in front of it */
        /* length of the run that is open after the event: R_i = carry_i R_(i-1) + add_i; lanes past the last event pass R on */
        uint32_t cr = !act ? 1u : hasA ? (uint32_t)(pL == 'M' && L == 'M') : (uint32_t)(L == pL);
        uint32_t ad = hasA && L == 'M' ? eq + len : len;
        lf_wave_affine_scan(cr, ad);
        const uint32_t Rn = (cr ? c_run : 0u) + ad;
        const uint32_t Rp = (uint32_t)__builtin_amdgcn_update_dpp((int)c_run, (int)Rn, 0x138, 0xf, 0xf, false);      /* R_(i-1) */
        const uint32_t valA = Rp;
        const uint32_t valB = hasA ? (pL == 'M' ? Rp : 0u) + eq : Rp;
        /* ---- MD ---- */
        const int pty0 = __builtin_amdgcn_update_dpp(m_last, act ? (int)ty : -1, 0x138, 0xf, 0xf, false);      /* (every lane executes the move: a lane that sat out would not be a source) */
        const int pty = hasA ? -1 : pty0;
        const bool isx = act && ty == EV_X, isd = act && ty == EV_D, isxd = isx || isd;
        const bool flush = isx || (isd && pty != EV_D);
        const uint64_t fmask = lf_ballot(flush);
        const uint32_t E = lf_wave_incl_sum(eq);                                               /* matches up to and including the event's own */
        const uint64_t fb = fmask & below;
        const int p = fb ? 63 - __clzll((long long)fb) : 0;
        const uint32_t Ep = (uint32_t)__shfl((int)E, p);
        const uint32_t num = fb ? E - Ep : m_num + E;
        /* ---- digits, placement ---- */
        const bool big = lf_any((tokA && valA >= 10000u) || (tokB && valB >= 10000u) || (flush && num >= 10000u));
        int ndA = lf_ndigits4(valA), ndB = lf_ndigits4(valB), mnd = flush ? lf_ndigits4(num) : 0;
        if (big) { ndA = lf_ndigits(valA); ndB = lf_ndigits(valB); mnd = flush ? lf_ndigits(num) : 0; }
        const uint32_t lenA = tokA ? (uint32_t)ndA + 1u : 0u, lenB = tokB ? (uint32_t)ndB + 1u : 0u;
        const uint32_t mlen = isxd ? (uint32_t)mnd + (flush && isd ? 2u : 1u) : 0u;
        const uint32_t packed = (lenA + lenB) | (mlen << 16), incl = lf_wave_incl_sum(packed), excl = incl - packed;
        const uint32_t both = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t pos = excl & 0xffffu, mpos = excl >> 16, tot = both & 0xffffu, mtot = both >> 16;
        const uint64_t tmask = lf_ballot(tokA || tokB);
        if (WRITE) {
            uint32_t letA = (uint32_t)pL, letB = (uint32_t)lbB;
            if (c_first && tmask && !(tmask & below) && (tokA || tokB)) {                      /* the record's first token: I prints as S */
                if (tokA) { if (letA == 'I') letA = 'S'; } else if (letB == 'I') letB = 'S';
            }
            const uint32_t c1 = (flush && isd) ? (uint32_t)'^' : bch, c2 = (flush && isd) ? bch : 0u;
            if (!big) {
                if (tokA) lf_put_token4(cg + c_out + pos, valA, ndA, letA, 0);
                if (tokB) lf_put_token4(cg + c_out + pos + lenA, valB, ndB, letB, 0);
                if (isxd) lf_put_token4(md + m_out + mpos, num, mnd, c1, c2);
            } else {
                if (tokA) lf_put_token(cg + c_out + pos, valA, ndA, (char)letA, 0);
                if (tokB) lf_put_token(cg + c_out + pos + lenA, valB, ndB, (char)letB, 0);
                if (isxd) lf_put_token(md + m_out + mpos, num, mnd, (char)c1, (char)c2);
            }
        }
        /* carried state: the run open after the last event, the matches since the last flush, the last event's type */
        const int last = cnt - 1;
        c_run = (uint32_t)__builtin_amdgcn_readlane((int)Rn, last);
        c_ch = __builtin_amdgcn_readlane(L, last);
        c_out += tot;
        if (tmask) c_first = 0;
        m_out += mtot;
        const uint32_t Etot = (uint32_t)__builtin_amdgcn_readlane((int)E, 63);
        if (fmask) { const int pf = 63 - __clzll((long long)fmask); m_num = Etot - (uint32_t)__builtin_amdgcn_readlane((int)E, pf); }
        else m_num += Etot;
        m_last = (int)(__builtin_amdgcn_readlane((int)ev.y, last) & 3);
    };
    /* keep at most 63 events waiting */
    auto drain = [&]() {
        while (n_ev >= 64) {
            flush_events(64);
            const uint2 t = s_ev[64 + lane];                           /* (one wavefront: DS operations execute in order) */
            s_ev[lane] = t;
            n_ev -= 64;
        }
    };

    /* The items of a record form a chain of dependent loads (item -> its ops): the next item and the first tile of its ops are
     * fetched while the current item is processed, and inside a path the next tile's ops while the current tile is classified
     * (a record has ~240 items of ~100 ops). */
    auto is_path = [&](const lf_ritem_t &X) -> bool { return X.n != 0 && X.kind >= LF_RI_OPS_FWD && X.kind <= LF_RI_OPS_REV; };
    auto tile_op = [&](const lf_ritem_t &X, uint32_t base) -> uint32_t {
        const uint32_t k = base + (uint32_t)lane;
        if (k >= X.n) return 0;
        return (R.ops[X.round] + X.ops_begin)[X.kind == LF_RI_OPS_REV ? X.n - 1 - k : k];
    };
    lf_ritem_t Inext; memset(&Inext, 0, sizeof Inext);
    if (rr.nitems) Inext = items[rr.item0];
    uint32_t op_next = is_path(Inext) ? tile_op(Inext, 0) : 0;
    for (uint32_t it = 0; it < rr.nitems; it++) {
        const lf_ritem_t I = Inext;
        uint32_t op_cur = op_next;
        if (it + 1 < rr.nitems) { Inext = items[rr.item0 + it + 1]; op_next = is_path(Inext) ? tile_op(Inext, 0) : 0; }
        if (I.n == 0) continue;
        if (I.kind == LF_RI_RUN_M) { pending += I.n; continue; }
        if (I.kind == LF_RI_RUN_I) {
            if (lane == 0) s_ev[n_ev] = make_uint2(pending, (I.n << 10) | EV_I);
            n_ev++; pending = 0;
            drain();
            continue;
        }
        if (I.kind == LF_RI_DEL) {
            for (uint32_t j0 = 0; j0 < I.n; j0 += 64) {
                const uint32_t j = j0 + (uint32_t)lane, c = I.n - j0 < 64u ? I.n - j0 : 64u;
                if (j < I.n) s_ev[n_ev + lane] = make_uint2(j == 0 ? pending : 0u, (1u << 10) | ((uint32_t)(uint8_t)lf_pac_char(pac, I.tpos + j, 0) << 2) | EV_D);
                n_ev += (int)c; pending = 0;
                drain();
            }
            continue;
        }
        /* an edit path: I.n ops, forward or reversed */
        const int trc = (I.kind == LF_RI_OPS_FWD_TRC);
        uint32_t tcarry = 0;
        for (uint32_t base = 0; base < I.n; base += 64) {
            const uint32_t op = op_cur;
            if (base + 64 < I.n) op_cur = tile_op(I, base + 64);
            const uint32_t cnt = (I.n - base < 64u) ? I.n - base : 64u;
            const bool act = (uint32_t)lane < cnt;
            /* op 0 = 1 I 2 D 3 X  ->  event type (X 0, I 1, D 2); op 0 is no event */
            const bool isev = act && op != 0u;
            const uint64_t nz = lf_ballot(isev), nimask = lf_ballot(act && op != 1u);
            if (isev) {
                const uint32_t ety = op == 3u ? EV_X : op == 1u ? EV_I : EV_D;
                uint32_t bch = 0;
                if (op != 1u) { const uint32_t ti = tcarry + lf_mbcnt(nimask); bch = (uint32_t)(uint8_t)lf_pac_char(pac, trc ? I.tpos - ti : I.tpos + ti, trc); }
                const uint64_t nb = nz & below;
                const uint32_t eq = nb ? (uint32_t)(lane - 1 - (63 - __clzll((long long)nb))) : pending + (uint32_t)lane;      /* everything between two events is a match */
                s_ev[n_ev + (int)lf_mbcnt(nz)] = make_uint2(eq, (1u << 10) | (bch << 2) | ety);
            }
            if (nz) { pending = cnt - 1u - (uint32_t)(63 - __clzll((long long)nz)); n_ev += __popcll(nz); } else pending += cnt;
            tcarry += (uint32_t)__popcll(nimask);
            drain();
        }
    }
    if (n_ev) { flush_events(n_ev); n_ev = 0; }
    /* the matches behind the last event, then close the record (src/LordFAST.cpp:1704-1708, :1756-1760): the last CIGAR run
     * (I -> S), the match counter */
    if (pending) { cigar_run('M', pending); m_num += pending; }
    if (c_ch) {
        const int nd = lf_ndigits(c_run);
        if (WRITE && lane == 0) lf_put_token(cg + c_out, c_run, nd, c_ch == 'I' ? 'S' : (char)c_ch, 0);
        c_out += nd + 1;
    }
    {
        const int nd = lf_ndigits(m_num);
        if (WRITE && lane == 0) { lf_put_token(md + m_out, m_num, nd, 0, 0); cg[c_out] = 0; }
        m_out += nd;
    }
    if (MODE != LF_RM_WRITE && lane == 0) { lens[2 * rec] = c_out + 1; lens[2 * rec + 1] = m_out + 1; }      /* + NUL */
}

/* upper bounds of a record's CIGAR and MD text (single-pass mode).  A run of L ops prints digits(L) + 1 <= 2 L characters;
 * an MD token is digits(matches before it) + 2 characters at most and every match is counted by one token only:
 * CIGAR <= 2 ops + 11 per item, MD <= 3 ops + 11 per item + the deleted bases of the pure-deletion items. */
__global__ void lf_render_caps_kernel(const lf_rrecord_t *__restrict__ recs, int n_recs, const lf_ritem_t *__restrict__ items, uint32_t *__restrict__ caps /* 2 per record */)
{
    const int rec = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    if (rec >= n_recs) return;
    const lf_rrecord_t rr = recs[rec];
    uint32_t cg = 0, md = 0;
    for (uint32_t it = lane; it < rr.nitems; it += 64) {
        const lf_ritem_t I = items[rr.item0 + it];
        cg += 11; md += 11;
        if (I.kind >= LF_RI_OPS_FWD && I.kind <= LF_RI_OPS_REV) { cg += 2 * I.n; md += 3 * I.n; }
        else if (I.kind == LF_RI_DEL) md += I.n;
    }
    for (int o = 32; o > 0; o >>= 1) { cg += __shfl_xor(cg, o); md += __shfl_xor(md, o); }
    if (lane == 0) { caps[2 * rec] = (cg + 16 + 7) & ~7u; caps[2 * rec + 1] = (md + 16 + 7) & ~7u; }
}

struct lf_widen32 { __host__ __device__ uint64_t operator()(uint32_t v) const { return v; } };

/* host entry: items / records in (pinned) host memory; rounds[r] = device address of round r's ops (or NULL);
 * text comes back in a pinned slot, offs[2*rec], offs[2*rec+1] = start of the record's CIGAR / MD (NUL-terminated). */
extern "C" int lfg_render(const struct lf_index *ix, int n_dev_recs, const void *d_recs_dev, uint64_t n_dev_items, const void *d_items_dev,
                          int n_host_recs, const lf_rrecord_t *recs, uint64_t n_host_items, const lf_ritem_t *items,
                          const void *const *round_ops, const void *const *round_desc, char **text_out, uint64_t **offs_out, uint64_t *text_bytes, float *ms,
                          lfg_rtext_t *dev_text, uint32_t **lens_out)
{
    if (ms) *ms = 0;
    *text_out = nullptr; *offs_out = nullptr; *text_bytes = 0;
    const int n_recs = n_dev_recs + n_host_recs;
    const uint64_t n_items = n_dev_items + n_host_items;
    if (n_recs == 0) return LF_OK;
    lf_dev_state *st = (lf_dev_state *)ix->dev;
    if (!st) { lf_set_error("index is not on a device"); return LF_ERR_NO_DEVICE; }
    const int device = ix->device;
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)lfg_lane_stream(device, 1);
    if (!s) return LF_ERR_HIP;
    /* the device-planned recipe (lf_walk_emit_kernel) is used in place; host-planned records (the rare chains) are appended
     * behind a copy of it */
    lf_rrecord_t *d_recs = (lf_rrecord_t *)d_recs_dev;
    lf_ritem_t *d_items = (lf_ritem_t *)d_items_dev;
    if (n_host_recs > 0 || !d_recs) {
        d_recs = (lf_rrecord_t *)lfg_dev_slot(device, LF_DS_RENDER0 + 0, (size_t)n_recs * sizeof(lf_rrecord_t));
        d_items = (lf_ritem_t *)lfg_dev_slot(device, LF_DS_RENDER0 + 1, (size_t)n_items * sizeof(lf_ritem_t) + 64);
        if (!d_recs || !d_items) return LF_ERR_NOMEM;
        if (n_dev_recs) {
            HIPCHK(hipMemcpyAsync(d_recs, d_recs_dev, (size_t)n_dev_recs * sizeof(lf_rrecord_t), hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(d_items, d_items_dev, (size_t)n_dev_items * sizeof(lf_ritem_t), hipMemcpyDeviceToDevice, s));
        }
    }
    uint32_t *d_lens = (uint32_t *)lfg_dev_slot(device, LF_DS_RENDER0 + 2, (size_t)n_recs * 8);
    uint64_t *d_offs = (uint64_t *)lfg_dev_slot(device, LF_DS_RENDER0 + 3, (size_t)n_recs * 16 + 16);
    uint64_t *h_offs = (uint64_t *)lfg_pin_slot(LF_PS_RENDER0 + 0, (size_t)n_recs * 16 + 16);
    uint32_t *h_tail = (uint32_t *)lfg_pin_slot(LF_PS_RENDER0 + 1, 64);
    if (!d_recs || !d_items || !d_lens || !d_offs || !h_offs || !h_tail) return LF_ERR_NOMEM;
    lf_rrounds R;
    for (int r = 0; r < LF_MAX_ED_ROUNDS; r++) { R.ops[r] = (const uint8_t *)round_ops[r]; }
    (void)round_desc;
    hipEvent_t e0 = (hipEvent_t)lfg_lane_event(device, 32), e1 = (hipEvent_t)lfg_lane_event(device, 33);
    if (!e0 || !e1) return LF_ERR_HIP;
    if (n_host_recs) {       /* the caller has rebased the host records' item0 by n_dev_items */
        HIPCHK(hipMemcpyAsync(d_recs + n_dev_recs, recs, (size_t)n_host_recs * sizeof(lf_rrecord_t), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_items + n_dev_items, items, (size_t)n_host_items * sizeof(lf_ritem_t), hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipEventRecord(e0, s));
    const bool single = dev_text != nullptr;      /* the two-pass form stays for the host-side consumers (packed text) */
    uint32_t *d_caps = single ? (uint32_t *)lfg_dev_slot(device, LF_DS_RENDER0 + 6, (size_t)n_recs * 8) : d_lens;
    if (!d_caps) return LF_ERR_NOMEM;
    if (single) hipLaunchKernelGGL(lf_render_caps_kernel, dim3((unsigned)((n_recs + 3) / 4)), dim3(256), 0, s, d_recs, n_recs, d_items, d_caps);
    else hipLaunchKernelGGL(lf_render_kernel<LF_RM_COUNT>, dim3((unsigned)n_recs), dim3(64), 0, s, d_recs, n_recs, d_items, R, st->view.pac,
                            d_lens, (const uint64_t *)nullptr, (char *)nullptr);
    { lf_scan_u32 f; f.p = d_caps; const int src = lf_scan_excl(device, 5, s, f, d_offs, 2 * (size_t)n_recs); if (src != LF_OK) return src; }
    HIPCHK(hipMemcpyAsync(h_offs, d_offs, (size_t)n_recs * 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_tail, d_caps + 2 * (size_t)n_recs - 1, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const uint64_t total = h_offs[2 * (size_t)n_recs - 1] + h_tail[0];
    char *d_text = (char *)lfg_dev_slot(device, LF_DS_RENDER0 + 5, total + 64);
    char *h_text = dev_text ? nullptr : (char *)lfg_pin_slot(LF_PS_RENDER0 + 2, total + 64);
    uint32_t *h_lens = dev_text ? (uint32_t *)lfg_pin_slot(LF_PS_RENDER1 + 0, (size_t)n_recs * 8 + 16) : nullptr;      /* never the slot of an input (the caller's items live in LF_PS_RENDER0 + 3) */
    if (!d_text || (!dev_text && !h_text) || (dev_text && !h_lens)) return LF_ERR_NOMEM;
    if (single) hipLaunchKernelGGL(lf_render_kernel<LF_RM_SINGLE>, dim3((unsigned)n_recs), dim3(64), 0, s, d_recs, n_recs, d_items, R, st->view.pac,
                                   d_lens, (const uint64_t *)d_offs, d_text);
    else hipLaunchKernelGGL(lf_render_kernel<LF_RM_WRITE>, dim3((unsigned)n_recs), dim3(64), 0, s, d_recs, n_recs, d_items, R, st->view.pac,
                            d_lens, (const uint64_t *)d_offs, d_text);
    HIPCHK(hipEventRecord(e1, s));
    if (dev_text) HIPCHK(hipMemcpyAsync(h_lens, d_lens, (size_t)n_recs * 8, hipMemcpyDeviceToHost, s));      /* SA:Z strings need a few CIGARs (lfg_fetch) */
    else HIPCHK(hipMemcpyAsync(h_text, d_text, total, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (ms) HIPCHK(hipEventElapsedTime(ms, e0, e1));
    uint64_t used = total;
    if (single) { used = 0; for (size_t i = 0; i < 2 * (size_t)n_recs; i++) used += h_lens[i]; }      /* the regions are upper bounds: report the text itself */
    *text_out = h_text; *offs_out = h_offs; *text_bytes = used;
    if (dev_text) { dev_text->d_text = d_text; dev_text->d_offs = d_offs; dev_text->d_lens = d_lens; }
    if (lens_out) *lens_out = h_lens;
    return LF_OK;
}

/* small synchronous device -> host copy (rare per-base fallback of the chain walk) */
extern "C" int lfg_fetch(int device, void *dst, const void *src_dev, size_t bytes)
{
    /* called from pool workers: own non-blocking stream, so that the copy does not serialise with the lanes' kernels */
    static std::mutex mu; static hipStream_t fs[16];
    if (device < 0 || device >= 16) return LF_ERR_ARG;
    std::lock_guard<std::mutex> g(mu);
    HIPCHK(hipSetDevice(device));
    if (!fs[device]) HIPCHK(hipStreamCreateWithFlags(&fs[device], hipStreamNonBlocking));
    HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, fs[device]));
    HIPCHK(hipStreamSynchronize(fs[device]));
    return LF_OK;
}
/* many small pieces of device memory -> one pinned host buffer: ONE gather kernel, ONE copy, one wait (the bases of the reads
 * whose chains the host replays).  lfg_fetch per read -- one lock, one stream, one wait each, into pageable memory -- took 30 ms
 * of a 6 250-read chunk of the C4 workload; one asynchronous copy per read on the lane's stream still took 8 - 30 ms, each of the
 * 500 copies a blit kernel queued behind the other lanes' kernels. */
__global__ void __launch_bounds__(256)
lf_gather_bytes_kernel(int n, const uint64_t *__restrict__ meta /* n source addresses, n destination offsets, n lengths */, uint8_t *__restrict__ dst)
{
    const int j = blockIdx.x;
    if (j >= n) return;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(meta[j]);
    uint8_t *d = dst + meta[n + j];
    const uint64_t len = meta[2 * (size_t)n + j];
    for (uint64_t i = threadIdx.x; i < len; i += 256) d[i] = src[i];
}
extern "C" int lfg_fetch_gather(int device, int n, void *host_base, const uint64_t *host_off, const void *const *src_dev, const size_t *bytes, uint64_t total)
{
    if (n <= 0) return LF_OK;
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)lfg_lane_stream(device, 1);
    if (!s) return LF_ERR_HIP;
    uint64_t *h_meta = (uint64_t *)lfg_pin_slot(LF_PS_FETCH_META, (size_t)n * 24 + 64);
    uint64_t *d_meta = (uint64_t *)lfg_dev_slot(device, LF_DS_WALK0 + 15, (size_t)n * 24 + 64);
    uint8_t *d_stage = (uint8_t *)lfg_dev_slot(device, LF_DS_WALK0 + 16, total + 64);
    if (!h_meta || !d_meta || !d_stage) return LF_ERR_NOMEM;
    for (int i = 0; i < n; i++) { h_meta[i] = (uint64_t)(uintptr_t)src_dev[i]; h_meta[n + i] = host_off[i]; h_meta[2 * (size_t)n + i] = bytes[i]; }
    HIPCHK(hipMemcpyAsync(d_meta, h_meta, (size_t)n * 24, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lf_gather_bytes_kernel, dim3((unsigned)n), dim3(256), 0, s, n, (const uint64_t *)d_meta, d_stage);
    HIPCHK(hipMemcpyAsync(host_base, d_stage, total, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return LF_OK;
}
extern "C" int lfg_upload(int device, void *dst_dev, const void *src, size_t bytes)
{
    HIPCHK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)lfg_lane_stream(device, 1);
    if (!s) return LF_ERR_HIP;
    HIPCHK(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return LF_OK;
}
