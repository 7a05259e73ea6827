/*
 * hband_model.cpp -- TEST INFRASTRUCTURE.  A lane-by-lane host model of the BANDED Hirschberg sweep of
 * lordfast_amd/csrc/lf_hband.hip (same schedule: block b on lane b mod (L W), 16-step groups, lanes that switch blocks at group
 * boundaries, carries by "DPP" inside a wavefront and through a ring of words between wavefronts), with the lanes of a
 * wavefront as arrays.  It exists so that the band geometry, the lane schedule and the score bookkeeping can be checked on
 * the CPU against a plain full-matrix DP (tests/test_hband_model.py) -- with L = 64 as on the device, and with small L so
 * that the wrap-around of the lanes is reached by small problems.  Nothing in the product includes or links this file.
 *
 * What is modelled (reference: lib/edlib/edlib.cpp:1161-1330 obtainAlignmentHirschberg; :134-153, :484-566 the band):
 *   a node (q[0..n), t[0..m), best)  ->  split row, left / right score       (NW node, distance known or a trial bound k0)
 *   an SHW root (q, t, k0)           ->  distance, end column, or "k0 too small"
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

namespace {

const int INF = 1 << 28;

struct Myers { uint64_t Pv, Mv; };
/* lf_myers_step (lf_edlib_common.h): hin / hout as two bits, bit 0 = +1, bit 1 = -1 */
inline uint32_t myers_step(uint64_t &Pv, uint64_t &Mv, uint64_t Eq, uint32_t hin, uint64_t &ph_out, uint64_t &mh_out)
{
    const uint64_t hpos = hin & 1u, hneg = hin >> 1;
    const uint64_t Xv = Eq | Mv;
    Eq |= hneg;
    const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    uint64_t Ph = Mv | ~(Xh | Pv);
    uint64_t Mh = Pv & Xh;
    ph_out = Ph; mh_out = Mh;
    const uint32_t hout = (uint32_t)(Ph >> 63) | ((uint32_t)(Mh >> 63) << 1);
    Ph = (Ph << 1) | hpos;
    Mh = (Mh << 1) | hneg;
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    return hout;
}
inline int delta2(uint32_t h) { return (int)(h & 1u) - (int)(h >> 1); }

/* band of a node: diagonals d = j - i (0-based cell i, j) in [dlo, dhi]; the same interval serves both halves (the reversed
 * half's diagonals are (m - n) - d).  k = distance (or trial bound) of the WHOLE node (n x m). */
struct Band { int dlo, dhi; };
inline Band nw_band(int n, int m, int k)
{
    const int md = m - n, amd = md < 0 ? -md : md;
    int e = (k - amd) / 2; if (e < 0) e = 0;
    Band B; B.dlo = std::min(0, md) - e; B.dhi = std::max(0, md) + e;
    if (B.dhi < 1) B.dhi = 1;                 /* a block must still be inside the band where the block below starts */
    return B;
}
inline Band shw_band(int k) { Band B; B.dlo = -k; B.dhi = k < 1 ? 1 : k; return B; }

struct HalfOut {
    std::vector<uint64_t> pv, mv; std::vector<int> base; std::vector<char> valid;   /* per block: last column */
    int shw_best, shw_c;
    long steps, active_steps;
};

/* width limit of W wavefronts of L lanes (lane reuse: a lane must be done with block b before block b + L W starts) */
inline int lagw(int L, int W) { return W == 1 ? L : L + 32; }
/* (lf_hband_fits, lf_hirsch.h: the band inside the half's matrix of n rows and mm columns) */
inline bool band_fits(Band B, int n, int mm, int L, int W)
{
    const int nbk = (n + 63) >> 6;
    const int dhi = std::min(B.dhi, mm - 64), dlo = std::max(B.dlo, -64 * (nbk - 1));
    return dhi - dlo <= 64 * L * W + lagw(L, W) * W - 94;
}

/* one half: q codes (0..3, or -1 = never matches) for rows, t codes for columns; mm columns */
void sweep_half(const int8_t *q, int n, const int8_t *t, int mm, Band B, bool track, int L, int W, HalfOut &O)
{
    const int nbk = (n + 63) >> 6, lastb = (n - 1) >> 6, lastbit = (n - 1) & 63;
    O.pv.assign(nbk, 0); O.mv.assign(nbk, 0); O.base.assign(nbk, 0); O.valid.assign(nbk, 0);
    O.shw_best = (n & 63) ? n : 0x7fffffff; O.shw_c = 0; O.steps = 0; O.active_steps = 0;
    if (mm <= 0) return;
    const int LAGW = lagw(L, W), NL = L * W;
    auto skew = [&](int b) { return (b % L) + LAGW * (b / L); };
    auto jlo = [&](int b) { return std::max(0, 64 * b + B.dlo); };
    auto jhi = [&](int b) { return std::min(mm - 1, 64 * b + 63 + B.dhi); };
    int nbA = (mm - 1 - B.dlo) / 64 + 1; if (nbA > nbk) nbA = nbk;      /* blocks that ever enter the band (dlo <= 0) */
    /* per-lane state */
    std::vector<int> b(NL, -1), nbnext(NL), skw(NL, 0), sc(NL, 0), score(NL, 0), best(NL, 0x7fffffff), bestc(NL, 0);
    std::vector<uint32_t> jend(NL, 0), jtop(NL, 0), hout(NL, 1), acc(NL, 0);
    std::vector<uint64_t> Pv(NL, ~0ull), Mv(NL, 0);
    std::vector<std::vector<uint64_t> > peq(NL, std::vector<uint64_t>(4, 0));
    for (int g = 0; g < NL; g++) nbnext[g] = g;
    std::vector<std::vector<uint32_t> > ring_cw(W, std::vector<uint32_t>(8, 0));
    std::vector<std::vector<int> > ring_sc(W, std::vector<int>(8, 0));
    auto finish = [&](int g) {
        if (b[g] >= 0 && jend[g] == (uint32_t)mm) {
            const int bb = b[g];
            O.pv[bb] = Pv[g]; O.mv[bb] = Mv[g];
            O.base[bb] = sc[g] - (__builtin_popcountll(Pv[g]) - __builtin_popcountll(Mv[g]));
            O.valid[bb] = 1;
        }
        if (track && b[g] == lastb) { O.shw_best = best[g]; O.shw_c = bestc[g]; }
        b[g] = -1; jend[g] = 0;
    };
    const int last_step = nbA > 0 ? jhi(nbA - 1) + skew(nbA - 1) : -1;
    const int nG = last_step / 16 + 1;
    for (int G = 0; G < nG; G++) {
        const int s0 = 16 * G;
        /* --- what every lane sees of its left neighbour at the group's start (taken before anything changes) --- */
        std::vector<int> sc_left(NL); std::vector<uint32_t> h_left(NL), cin16(NL, 0);
        for (int g = 0; g < NL; g++) {
            const int p = g / L, l = g % L;
            if (W == 1) { const int src = (l + L - 1) % L; sc_left[g] = sc[src]; h_left[g] = hout[src]; }
            else if (l > 0) { sc_left[g] = sc[g - 1]; h_left[g] = hout[g - 1]; }
            else {
                const int pw = (p + W - 1) % W;
                const uint32_t a = ring_cw[pw][(G + 8 - 3) & 7], c = ring_cw[pw][(G + 8 - 2) & 7];
                cin16[g] = (a >> 30) | (c << 2);
                sc_left[g] = ring_sc[pw][(G + 8 - 3) & 7]; h_left[g] = cin16[g] & 3u;
            }
        }
        /* --- lanes switch blocks at group boundaries only --- */
        for (int g = 0; g < NL; g++) {
            if (b[g] >= 0 && s0 > (int)jend[g] - 1 + skw[g]) finish(g);
            if (b[g] < 0 && nbnext[g] < nbA && s0 + 16 > jlo(nbnext[g]) + skew(nbnext[g])) {
                const int bb = nbnext[g]; nbnext[g] += NL;
                b[g] = bb; skw[g] = skew(bb); jend[g] = (uint32_t)(jhi(bb) + 1);
                jtop[g] = bb == 0 ? 0u : (uint32_t)std::min(mm, 64 * bb + B.dhi);
                Pv[g] = ~0ull; Mv[g] = 0;
                const int p0 = s0 - skw[g];
                if (s0 > jlo(bb) + skw[g]) abort();                   /* the schedule must never be late */
                sc[g] = p0 <= 0 ? 64 * (bb + 1) : sc_left[g] - delta2(h_left[g]) + 64;
                if (track && bb == lastb) { score[g] = sc[g] - (63 - lastbit); best[g] = (n & 63) ? n : 0x7fffffff; bestc[g] = 0; }
                for (int c = 0; c < 4; c++) {
                    uint64_t e = 0;
                    for (int i = 0; i < 64; i++) { const int r = 64 * bb + i; if (r < n && q[r] == c) e |= 1ull << i; }
                    peq[g][c] = e;
                }
            }
        }
        /* --- 16 steps --- */
        for (int g = 0; g < NL; g++) acc[g] = 0;
        for (int k = 0; k < 16; k++) {
            std::vector<uint32_t> hprev(hout);
            for (int g = 0; g < NL; g++) {
                const int p = g / L, l = g % L; (void)p;
                uint32_t from_left;
                if (W == 1) from_left = hprev[(l + L - 1) % L];
                else if (l > 0) from_left = hprev[g - 1];
                else from_left = (cin16[g] >> (2 * k)) & 3u;
                const int j = s0 - skw[g] + k;
                O.steps++;
                if ((uint32_t)j < jend[g]) {
                    O.active_steps++;
                    const uint64_t Eq = t[j] >= 0 ? peq[g][t[j]] : 0ull;
                    const uint32_t hin = (uint32_t)j < jtop[g] ? from_left : 1u;
                    uint64_t ph, mh;
                    hout[g] = myers_step(Pv[g], Mv[g], Eq, hin, ph, mh);
                    acc[g] |= hout[g] << (2 * k);
                    if (track && b[g] == lastb) {
                        score[g] += (int)((ph >> lastbit) & 1) - (int)((mh >> lastbit) & 1);
                        if (score[g] < best[g]) { best[g] = score[g]; bestc[g] = j + 1; }
                    }
                }
            }
        }
        for (int g = 0; g < NL; g++) sc[g] += __builtin_popcount(acc[g] & 0x55555555u) - __builtin_popcount(acc[g] & 0xAAAAAAAAu);
        for (int p = 0; p < W; p++) { const int g = p * L + L - 1; ring_cw[p][G & 7] = acc[g]; ring_sc[p][G & 7] = sc[g]; }
    }
    for (int g = 0; g < NL; g++) if (b[g] >= 0) finish(g);
}

int col_at(const HalfOut &O, int x, int zero, int n)
{
    if (x == 0) return zero;
    const int bb = (x - 1) >> 6, bit = (x - 1) & 63;
    if (bb >= (int)O.valid.size() || !O.valid[bb]) return INF;
    const uint64_t msk = bit == 63 ? ~0ull : ((2ull << bit) - 1);
    (void)n;
    return O.base[bb] + __builtin_popcountll(O.pv[bb] & msk) - __builtin_popcountll(O.mv[bb] & msk);
}

/* plain DP: last column of dist(q[0..r), t[0..mm)) for r = 0..n */
void dp_last_col(const int8_t *q, int n, const int8_t *t, int mm, std::vector<int> &col)
{
    col.resize(n + 1);
    for (int r = 0; r <= n; r++) col[r] = r;
    for (int j = 0; j < mm; j++) {
        int diag = col[0]; col[0] = j + 1;
        for (int r = 1; r <= n; r++) {
            const int up = col[r - 1] + 1, left = col[r] + 1, dg = diag + ((q[r - 1] >= 0 && q[r - 1] == t[j]) ? 0 : 1);
            diag = col[r];
            col[r] = std::min(std::min(up, left), dg);
        }
    }
}

/* the split rule of lib/edlib/edlib.cpp:1263-1289 on two last-column functions */
template <class FF, class RF>
int split_rule(int n, int lw, int rw, int best, FF F, RF R, int &ls, int &rs)
{
    for (int qi = 0; qi + 2 <= n; qi++) if (F(qi + 1) + R(n - qi - 1) == best) { ls = F(qi + 1); rs = R(n - qi - 1); return qi; }
    if (lw + R(n) == best) { ls = lw; rs = R(n); return -1; }
    if (F(n) + rw == best) { ls = F(n); rs = rw; return n - 1; }
    return -2;
}

}  // namespace

extern "C" {

/* returns the number of wavefronts per half the device would pick for this band (0: does not fit any class up to Wmax) */
int hbm_pick_w(int dlo, int dhi, int n, int mm, int L, int Wmax)
{
    Band B; B.dlo = dlo; B.dhi = dhi;
    for (int W = 1; W <= Wmax; W *= 2) if (band_fits(B, n, mm, L, W)) return W;
    return 0;
}

/* NW node with the banded sweeps.  k = the node's distance (trial = 0) or a trial bound (trial = 1).
 * out[0] = status (1 ok, 0 trial failed, -1 band does not fit, -2 no split row), out[1] = split, out[2] = ls, out[3] = rs, out[4] = best,
 * out[5] = wavefronts per half, out[6] = sweep steps of all lanes, out[7] = active steps */
void hbm_node(const int8_t *q, int n, const int8_t *t, int m, int k, int trial, int L, int Wmax, long *out)
{
    const int lw = m / 2, rw = m - lw;
    const Band B = nw_band(n, m, k);
    const int W = hbm_pick_w(B.dlo, B.dhi, n, rw, L, Wmax);
    out[5] = W;
    if (!W) { out[0] = -1; return; }
    std::vector<int8_t> qr(q, q + n), tr(t, t + m);
    std::reverse(qr.begin(), qr.end()); std::reverse(tr.begin(), tr.end());
    HalfOut FO, RO;
    sweep_half(q, n, t, lw, B, false, L, W, FO);
    sweep_half(qr.data(), n, tr.data(), rw, B, false, L, W, RO);
    out[6] = FO.steps + RO.steps; out[7] = FO.active_steps + RO.active_steps;
    auto F = [&](int x) { return lw ? col_at(FO, x, lw, n) : x; };
    auto R = [&](int x) { return col_at(RO, x, rw, n); };
    int best = k;
    if (trial) {
        int mn = INF;
        for (int r = 0; r <= n; r++) mn = std::min(mn, F(r) + R(n - r));
        if (mn > k) { out[0] = 0; out[4] = mn; return; }
        best = mn;
    }
    int ls = 0, rs = 0;
    const int sp = split_rule(n, lw, rw, best, F, R, ls, rs);
    out[0] = sp == -2 ? -2 : 1; out[1] = sp; out[2] = ls; out[3] = rs; out[4] = best;
}

/* the same node by plain DP over the full matrix */
void hbm_ref_node(const int8_t *q, int n, const int8_t *t, int m, long *out)
{
    const int lw = m / 2, rw = m - lw;
    std::vector<int8_t> qr(q, q + n), tr(t, t + m);
    std::reverse(qr.begin(), qr.end()); std::reverse(tr.begin(), tr.end());
    std::vector<int> Fc, Rc, Wc;
    dp_last_col(q, n, t, lw, Fc); dp_last_col(qr.data(), n, tr.data(), rw, Rc); dp_last_col(q, n, t, m, Wc);
    const int best = Wc[n];
    auto F = [&](int x) { return Fc[x]; };
    auto R = [&](int x) { return Rc[x]; };
    int ls = 0, rs = 0;
    const int sp = split_rule(n, lw, rw, best, F, R, ls, rs);
    out[0] = sp == -2 ? -2 : 1; out[1] = sp; out[2] = ls; out[3] = rs; out[4] = best;
}

/* SHW root, banded with the trial bound k: out[0] = status (1 ok, 0 k too small, -1 band does not fit), out[1] = distance, out[2] = end column (1-based
 * count of target symbols used; 0 = empty prefix) */
void hbm_shw(const int8_t *q, int n, const int8_t *t, int m, int k, int L, int Wmax, long *out)
{
    const Band B = shw_band(k);
    const long mme = std::min<long>(m, (long)n + k);
    const int W = hbm_pick_w(B.dlo, B.dhi, n, (int)mme, L, Wmax);
    out[5] = W;
    if (!W) { out[0] = -1; return; }
    HalfOut O;
    sweep_half(q, n, t, (int)mme, B, true, L, W, O);
    out[6] = O.steps; out[7] = O.active_steps;
    if (O.shw_best > k) { out[0] = 0; out[1] = O.shw_best; return; }
    out[0] = 1; out[1] = O.shw_best; out[2] = O.shw_c;
}
void hbm_ref_shw(const int8_t *q, int n, const int8_t *t, int m, long *out)
{
    /* last row of the full matrix, column by column (lib/edlib/edlib.cpp:583-618: smallest column on ties; the empty prefix only when n % 64 != 0) */
    std::vector<int> col(n + 1);
    for (int r = 0; r <= n; r++) col[r] = r;
    int best = (n & 63) ? n : 0x7fffffff, bc = 0;
    for (int j = 0; j < m; j++) {
        int diag = col[0]; col[0] = j + 1;
        for (int r = 1; r <= n; r++) {
            const int up = col[r - 1] + 1, left = col[r] + 1, dg = diag + ((q[r - 1] >= 0 && q[r - 1] == t[j]) ? 0 : 1);
            diag = col[r];
            col[r] = std::min(std::min(up, left), dg);
        }
        if (col[n] < best) { best = col[n]; bc = j + 1; }
    }
    out[0] = 1; out[1] = best; out[2] = bc;
}

}
