/* hqueue_check.hip -- host-side check of lf_hqueue_of / lf_hqueue_of_bound (lordfast_amd/csrc/lf_hirsch.h): the queue a Hirschberg node is sent to must be
 * one whose kernel HOLDS the node's band (lanes x wavefronts), and the narrowest such one.  No kernel is launched: the functions are __host__ __device__.
 * Test infrastructure (tests/test_hband_model.py builds and runs it); nothing in the product includes this file. */
#include "lf_hirsch.h"
#include <cstdio>
#include <cstdlib>
#include <cstdint>

static uint64_t rng_s = 88172645463325252ull;
static uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 11); }

/* capacity (diagonals) of a queue's kernel, -1: the unbanded queues hold anything */
static int lanes_of(int q) { return q == LF_HQ_NW16 || q == LF_HQ_SHW16 ? 16 : q == LF_HQ_NW32 || q == LF_HQ_SHW32 ? 32 : 0; }
static int waves_of(int q) { return q >= LF_HQ_SHW0 && q < LF_HQ_SHW0 + 5 ? 1 << (q - LF_HQ_SHW0) : q >= LF_HQ_NW0 && q < LF_HQ_NW0 + 4 ? 1 << (q - LF_HQ_NW0) : 0; }
static bool holds(int q, lf_hband B, uint32_t n, int mm) { return lanes_of(q) ? lf_hband_fits_lanes(B, n, mm, lanes_of(q)) : waves_of(q) ? lf_hband_fits(B, n, mm, waves_of(q)) : true; }

int main()
{
    long checked = 0, by_queue[LF_HQ] = { 0 };
    for (int it = 0; it < 400000; it++) {
        const uint32_t n = 1 + rnd() % (it % 7 == 0 ? 200000u : it % 3 == 0 ? 9000u : 40000u);
        uint32_t m = (uint32_t)((double)n * (0.5 + (rnd() % 1000) / 1000.0)) + 1; if (it % 11 == 0) m = 1 + rnd() % 300000u;
        const int kind = rnd() & 1;
        const unsigned pad = (it % 13 == 0 ? 1u : 0u) | (it % 5 == 0 ? 2u : 0u);
        const uint32_t trial16 = rnd() % 17, trial_min = it % 2 ? LF_HTRIAL_MIN_ROWS : 4096;
        const int best = kind == 0 && (rnd() & 1) ? (int)(rnd() % (n + m)) : -1;
        uint32_t k0 = 12345;
        int q;
        if (it % 4 == 3 && best < 0) q = lf_hqueue_of_bound(n, m, -1, kind, pad, rnd() % (n + m + 1), &k0);      /* a failed trial's second bound */
        else q = lf_hqueue_of(n, m, best, kind, pad, kind ? trial16 : trial16, trial_min, &k0);
        if (q < 0 || q >= LF_HQ) { printf("queue out of range %d\n", q); return 1; }
        by_queue[q]++;
        if (q < LF_HQ_NW0) {            /* unbanded by rows */
            if (q != lf_hkb_class(n) || k0 != 0) { printf("unbanded queue %d for n %u (k0 %u)\n", q, n, k0); return 1; }
            continue;
        }
        if (pad & 1u) { printf("a node that must not be banded went to queue %d\n", q); return 1; }
        if ((pad & 2u) && lanes_of(q)) { printf("lane-group queue %d although switched off\n", q); return 1; }
        const bool shwq = (q >= LF_HQ_SHW0 && q < LF_HQ_SHW0 + 5) || q == LF_HQ_SHW16 || q == LF_HQ_SHW32;
        if (shwq != (kind == 1)) { printf("kind %d in queue %d\n", kind, q); return 1; }
        lf_hband B; int mm;
        if (kind == 1) { B = lf_hband_shw((int)k0); const uint64_t me = (uint64_t)n + k0; mm = (int)(me < m ? me : m); }
        else { if (best >= 0 && k0 != 0) { printf("known distance but k0 %u\n", k0); return 1; } B = lf_hband_nw(n, m, best >= 0 ? best : (int)k0); mm = (int)(m - m / 2); }
        if (!holds(q, B, n, mm)) { printf("queue %d does not hold the band [%d, %d] of n %u m %u\n", q, B.dlo, B.dhi, n, m); return 1; }
        /* the narrowest: no lane-group / smaller wavefront class would have held it */
        if (!(pad & 2u)) { if (lanes_of(q) != 16 && lf_hband_fits_lanes(B, n, mm, 16)) { printf("16 lanes would do, queue %d\n", q); return 1; }
                           if (!lanes_of(q) && lf_hband_fits_lanes(B, n, mm, 32)) { printf("32 lanes would do, queue %d\n", q); return 1; } }
        if (waves_of(q) > 1 && lf_hband_fits(B, n, mm, waves_of(q) / 2)) { printf("half the wavefronts would do, queue %d\n", q); return 1; }
        if (k0 >= (1u << 31)) { printf("bound does not fit the kernels' int\n"); return 1; }      /* (a trial bound may exceed n + m for thin matrices: the band is then the matrix) */
        checked++;
    }
    for (int q = 0; q < LF_HQ; q++) if (by_queue[q] == 0) { printf("queue %d never chosen: the cases do not cover it\n", q); return 1; }
    printf("ok %ld banded choices\n", checked);
    return 0;
}
