"""GPU parity, stage level: lf_edlib_batch / lf_chain_n2_batch / lf_ksw_extend2_batch (HIP, through the
C ABI) vs the golden vectors produced by the real reference, and vs the oracle on seeded random inputs."""
import os

import numpy as np
import pytest

from conftest import split_ragged
from lordfast_amd import synth

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rseq(rng, n, alphabet=ACGT, p=None):
    return bytes(rng.choice(alphabet, size=n, p=p))


def test_chain_clasp_golden(stages_clasp):
    """lf_chain_clasp_batch (lf_clasp_kernel.h) vs chain_seeds_clasp of the compiled reference"""
    import lordfast_amd as la
    st = stages_clasp
    ins = split_ragged(st["clasp_in"], st["clasp_n"])
    outs = split_ragged(st["clasp_out"], st["clasp_out_n"])
    res = la.chain_clasp_batch(ins)
    for i, ((ch, sc), exp, esc) in enumerate(zip(res, outs, st["clasp_score"])):
        assert np.array_equal(ch, exp), (i, len(ins[i]), len(ch), len(exp))
        assert np.float32(sc) == np.float32(esc), i


def test_chain_clasp_fuzz_vs_oracle(oracle_lib):
    import lordfast_amd as la
    from test_oracle_vs_ref import clasp_window
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(21)
    wins = [clasp_window(rng, it % 5) for it in range(1200)]
    wins += [np.zeros((0, 3), dtype=np.uint32)]                                 # empty window: score -1, no chain
    for n in (513, 700, 1500):                                                  # beyond the LDS classes: HBM workspace
        q = np.sort(rng.integers(0, 30000, size=n))
        t = q + rng.choice([100000, 100004, 160000], size=n) + rng.integers(-2, 3, size=n)
        wins.append(np.stack([t, q, rng.integers(14, 22, size=n)], axis=1).astype(np.uint32)[rng.permutation(n)])
    res = la.chain_clasp_batch(wins)
    for i, (w, (ch, sc)) in enumerate(zip(wins, res)):
        ech, esc = orc.chain_clasp(w)
        assert np.array_equal(ch, ech), (i, len(w), len(ch), len(ech))
        assert np.float32(sc) == np.float32(esc), i


def test_edlib_golden(stages):
    import lordfast_amd as la
    qs = split_ragged(stages["ed_q"].tobytes(), stages["ed_qn"])
    ts = split_ragged(stages["ed_t"].tobytes(), stages["ed_tn"])
    ops = split_ragged(stages["ed_ops"], stages["ed_opsn"])
    res, ms = la.edlib_batch(qs, ts, stages["ed_mode"])
    assert len(res) == len(qs)
    for i, (r, ed, end, op) in enumerate(zip(res, stages["ed_dist"], stages["ed_end"], ops)):
        assert (r[0], r[1]) == (int(ed), int(end)), (i, len(qs[i]), len(ts[i]), int(stages["ed_mode"][i]))
        assert np.array_equal(r[2], op), i


def test_edlib_fuzz_vs_oracle(oracle_lib):
    import lordfast_amd as la
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(11)
    qs, ts, modes = [], [], []
    lens = list(range(1, 40)) + [62, 63, 64, 65, 66, 127, 128, 129, 191, 192, 193, 255, 256, 257, 320, 511, 512, 513, 700, 1100]
    for it in range(1500):
        n = int(rng.choice(lens))
        kind = it % 5
        if kind == 0:
            q = rseq(rng, n); t = rseq(rng, int(rng.choice(lens)))
        elif kind == 1:
            q = rseq(rng, n, ACGT[:2], [0.85, 0.15]); t = rseq(rng, int(rng.choice(lens)), ACGT[:2], [0.85, 0.15])
        else:
            q = rseq(rng, n)
            t = synth.mutate(np.frombuffer(q, dtype=np.uint8), float(rng.uniform(0.0, 0.4)), rng).tobytes() or q
        if it % 37 == 0:
            q = q.lower()                      # raw-byte comparison: lower case never matches the reference
        if it % 41 == 0:
            q = q[:len(q) // 2] + b"N" + q[len(q) // 2:]
        for mode in (0, 1):
            tt = t + rseq(rng, int(rng.integers(0, 25))) if mode == 1 else t
            qs.append(q); ts.append(tt); modes.append(mode)
    # Hirschberg regime and a skinny-but-long problem
    for n, e in ((1900, 0.15), (3000, 0.1), (2400, 0.3)):
        q = np.frombuffer(rseq(rng, n), dtype=np.uint8)
        t = synth.mutate(q, e, rng)
        q2 = np.concatenate([q[:n // 2], np.frombuffer(b"AT" * 150, dtype=np.uint8), q[n // 2:]])
        qs += [q.tobytes(), q2.tobytes(), q.tobytes()]; ts += [t.tobytes(), t.tobytes(), t.tobytes() + rseq(rng, 20)]; modes += [0, 0, 1]
    qs.append(rseq(rng, 90)); ts.append(rseq(rng, 40000)); modes.append(1)
    res, ms = la.edlib_batch(qs, ts, modes)
    for i, r in enumerate(res):
        o = orc.edlib(qs[i], ts[i], modes[i])
        assert (r[0], r[1]) == (o[0], o[1]), (i, len(qs[i]), len(ts[i]), modes[i])
        assert np.array_equal(r[2], o[2]), (i, len(qs[i]), len(ts[i]), modes[i])


@pytest.mark.parametrize("band", [16, 64, 0])
def test_edlib_checkpoint_tiles_and_device_hirschberg(oracle_lib, monkeypatch, band):
    """shapes aimed at the recompute-from-checkpoint traceback and the breadth-first Hirschberg levels (lf_rsweep.hip,
    lf_align.hip, lf_hirsch.hip): tile boundaries (m around multiples of 8 / 16), paths that climb > 64 rows inside one
    16-column tile, every block count per problem (1 .. 64 lanes) and the 4 / 8 blocks-per-lane classes, targets longer
    than the LDS ring of the level kernels, tall-and-thin / short-and-wide leaves, several recursion levels, SHW roots
    whose prefix is a leaf"""
    import lordfast_amd as la
    # 16: nodes are swept inside the band of their distance (sixteen / thirty-two lanes, or one to eight wavefronts per half; 1, the default, leaves the lane groups to calls of
    # 16 384 roots and more); 64: whole wavefronts only; 0: every block of every column
    monkeypatch.setenv("LF_HIRSCH_BAND", str(band))
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(77)
    qs, ts, modes = [], [], []

    def add(q, t, both=True):
        for mode in ((0, 1) if both else (0,)):
            qs.append(bytes(q)); ts.append(bytes(t) + (rseq(rng, 7) if mode else b"")); modes.append(mode)

    for m in list(range(1, 20)) + [23, 24, 25, 31, 32, 33, 63, 64, 65, 127, 128, 129]:          # tile edges, lane classes
        for n in (5, 64, 65, 130, 200, 300, 420, 512):
            q = rseq(rng, n); t = synth.mutate(np.frombuffer(q, dtype=np.uint8), 0.2, rng).tobytes()[:m] or b"A"
            add(q, t)
    for n, ins_at, ins_len in ((300, 10, 180), (500, 3, 400), (200, 100, 90), (512, 250, 200), (128, 5, 100)):   # big insertions: window exits
        t = rseq(rng, n - ins_len)
        q = t[:ins_at] + rseq(rng, ins_len) + t[ins_at:]
        add(q, t)
        add(t, q)                                                                                  # and big deletions
    for n in (513, 640, 1000, 1024, 1025, 1500, 2048, 2049, 3000, 4096, 4097, 6000):               # group / wave classes, leaf size
        q = np.frombuffer(rseq(rng, n), dtype=np.uint8)
        m_cap = max(40, (1 << 20) // (20 * ((n + 63) // 64) + 8) - 30)                             # stay below edlib's traceback switch
        t = synth.mutate(q, float(rng.uniform(0.05, 0.3)), rng)[:m_cap]
        add(q.tobytes(), t.tobytes())
    add(rseq(rng, 3000), rseq(rng, 45))                                                            # tall and thin
    add(rseq(rng, 45), rseq(rng, 5000))                                                            # short and wide (target longer than the LDS ring of its class)
    add(rseq(rng, 700), rseq(rng, 3000))                                                           # group class, ring refills
    for n, e in ((2500, 0.12), (5000, 0.15), (5000, 0.45), (9000, 0.1), (17000, 0.08)):            # Hirschberg on the device: 1-4 levels, KB 1 / 4 / 8
        q = np.frombuffer(rseq(rng, n), dtype=np.uint8)
        t = synth.mutate(q, e, rng)
        add(q.tobytes(), t.tobytes(), both=(n <= 5000))
    q = rseq(rng, 2600)                                                                             # SHW: (n, m) above the switch, (n, end) below it
    qs.append(q); ts.append(synth.mutate(np.frombuffer(q[:900], dtype=np.uint8), 0.1, rng).tobytes() + rseq(rng, 3000)); modes.append(1)
    for n, m in ((1900, 1900), (2040, 2100), (1000, 3400), (500, 6500), (60, 38000)):             # short queries above the switch: Hirschberg too
        q = np.frombuffer(rseq(rng, n, ACGT[:2], [0.7, 0.3]), dtype=np.uint8)                      # two-letter strings: ties everywhere
        t = np.concatenate([synth.mutate(q, 0.2, rng), np.frombuffer(rseq(rng, max(0, m - n), ACGT[:2], [0.7, 0.3]), dtype=np.uint8)])[:m]
        add(q.tobytes(), t.tobytes())
    q = rseq(rng, 2100); t = b"AC" * 1200 + q[1000:] + b"GT" * 900                                  # low complexity: many co-optimal paths, tie rules decide
    add(q, t)
    res, ms = la.edlib_batch(qs, ts, modes)
    want = []
    for i, r in enumerate(res):
        o = orc.edlib(qs[i], ts[i], modes[i])
        want.append(o)
        assert (r[0], r[1]) == (o[0], o[1]), (i, len(qs[i]), len(ts[i]), modes[i])
        assert np.array_equal(r[2], o[2]), (i, len(qs[i]), len(ts[i]), modes[i])
    if band == 16:
        # roots of 1 800 - 6 000 rows inside tight trial bounds: SHW roots on sixteen / thirty-two lanes (two sixteenths of 3 700 rows fit 946 diagonals), most of
        # the 20 - 45 % ones failing their trial and going back with the bound they found; one sixteenth: nearly every trial fails
        for trial in ("2,2", "1,1"):
            monkeypatch.setenv("LF_HIRSCH_TRIAL", trial)
            res2, _ = la.edlib_batch(qs, ts, modes)
            for i, (r, o) in enumerate(zip(res2, want)):
                assert (r[0], r[1]) == (o[0], o[1]) and np.array_equal(r[2], o[2]), (trial, i, len(qs[i]), len(ts[i]), modes[i])


@pytest.mark.parametrize("band", [16, 64, 0])
def test_edlib_queries_above_32768_rows(oracle_lib, monkeypatch, band):
    """queries longer than one wavefront holds as register-resident blocks (64 lanes x 8 blocks x 64 rows = 32 768): the
    Hirschberg levels sweep them in row bands whose boundary carries go through HBM (lf_hirsch.hip); NW and SHW roots,
    a band boundary one row before the end of the query, a tall-and-thin problem above the traceback switch"""
    import lordfast_amd as la
    monkeypatch.setenv("LF_HIRSCH_BAND", str(band))      # 1: banded where the band fits four wavefronts (the 41 k-row root, the children of all); 0: super-bands only
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(3276833)
    qs, ts, modes = [], [], []
    for n, e, mode, tail in ((32769, 0.12, 0, 0), (33000, 0.1, 1, 900), (41000, 0.2, 0, 0), (70001, 0.08, 0, 0), (66000, 0.1, 1, 300)):      # the last two: three super-bands
        q = np.frombuffer(rseq(rng, n), dtype=np.uint8)
        t = synth.mutate(q, e, rng).tobytes() + (rseq(rng, tail) if tail else b"")
        qs.append(q.tobytes()); ts.append(t); modes.append(mode)
    q = rseq(rng, 36000)
    qs.append(q); ts.append(q[17000:17300]); modes.append(0)
    res, ms = la.edlib_batch(qs, ts, modes)
    for i, r in enumerate(res):
        o = orc.edlib(qs[i], ts[i], modes[i])
        assert (r[0], r[1]) == (o[0], o[1]), (i, len(qs[i]), len(ts[i]), modes[i])
        assert np.array_equal(r[2], o[2]), (i, len(qs[i]), len(ts[i]), modes[i])


def test_edlib_banded_levels_on_low_complexity_strings(oracle_lib, monkeypatch):
    """the banded Hirschberg levels (lf_hirsch.hip: lf_hband_level_kernel) where ties decide: two-letter and tandem-repeat strings of 4 500 - 20 000
    rows, both modes -- the split rule `first row with left + right == best` (lib/edlib/edlib.cpp:1263-1289) must see every candidate row although only
    the band's diagonals are swept.  Roots whose trial bound holds (15 % error), roots whose bound fails (unrelated strings: back to the unbanded sweep),
    bands for one / two / four wavefronts, a lane that takes a second and a third block (queries above 4096 + band rows).  Every result against the
    oracle, and the banded run against the unbanded one."""
    import lordfast_amd as la
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(60606)
    qs, ts, modes = [], [], []

    def add(q, t, ms=(0, 1)):
        for mode in ms:
            qs.append(bytes(q)); ts.append(bytes(t) + (rseq(rng, 300) if mode else b"")); modes.append(mode)

    for n, e in ((4500, 0.15), (6000, 0.1), (9000, 0.17), (12000, 0.12)):
        q = np.frombuffer(rseq(rng, n, ACGT[:2], [0.6, 0.4]), dtype=np.uint8)                       # two letters
        add(q.tobytes(), synth.mutate(q, e, rng).tobytes())
        unit = rseq(rng, int(rng.integers(2, 9)))
        q = np.frombuffer((unit * (n // len(unit) + 1))[:n], dtype=np.uint8)                        # a tandem array: every shift of a unit is co-optimal
        add(q.tobytes(), synth.mutate(q, e, rng).tobytes())
    q = np.frombuffer(rseq(rng, 20000), dtype=np.uint8)                                              # 15 %: the band of the children needs one, the root's trial two wavefronts
    add(q.tobytes(), synth.mutate(q, 0.15, rng).tobytes())
    q = np.frombuffer(rseq(rng, 16000), dtype=np.uint8)                                              # 30 %: the root's trial bound fails, the children's bands are wide
    add(q.tobytes(), synth.mutate(q, 0.30, rng).tobytes(), ms=(0,))
    add(rseq(rng, 5200), rseq(rng, 5600))                                                            # unrelated: trial fails, children near the width limit
    add(rseq(rng, 9000), rseq(rng, 2500), ms=(0,))                                                   # much longer than the target: the band is the length difference
    q = rseq(rng, 7000); add(q, q[:3000] + q[3400:], ms=(0,))                                        # one long gap, otherwise identical: distance == |m - n|, band of +- 0
    q = rseq(rng, 4200 + 64 * 3); add(q, q)                                                           # identical strings
    # the roots' trial bounds follow what the process has seen (lf_align.hip: lf_htrial_pick); fixed here: generous, tight (most trials fail), none
    runs = []
    monkeypatch.setenv("LF_HIRSCH_BAND", "16")                                                      # lane-group queues for the narrow bands although the call is small
    for trial in ("4,4", "2,2", "0,0"):
        monkeypatch.setenv("LF_HIRSCH_TRIAL", trial)
        runs.append(la.edlib_batch(qs, ts, modes)[0])
    monkeypatch.delenv("LF_HIRSCH_TRIAL")
    runs.append(la.edlib_batch(qs, ts, modes)[0])                                                   # ... and chosen from the three runs above
    monkeypatch.setenv("LF_HIRSCH_BAND", "1")
    runs.append(la.edlib_batch(qs, ts, modes)[0])                                                   # the default: whole wavefronts for a call of a few roots
    res = runs[0]
    for other in runs[1:]:
        for i, (r, r0) in enumerate(zip(res, other)):
            assert (r[0], r[1]) == (r0[0], r0[1]) and np.array_equal(r[2], r0[2]), ("trial bounds changed a result", i, len(qs[i]), len(ts[i]), modes[i])
    monkeypatch.setenv("LF_HIRSCH_BAND", "0")
    res0, _ = la.edlib_batch(qs, ts, modes)
    for i, (r, r0) in enumerate(zip(res, res0)):
        assert (r[0], r[1]) == (r0[0], r0[1]) and np.array_equal(r[2], r0[2]), ("banded != unbanded", i, len(qs[i]), len(ts[i]), modes[i])
        o = orc.edlib(qs[i], ts[i], modes[i])
        assert (r[0], r[1]) == (o[0], o[1]), (i, len(qs[i]), len(ts[i]), modes[i])
        assert np.array_equal(r[2], o[2]), (i, len(qs[i]), len(ts[i]), modes[i])


def test_edlib_empty_batch():
    import lordfast_amd as la
    res, _ = la.edlib_batch([], [], [])
    assert res == []


def test_chain_golden(stages):
    import lordfast_amd as la
    ins = split_ragged(stages["chain_in"], stages["chain_n"])
    srt = split_ragged(stages["chain_sorted"], stages["chain_n"])
    outs = split_ragged(stages["chain_out"], stages["chain_out_n"])
    res = la.chain_n2_batch(ins)
    for r, ss, ch, sc in zip(res, srt, outs, stages["chain_score"]):
        assert np.array_equal(r[0], ss), "introsort order differs from std::sort"
        assert np.array_equal(r[1], ch)
        assert np.float32(r[2]) == np.float32(sc)


def test_chain_fuzz_vs_oracle(oracle_lib):
    import lordfast_amd as la
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(4)
    wins = []
    for it in range(300):
        n = int(rng.integers(0, 400)) if it % 10 else int(rng.integers(400, 6000))
        nq = max(1, int(n * rng.uniform(0.05, 1.0)))
        qpool = np.sort(rng.integers(0, 20000, size=nq))
        q = qpool[rng.integers(0, nq, size=n)].astype(np.uint32) if n else np.zeros(0, np.uint32)
        if it % 4 == 0:
            t = (q.astype(np.int64) + rng.integers(-2, 3, size=n) + 100000).astype(np.uint32)
        elif it % 4 == 1:
            t = (q.astype(np.int64) + rng.choice([100000, 100003, 250000], size=n)).astype(np.uint32)
        else:
            t = rng.integers(50000, 90000, size=n).astype(np.uint32)
        ln = rng.integers(14, 25, size=n).astype(np.uint32)
        wins.append(np.stack([t, q, ln], axis=1) if n else np.zeros((0, 3), dtype=np.uint32))
    # large windows (the workgroup kernel, lf_chain_kernel.h: 513 .. 16 384 seeds; above: the one-wavefront kernel on its HBM workspace): satellite-array shapes --
    # few distinct query positions, a handful of diagonals, equal lengths: ties in dp everywhere, the reference's `first j from the top wins` decides
    for n, nq, kind in ((600, 40, 0), (2048, 300, 1), (2049, 2049, 2), (4096, 500, 0), (4097, 64, 1), (8000, 900, 0), (8000, 8000, 2), (12000, 200, 1), (16384, 3000, 0), (17000, 1500, 1)):
        qpool = np.sort(rng.integers(0, 60000, size=nq))
        q = np.sort(qpool[rng.integers(0, nq, size=n)]).astype(np.uint32)
        if kind == 0:
            t = (q.astype(np.int64) + rng.choice([100000, 100171, 100342, 100513], size=n)).astype(np.uint32)       # a few diagonals one monomer apart
        elif kind == 1:
            t = (q.astype(np.int64) * 0 + rng.integers(100000, 100000 + 171 * 40, size=n)).astype(np.uint32)       # any query position anywhere in the array
        else:
            t = (q.astype(np.int64) + rng.integers(-3, 4, size=n) + 100000).astype(np.uint32)                      # one noisy diagonal
        ln = np.full(n, 17, dtype=np.uint32) if kind != 2 else rng.integers(14, 25, size=n).astype(np.uint32)
        wins.append(np.stack([t, q, ln], axis=1))
    for kw in (dict(), dict(chain_reward=5.0, chain_penalty=8.0, min_anchor_len=17)):
        res = la.chain_n2_batch(wins, la.default_params(**kw))
        for i, (w, r) in enumerate(zip(wins, res)):
            o = orc.chain_n2(w, oracle_lib.default_params(**kw))
            assert np.array_equal(r[0], o[0]), i
            assert np.array_equal(r[1], o[1]), (i, len(w))
            assert np.float32(r[2]) == np.float32(o[2]), i


def test_ksw_golden_and_fuzz(stages, oracle_lib):
    import lordfast_amd as la
    qs = split_ragged(stages["ksw_q"], stages["ksw_qn"])
    ts = split_ragged(stages["ksw_t"], stages["ksw_tn"])
    res = la.ksw_extend2_batch(qs, ts, stages["ksw_prm"])
    for r, exp in zip(res, stages["ksw_res"]):
        assert tuple(r) == tuple(int(x) for x in exp)
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(3)
    qs, ts, prms = [], [], []
    for it in range(600):
        n = int(rng.integers(1, 700))
        q = rng.integers(0, 4, size=n).astype(np.uint8)
        if it % 3 == 0:
            t = rng.integers(0, 4, size=int(rng.integers(1, 700))).astype(np.uint8)
        else:
            keep = int(rng.integers(1, n + 1))
            t = np.concatenate([q[:keep], rng.integers(0, 4, size=int(rng.integers(0, 200))).astype(np.uint8)])
            mut = rng.random(len(t)) < rng.uniform(0, 0.2)
            t[mut] = rng.integers(0, 4, size=int(mut.sum())).astype(np.uint8)
        if it % 11 == 0:
            q[rng.integers(0, len(q), size=2)] = 4
        qs.append(q); ts.append(t)
        prms.append((0, 1, 0, 1, 40, 40, len(q)) if it % 2 else (8, 1, 4, 1, 100, 200, len(q)))
    # long problems with indels: several 64-column tiles per band row, band trimming and z-drop at work, queries beyond
    # the LDS limit of the kernel (6000) in the HBM-workspace instantiation, small and large h0
    code = np.zeros(256, dtype=np.uint8); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
    for it in range(60):
        n = int(rng.integers(800, 7500 if it % 6 == 0 else 3000))
        qa = np.frombuffer(rseq(rng, n), dtype=np.uint8)
        ta = synth.mutate(qa, float(rng.uniform(0.02, 0.3)), rng)
        if it % 5 == 0:
            ta = np.concatenate([ta[:len(ta) // 2], np.frombuffer(rseq(rng, int(rng.integers(50, 400))), dtype=np.uint8), ta[len(ta) // 2:]])
        q, t = code[qa], code[ta]
        h0 = len(q) if it % 3 else int(rng.integers(1, 200))
        qs.append(q); ts.append(t)
        prms.append((0, 1, 0, 1, 40, 40, h0) if it % 2 else (8, 1, 4, 1, 100, 200, h0))
    res = la.ksw_extend2_batch(qs, ts, prms)
    for i, r in enumerate(res):
        assert tuple(r) == tuple(orc.ksw_extend2(qs[i], ts[i], *prms[i])), (i, len(qs[i]), len(ts[i]), prms[i])


WINDOW_CONFIGS = {"default": dict(), "n30": dict(max_map=30), "k12c300m20": dict(min_anchor_len=12, sampling_count=300, max_ref_hits=20),
                  "clasp": dict(chain_alg=1), "clasp_n30": dict(chain_alg=1, max_map=30)}


@pytest.mark.parametrize("cfg", list(WINDOW_CONFIGS))
def test_windows_and_alignwin_golden(golden_dir, golden_reads, cfg):
    """Stage vectors of the window vote and alignWin, dumped from the reference's own
    findTopWins_coarse / _fine / alignWin (src/LordFAST.cpp:582-657, 819-904, 995-1189; oracle/ref_harness.cpp:
    ref_stage_windows): the coarse / fine decision, the windows alignWin is called with (float score bits; fine mode: the
    top-N heap array in array order, i.e. with libstdc++'s push_heap / pop_heap moves) and alignWin's totalScore + record
    fields per window -- for dp-n2 and clasp."""
    import lordfast_amd as la
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "stages_windows.npz"))
    names, seqs = golden_reads
    lf = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0)
    try:
        got = lf.map_stages(seqs, params=la.default_params(**WINDOW_CONFIGS[cfg]))
    finally:
        lf.close()
    mode = z[f"{cfg}_mode"]; wn = z[f"{cfg}_wins_n"]; wins = z[f"{cfg}_wins"]; mn = z[f"{cfg}_maps_n"]; maps = z[f"{cfg}_maps"]
    assert len(got) == len(mode)
    wo = np.concatenate([[0], np.cumsum(wn)]); mo = np.concatenate([[0], np.cumsum(mn)])
    n_fine = 0
    for i, g in enumerate(got):
        assert g["mode"] == mode[i], (i, g["mode"], mode[i])
        ew = wins[wo[i]:wo[i + 1]]
        assert g["wins"].shape == ew.shape and np.array_equal(g["wins"], ew), (i, names[i], g["wins"], ew)
        em = maps[mo[i]:mo[i + 1]]
        assert np.array_equal(g["maps"], em), (i, names[i], g["maps"][:12], em[:12])
        n_fine += mode[i] == 3
    assert n_fine >= 5 and (mode == 2).sum() >= 20, "fixture must exercise both branches"
