"""The oracle (oracle/liblforacle.so) against the committed golden vectors that were produced by the
real reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import GOLDEN_CONFIGS, golden_sam, split_ragged


@pytest.mark.parametrize("cfg", list(GOLDEN_CONFIGS))
def test_sam_matches_reference(oracle, oracle_lib, golden_reads, cfg):
    names, seqs = golden_reads
    p = oracle_lib.default_params(**GOLDEN_CONFIGS[cfg])
    got = oracle.map_batch(names, seqs, params=p)
    exp = golden_sam(cfg)
    assert got.count(b"\n") == exp.count(b"\n")
    assert got == exp


def test_sam_threads_and_order(oracle, oracle_lib, golden_reads):
    names, seqs = golden_reads
    a = oracle.map_batch(names, seqs, params=oracle_lib.default_params(threads=4))
    assert a == golden_sam("default")


def test_branch_coverage_of_fixture():
    """the fixture must actually reach the rare branches: split (supplementary), inverted segment,
    secondary (fine mode), unmapped, clipping"""
    sam = golden_sam("default").splitlines()
    flags = [int(l.split(b"\t")[1]) for l in sam]
    assert any(f & 2048 for f in flags), "no supplementary record (split branch)"
    assert any(f & 256 for f in flags), "no secondary record (fine mode)"
    assert any(f == 4 for f in flags), "no unmapped record"
    assert any(f & 16 for f in flags) and any(f == 0 for f in flags)
    by_read = {}
    for l in sam:
        f = l.split(b"\t")
        by_read.setdefault(f[0], []).append(int(f[1]))
    # inverted middle segment: a supplementary on the opposite strand of its primary
    assert any((fl[0] & 16) != (x & 16) for fl in by_read.values() for x in fl[1:] if x & 2048), \
        "no inverted supplementary segment"
    assert any(b"S" in l.split(b"\t")[5] for l in sam if l.split(b"\t")[5] != b"*")


def test_seeds(oracle, golden_reads, stages):
    names, seqs = golden_reads
    F = split_ragged(stages["seed_F"], stages["seed_F_n"])
    R = split_ragged(stages["seed_R"], stages["seed_R_n"])
    assert len(F) == len(seqs)
    for s, f, r in zip(seqs, F, R):
        gf, gr = oracle.seed(s)
        assert np.array_equal(gf, f)
        assert np.array_equal(gr, r)


def test_chain_n2(oracle, stages):
    ins = split_ragged(stages["chain_in"], stages["chain_n"])
    srt = split_ragged(stages["chain_sorted"], stages["chain_n"])
    outs = split_ragged(stages["chain_out"], stages["chain_out_n"])
    for sd, ss, ch, sc in zip(ins, srt, outs, stages["chain_score"]):
        s2, c2, score = oracle.chain_n2(sd)
        assert np.array_equal(s2, ss), "introsort order differs from std::sort"
        assert np.array_equal(c2, ch)
        assert np.float32(score) == np.float32(sc)


def test_chain_clasp(oracle, stages_clasp):
    st = stages_clasp
    ins = split_ragged(st["clasp_in"], st["clasp_n"])
    outs = split_ragged(st["clasp_out"], st["clasp_out_n"])
    assert max(st["clasp_out_n"]) > 20
    for sd, ch, sc in zip(ins, outs, st["clasp_score"]):
        c2, score = oracle.chain_clasp(sd)
        assert np.array_equal(c2, ch)
        assert np.float32(score) == np.float32(sc)


def test_edlib(oracle, stages):
    qs = split_ragged(stages["ed_q"].tobytes(), stages["ed_qn"])
    ts = split_ragged(stages["ed_t"].tobytes(), stages["ed_tn"])
    ops = split_ragged(stages["ed_ops"], stages["ed_opsn"])
    n_hirsch = 0
    for q, t, mode, ed, end, op in zip(qs, ts, stages["ed_mode"], stages["ed_dist"], stages["ed_end"], ops):
        g_ed, g_end, g_ops = oracle.edlib(q, t, int(mode))
        assert (g_ed, g_end) == (int(ed), int(end))
        assert np.array_equal(g_ops, op)
        n_hirsch += (20 * ((len(q) + 63) // 64) * len(t) + 8 * len(t)) >= 1 << 20
    assert n_hirsch >= 4
    assert (stages["ed_end"] == -1).any(), "fixture lacks the SHW empty-prefix case"


def test_ksw_extend2(oracle, stages):
    qs = split_ragged(stages["ksw_q"], stages["ksw_qn"])
    ts = split_ragged(stages["ksw_t"], stages["ksw_tn"])
    for q, t, prm, res in zip(qs, ts, stages["ksw_prm"], stages["ksw_res"]):
        got = oracle.ksw_extend2(q, t, *[int(x) for x in prm])
        assert tuple(got) == tuple(int(x) for x in res)
