"""GPU parity, stage 1: lf_seed_batch (HIP, through the C ABI) vs golden vectors and the oracle."""
import os

import numpy as np
import pytest

from conftest import split_ragged

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lf(golden_dir):
    import lordfast_amd as la
    assert la.device_count() >= 1, "no gfx950 device: the HIP path has no CPU fallback"
    h = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
    yield h
    h.close()


@pytest.fixture(scope="module")
def lf_walk(golden_dir):
    import lordfast_amd as la
    h = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=False)
    yield h
    h.close()


def test_seeds_vs_golden(lf, lf_walk, golden_reads, stages):
    names, seqs = golden_reads
    F = split_ragged(stages["seed_F"], stages["seed_F_n"])
    R = split_ragged(stages["seed_R"], stages["seed_R_n"])
    for h in (lf, lf_walk):
        gF, gR, info = h.seed_batch(seqs)
        assert len(gF) == len(seqs)
        for i in range(len(seqs)):
            assert np.array_equal(gF[i], F[i]), names[i]
            assert np.array_equal(gR[i], R[i]), names[i]
        assert info["n_sa"] == sum(len(x) for x in F) + sum(len(x) for x in R)


@pytest.mark.parametrize("kw", [dict(min_anchor_len=12, sampling_count=300, max_ref_hits=20),
                                dict(min_anchor_len=17, sampling_count=2000), dict(min_anchor_len=20, sampling_count=50)])
def test_seeds_vs_oracle_configs(lf, oracle, oracle_lib, golden_reads, kw):
    import lordfast_amd as la
    names, seqs = golden_reads
    gF, gR, _ = lf.seed_batch(seqs, la.default_params(**kw))
    for i, s in enumerate(seqs):
        oF, oR = oracle.seed(s, oracle_lib.default_params(**kw))
        assert np.array_equal(gF[i], oF), names[i]
        assert np.array_equal(gR[i], oR), names[i]


def test_seed_edge_inputs(lf, oracle):
    seqs = [b"ACGT" * 3, b"A" * 14, b"N" * 500, b"ACGTACGTACGTAC", b"acgtacgtacgtacgtacgt" * 30,
            b"ACGTTGCATGCATGCANNNNACGTAGCTAGCTAGCATCGATCAGCTACGACTAGC" * 40]
    gF, gR, _ = lf.seed_batch(seqs)
    for i, s in enumerate(seqs):
        oF, oR = oracle.seed(s)
        assert np.array_equal(gF[i], oF), i
        assert np.array_equal(gR[i], oR), i


def test_seed_empty_batch(lf):
    gF, gR, _ = lf.seed_batch([])
    assert gF == [] and gR == []


def test_seeds_with_wide_table(golden_dir, golden_reads, stages, monkeypatch):
    """LF_WIDE_TABLE=1 adds the 4^14-entry table (default only for genomes >= 2^28 symbols): same seeds, two search
    steps fewer per sample; -k 12 must still come out of the 12-mer table"""
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_WIDE_TABLE", "1")
    h = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
    monkeypatch.delenv("LF_WIDE_TABLE")
    base = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
    try:
        F = split_ragged(stages["seed_F"], stages["seed_F_n"])
        R = split_ragged(stages["seed_R"], stages["seed_R_n"])
        gF, gR, info = h.seed_batch(seqs)
        _, _, info0 = base.seed_batch(seqs)
        for i in range(len(seqs)):
            assert np.array_equal(gF[i], F[i]) and np.array_equal(gR[i], R[i]), names[i]
        assert info["n_occblk"] < info0["n_occblk"]
        for kw in (dict(min_anchor_len=12, sampling_count=300, max_ref_hits=20), dict(min_anchor_len=17, sampling_count=2000)):
            a = h.seed_batch(seqs[:30], params=la.default_params(**kw))
            b = base.seed_batch(seqs[:30], params=la.default_params(**kw))
            for x, y in zip(a[0] + a[1], b[0] + b[1]):
                assert np.array_equal(x, y)
    finally:
        h.close(); base.close()


def test_seeds_with_16mer_table(golden_dir, golden_reads, stages, monkeypatch):
    """LF_TABLE16=1 adds the 4^16-entry table (default only for genomes >= 2^30 symbols with HBM to spare: 68.7 GB): a sample
    whose 16-mer occurs starts there, any other falls back to the 14- / 12-mer table.  Same seeds for -k 14, 12 and 17."""
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_WIDE_TABLE", "1"); monkeypatch.setenv("LF_TABLE16", "1")
    h = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
    monkeypatch.delenv("LF_WIDE_TABLE"); monkeypatch.setenv("LF_TABLE16", "0")
    base = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
    try:
        F = split_ragged(stages["seed_F"], stages["seed_F_n"])
        R = split_ragged(stages["seed_R"], stages["seed_R_n"])
        gF, gR, info = h.seed_batch(seqs)
        _, _, info0 = base.seed_batch(seqs)
        for i in range(len(seqs)):
            assert np.array_equal(gF[i], F[i]) and np.array_equal(gR[i], R[i]), names[i]
        assert info["n_occblk"] < info0["n_occblk"]
        for kw in (dict(min_anchor_len=12, sampling_count=300, max_ref_hits=20), dict(min_anchor_len=17, sampling_count=2000), dict(min_anchor_len=16)):
            a = h.seed_batch(seqs[:30], params=la.default_params(**kw))
            b = base.seed_batch(seqs[:30], params=la.default_params(**kw))
            for x, y in zip(a[0] + a[1], b[0] + b[1]):
                assert np.array_equal(x, y)
        # and through the whole path
        from conftest import golden_sam
        sam, _ = h.map_batch(names, seqs)
        assert sam == golden_sam("default")
    finally:
        h.close(); base.close()
