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
