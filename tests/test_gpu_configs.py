"""GPU parity at the SHAPES of BASELINE.json's configs (smaller counts, oracle-checkable in seconds):
C1  1 k reads ~5 kbp / 12 % error vs an E. coli-size (4.6 Mbp) genome, default options;
C4  --chainAlg clasp with -n 30 multi-candidate extension (and -n 30 with dp-n2);
C5  ONT-profile reads ~50 kbp / 10 % error, -k 17 -c 2000."""
import os

import pytest

from lordfast_amd import synth
from test_gpu_map import first_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ecoli_like(tmp_path_factory):
    import lordfast_amd as la
    d = tmp_path_factory.mktemp("ecoli")
    g = synth.make_genome(4600000, 1, seed=101, n_families=40, repeat_frac=0.03)
    fa = la.index_build(g, os.path.join(str(d), "ecoli_like.fa"))
    return fa, g


def _run(fa, reads, oracle_lib, **kw):
    import lordfast_amd as la
    names = [r[0].encode() for r in reads]
    seqs = [r[1] for r in reads]
    h = la.LordFast(fa, device=0)
    sam, st = h.map_batch(names, seqs, params=la.default_params(**kw))
    h.close()
    orc = oracle_lib.Oracle(fa)
    exp = orc.map_batch(names, seqs, params=oracle_lib.default_params(threads=8, **kw))
    orc.close()
    assert sam == exp, first_diff(sam, exp)
    return st


def test_c1_shape(ecoli_like, oracle_lib):
    fa, g = ecoli_like
    st = _run(fa, synth.make_reads(g, 1000, 5000, 0.12, seed=2024), oracle_lib)
    assert st["n_reads"] == 1000


def test_c4_shape_clasp_n30(ecoli_like, oracle_lib):
    fa, g = ecoli_like
    reads = synth.make_reads(g, 150, 15000, 0.15, seed=7)
    _run(fa, reads, oracle_lib, max_map=30)
    st = _run(fa, reads, oracle_lib, max_map=30, chain_alg=1)
    assert st["n_chain_problems"] >= 150


def test_c5_shape(ecoli_like, oracle_lib):
    fa, g = ecoli_like
    reads = synth.make_reads(g, 40, 50000, 0.10, seed=5, mix=(0.40, 0.25, 0.35))
    _run(fa, reads, oracle_lib, min_anchor_len=17, sampling_count=2000)
