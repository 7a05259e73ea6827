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


# ---- the same shapes on a HUMAN-LIKE genome: 120 Mbp, about half of it repeats, incl. a 300 bp family with > 10^4 copies
# ---- (Alu-like) and truncated 1-6 kbp families: seeds with hundreds of hits, fine mode, big chain windows.  ALL records
# ---- (primary, secondary, supplementary) are compared, with the oracle and -- where it was built -- the compiled reference.

@pytest.fixture(scope="module")
def human_like(tmp_path_factory):
    import lordfast_amd as la
    d = tmp_path_factory.mktemp("humanlike")
    g = synth.make_genome(120_000_000, 5, seed=11, n_families=200, profile="grch38like")
    places = synth.add_duplications(g, seg_len=40000, copies=4, div=0.02)          # segmental duplications: fine mode
    fa = la.index_build(g, os.path.join(str(d), "human_like.fa"))
    return fa, g, places


def _run_big(fa, reads, oracle_lib, **kw):
    import lordfast_amd as la
    from conftest import have_ref
    names = [r[0].encode() for r in reads]
    seqs = [r[1] for r in reads]
    h = la.LordFast(fa, device=0)
    sam, st = h.map_batch(names, seqs, params=la.default_params(**kw))
    h.close()
    orc = oracle_lib.Oracle(fa)
    exp = orc.map_batch(names, seqs, params=oracle_lib.default_params(threads=16, **kw))
    orc.close()
    assert sam == exp, first_diff(sam, exp)
    if have_ref() and os.path.exists(fa + ".cache"):
        ref = oracle_lib.Ref()
        ref.load(fa)
        ref.set_params(oracle_lib.default_params(threads=1, **kw), "test")
        rsam, _ = ref.map_mem(names, seqs)
        assert sam == rsam, first_diff(sam, rsam)
    return st, sam


def test_c2_shape_on_a_repeat_rich_genome(human_like, oracle_lib):
    import numpy as np
    fa, g, places = human_like
    reads = synth.make_reads(g, 300, 15000, 0.15, seed=2024)
    rng = np.random.default_rng(3)
    for k, (ci, p) in enumerate(places * 4):                                       # reads out of the duplicated segment
        frag = g[ci][1][p + 2000 * (k % 5):p + 2000 * (k % 5) + 14000]
        if k % 2:
            frag = synth.revcomp(frag)
        reads.append((f"dup{k}", synth.mutate(frag, 0.15, rng).tobytes()))
    st, sam = _run_big(fa, reads, oracle_lib)
    flags = [int(l.split(b"\t")[1]) for l in sam.split(b"\n") if l]
    assert st["n_seeds"] / len(reads) > 600
    assert sum(1 for f in flags if f & 256) > 0, "fine mode with secondaries must occur"


def test_c4_shape_on_a_repeat_rich_genome(human_like, oracle_lib):
    fa, g, _ = human_like
    reads = synth.make_reads(g, 120, 15000, 0.15, seed=9)
    _run_big(fa, reads, oracle_lib, max_map=30, chain_alg=1)


def test_c5_shape_on_a_repeat_rich_genome(human_like, oracle_lib):
    fa, g, _ = human_like
    reads = synth.make_reads(g, 30, 50000, 0.10, seed=5, mix=(0.40, 0.25, 0.35))
    _run_big(fa, reads, oracle_lib, min_anchor_len=17, sampling_count=2000)


# ---- BASELINE config C4 as it is benchmarked (bench.py --config c4): reads out of 2-4-copy segmental duplications, --chainAlg
# ---- clasp, -n 30: the fine branch of mapSeq aligns every near-equal candidate window.  Many small contigs, so that candidate
# ---- windows at contig borders occur whose seeds all lie in the neighbouring contig: chain_seeds_clasp then leaves its previous
# ---- chain in place and the reference extends that stale chain (src/Chain.cpp:68,92) -- restated, not avoided.
@pytest.fixture(scope="module")
def segdup_genome(tmp_path_factory):
    import lordfast_amd as la
    d = tmp_path_factory.mktemp("segdup")
    g = synth.make_genome(24_000_000, 24, seed=11, repeat_frac=0.10, n_families=100)
    fams = synth.add_segdups(g, 120, seg_len=(12000, 20000), seed=7)
    fa = la.index_build(g, os.path.join(str(d), "segdup.fa"))
    return fa, g, fams


def test_c4_segdup_workload_clasp_n30(segdup_genome, oracle_lib):
    fa, g, fams = segdup_genome
    reads = synth.make_reads(g, 700, 15000, 0.15, seed=2024, segdups=fams, dup_frac=0.5)
    st, sam = _run_big(fa, reads, oracle_lib, max_map=30, chain_alg=1)
    assert st["n_chain_problems"] / len(reads) > 1.5, "the duplicated reads must reach the fine branch"
    # the one corner of -a clasp where the reference depends on what its thread mapped before (a read's FIRST window without seeds: it extends the previous
    # read's chain, src/Chain.cpp:68,92) is counted, not restated: the comparison above is only meaningful while this workload does not hit it
    assert st["n_stale_first_windows"] == 0, st["n_stale_first_windows"]
    flags = [int(l.split(b"\t")[1]) for l in sam.split(b"\n") if l]
    assert sum(1 for f in flags if f & 256) > 100
    st, _ = _run_big(fa, reads[:300], oracle_lib, max_map=30)                        # the same reads through dp-n2
