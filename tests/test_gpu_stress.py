"""Concurrency stress: many tiny chunks through eight lanes + helper threads, repeatedly, with different thread counts.
Every run must print the golden records (races in the chunk pipeline would show up as occasional differences)."""
import os

import pytest

from conftest import GOLDEN_CONFIGS, golden_sam

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("threads,chunk,lanes", [(16, 5, 8), (16, 3, 8), (9, 4, 4), (5, 11, 2), (2, 6, 1)])
def test_many_tiny_chunks_repeatedly(golden_dir, golden_reads, monkeypatch, threads, chunk, lanes):
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_CHUNK_READS", str(chunk))
    monkeypatch.setenv("LF_LANES", str(lanes))
    h = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
    exp = golden_sam("default")
    try:
        for it in range(6):
            sam, st = h.map_batch(names, seqs, params=la.default_params(threads=threads, **GOLDEN_CONFIGS["default"]))
            assert sam == exp, f"iteration {it}: output differs"
    finally:
        h.close()


@pytest.mark.parametrize("n_threads,chunk", [(4, 7), (3, 58)])
def test_concurrent_batches_share_the_lanes(golden_dir, golden_reads, monkeypatch, n_threads, chunk):
    """several threads call the library at once (a rank that keeps a few small shards in flight): the lane allocator hands the
    device's lane ids out across the calls; every call must still return the golden records -- under different option sets"""
    import threading
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_CHUNK_READS", str(chunk))
    h = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
    cfgs = ["default", "n30", "clasp", "k17c2000"]
    errs = []

    def work(t):
        try:
            for it in range(5):
                cfg = cfgs[(t + it) % len(cfgs)]
                sam, _ = h.map_batch(names, seqs, params=la.default_params(threads=8, **GOLDEN_CONFIGS[cfg]))
                if sam != golden_sam(cfg):
                    errs.append((t, it, cfg))
        except Exception as e:                                   # noqa: BLE001
            errs.append((t, repr(e)))
    th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    try:
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
    finally:
        h.close()
