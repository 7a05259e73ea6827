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
