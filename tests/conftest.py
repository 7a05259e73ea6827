"""pytest configuration: `gpu` marker + shared fixtures.

`-m "not gpu"`: oracle vs golden vectors / vs the compiled reference, host logic, C-ABI symbol checks.
`-m gpu`     : parity of the HIP path (through the C-ABI) with the oracle and the golden vectors.
"""
import gzip
import os

import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_fasta(path):
    names, seqs = [], []
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as fh:
        name, chunks = None, []
        for line in fh:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    names.append(name); seqs.append(b"".join(chunks))
                name, chunks = line[1:].split()[0], []
            elif line:
                chunks.append(line)
        if name is not None:
            names.append(name); seqs.append(b"".join(chunks))
    return names, seqs


@pytest.fixture(scope="session")
def golden_dir(tmp_path_factory):
    """A scratch copy of the golden index (genome.fa + .bwt/.sa/.pac/.ann/.amb; no .cache)."""
    d = tmp_path_factory.mktemp("golden")
    for f in os.listdir(GOLDEN):
        if f.startswith("genome.fa."):
            if f.endswith(".gz"):
                with gzip.open(os.path.join(GOLDEN, f), "rb") as fi, open(d / "genome.fa", "wb") as fo:
                    shutil.copyfileobj(fi, fo)
            else:
                shutil.copy(os.path.join(GOLDEN, f), d / f)
    return str(d)


@pytest.fixture(scope="session")
def golden_reads():
    return read_fasta(os.path.join(GOLDEN, "reads.fa.gz"))


@pytest.fixture(scope="session")
def stages():
    return np.load(os.path.join(GOLDEN, "stages.npz"))


@pytest.fixture(scope="session")
def stages_clasp():
    return np.load(os.path.join(GOLDEN, "stages_clasp.npz"))


def golden_sam(cfg):
    with gzip.open(os.path.join(GOLDEN, f"expected_{cfg}.sam.gz"), "rb") as fh:
        return fh.read()


GOLDEN_CONFIGS = {
    "default": dict(),
    "n30": dict(max_map=30),
    "k17c2000": dict(min_anchor_len=17, sampling_count=2000),
    "k12c300m20": dict(min_anchor_len=12, sampling_count=300, max_ref_hits=20),
    "clasp": dict(chain_alg=1),
    "clasp_n30": dict(chain_alg=1, max_map=30),
}


def split_ragged(flat, counts):
    out, o = [], 0
    for c in counts:
        out.append(flat[o:o + int(c)])
        o += int(c)
    return out


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle as po
    po.build()
    return po


@pytest.fixture(scope="session")
def oracle(oracle_lib, golden_dir):
    return oracle_lib.Oracle(os.path.join(golden_dir, "genome.fa"))


def have_ref():
    from oracle import pyoracle as po
    return os.path.exists(po.REF_SO)


@pytest.fixture(scope="session")
def ref(oracle_lib):
    if not have_ref():
        pytest.skip("oracle/_ref/liblfref.so not built (needs /root/reference)")
    return oracle_lib.Ref()


@pytest.fixture(autouse=True)
def _gpu_watchdog(request, monkeypatch):
    """GPU tests only: a batch that hangs becomes an abort that names where every lane is (lf_pipeline.c) instead of a stuck
    run.  Scoped to the test (monkeypatch), generous (a slow box must not trip it), and not inherited by the subprocesses of
    the CPU tests."""
    if request.node.get_closest_marker("gpu") is not None and "LF_WATCHDOG" not in os.environ:
        monkeypatch.setenv("LF_WATCHDOG", "600")
    yield


@pytest.fixture(autouse=True)
def _gpu_streams_drain(request):
    """After every GPU test all lane streams of device 0 must be idle: a kernel left running would only show up later, as a
    hang inside some hipMalloc / hipFree (lfg_drain_check names the stream and aborts after LF_WATCHDOG seconds)."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import lordfast_amd.api as api
        api.lib().lfg_drain_check(0)
