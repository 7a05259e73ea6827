"""The reference-side binding of INTEGRATION.md section 2, for real: oracle/_ref/lordfast_shim is the reference's OWN
driver (src/baseFAST.cpp main, CommandLineParser, Reads, Common -- compiled from /root/reference by `make -C oracle
shim`) linked with lordfast_amd/integration/lf_shim.cpp against liblfgpu.so in place of the reference's mapper objects.

CPU: it links, parses options like the reference, and refuses to map without a device (no CPU fallback).
GPU: the reference's main() drives bwt_load -> initializeFAST -> initFASTChunk -> mapSeqMT -> finalizeFAST
(src/baseFAST.cpp:44-81) and must print the golden SAM of the compiled reference."""
import gzip
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, golden_sam

SHIM = os.path.join(ROOT, "oracle", "_ref", "lordfast_shim")


@pytest.fixture(scope="module")
def shim():
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "shim"], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(SHIM):
        pytest.skip("oracle/_ref/lordfast_shim not built (needs /root/reference)")
    return SHIM


@pytest.fixture()
def reads_fa(tmp_path):
    p = tmp_path / "reads.fa"
    with gzip.open(os.path.join(GOLDEN, "reads.fa.gz"), "rb") as fi:
        p.write_bytes(fi.read())
    return str(p)


def test_shim_links_and_keeps_the_reference_cli(shim):
    r = subprocess.run([shim, "-v"], capture_output=True, text=True)
    assert r.returncode == 0 and "lordFAST 0.0.10" in r.stdout
    r = subprocess.run([shim, "-k", "5", "--search", "x.fa", "--seq", "y.fa"], capture_output=True, text=True)
    assert r.returncode != 0 and "requires an argument in [12..20]" in r.stderr
    # every mapper symbol the driver objects reference is resolved by the shim / liblfgpu.so, none by reference objects
    nm = subprocess.run(["nm", "-D", "--undefined-only", shim], capture_output=True, text=True).stdout
    for sym in ("bwt_load", "bwt_index", "initializeFAST", "initFASTChunk", "mapSeqMT", "finalizeFAST", "lf_global_params"):
        assert sym in nm, sym


def test_shim_without_a_device_fails_loudly(shim, golden_dir, reads_fa, tmp_path):
    import lordfast_amd as la
    if la.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([shim, "--search", os.path.join(golden_dir, "genome.fa"), "--seq", reads_fa, "-o", str(tmp_path / "o.sam")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "no gfx950 device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("opts,cfg", [([], "default"), (["-n", "30"], "n30"), (["-a", "clasp"], "clasp"),
                                      (["-k", "17", "-c", "2000"], "k17c2000")])
def test_shim_prints_the_golden_sam(shim, golden_dir, reads_fa, tmp_path, opts, cfg):
    out = str(tmp_path / "o.sam")
    cmd = [shim, "--search", os.path.join(golden_dir, "genome.fa"), "--seq", reads_fa, "-o", out, "-t", "4"] + opts
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mapping... done in" in r.stderr                       # the reference's own progress line (src/baseFAST.cpp:71-75)
    txt = open(out, "rb").read()
    head = [l for l in txt.split(b"\n") if l.startswith(b"@")]
    body = b"".join(l + b"\n" for l in txt.split(b"\n") if l and not l.startswith(b"@"))
    assert head[0] == b"@HD\tVN:1.5\tSO:unsorted" and head[-1].startswith(b"@PG\tID:lordfast\tPN:lordfast\tVN:0.0.10\tCL:")
    assert body == golden_sam(cfg)
