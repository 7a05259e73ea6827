"""The `lordfast` front end (lordfast_amd/cli/lordfast.c): the reference's process interface (SURVEY 8b,
src/CommandLineParser.cpp:126-310, src/baseFAST.cpp:30-95) over the C ABI.  Option handling is checked on CPU
(it fails before any device call); the end-to-end run is a GPU test against the golden SAM of the compiled reference."""
import gzip
import os
import subprocess

import pytest

from conftest import GOLDEN, golden_sam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "lordfast_amd", "lordfast")


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", os.path.join(ROOT, "lordfast_amd", "csrc"), "-j4"], check=True, stdout=subprocess.DEVNULL)
    return CLI


def run(cli, *args, env=None):
    return subprocess.run([cli, *args], capture_output=True, timeout=600, env=env)


def test_usage_and_validation(cli):
    r = run(cli)
    assert r.returncode == 1 and b"usage: lordfast --index ref.fa" in r.stderr
    r = run(cli, "-v")
    assert r.returncode == 0 and r.stdout == b"lordFAST 0.0.10\n"
    r = run(cli, "--seq", "x.fa")
    assert r.returncode == 1 and b"indexing / searching mode should be selected" in r.stderr
    r = run(cli, "--search", "x.fa")
    assert r.returncode == 1 and b"please indicate a sequence file for searching" in r.stderr
    for opt, msg in (("-k 11", b"-k/--minAnchorLen requires an argument in [12..20]"), ("-k 21", b"-k/--minAnchorLen"),
                     ("-c 0", b"-c/--anchorCount requires a positive integer"), ("-n 0", b"-n/--numMap requires a positive integer"),
                     ("-m 0", b"-m/--maxRefHit requires a positive integer")):
        r = run(cli, "--search", "x.fa", "--seq", "y.fa", *opt.split())
        assert r.returncode == 1 and msg in r.stderr, opt
    r = run(cli, "--search", "x.fa", "--seq", "y.fa", "-R", "RG\\tID:a")
    assert r.returncode == 1 and b"does not start with @RG" in r.stderr
    r = run(cli, "--search", "x.fa", "--seq", "y.fa", "-R", "@RG\\tSM:a")
    assert r.returncode == 1 and b"no ID within the read group line" in r.stderr
    r = run(cli, "--search", "x.fa", "--seq", "y.fa", "-Z")
    assert r.returncode == 1 and b"usage:" in r.stderr


def test_no_device_is_loud(cli):
    import lordfast_amd as la
    if la.device_count() > 0:
        pytest.skip("a device is visible")
    r = run(cli, "--search", "x.fa", "--seq", "y.fa")
    assert r.returncode == 1 and b"no gfx950 device" in r.stderr and b"no CPU path" in r.stderr


@pytest.mark.gpu
def test_search_end_to_end(cli, golden_dir, oracle, oracle_lib, tmp_path):
    reads = str(tmp_path / "reads.fa")
    with gzip.open(os.path.join(GOLDEN, "reads.fa.gz"), "rb") as fi, open(reads, "wb") as fo:
        fo.write(fi.read())
    fa = os.path.join(golden_dir, "genome.fa")
    out = str(tmp_path / "out.sam")
    args = ["--search", fa, "--seq", reads, "-o", out, "-t", "4"]
    r = run(cli, *args)
    assert r.returncode == 0, r.stderr.decode()
    cmdline = " ".join([cli] + args) + " "
    exp = oracle.sam_header(cmdline) + golden_sam("default")
    assert open(out, "rb").read() == exp
    # stdout, no header, clasp + -n 30, read group
    args = ["--search", fa, "--seq", os.path.join(GOLDEN, "reads.fa.gz"), "--noSamHeader", "-a", "clasp", "-n", "30", "-t", "0"]
    r = run(cli, *args)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == golden_sam("clasp_n30")
    args = ["--search", fa, "--seq", reads, "-R", "@RG\\tID:grp1\\tSM:x", "-k", "17", "-c", "2000"]
    r = run(cli, *args)
    assert r.returncode == 0, r.stderr.decode()
    p = oracle_lib.default_params(min_anchor_len=17, sampling_count=2000, read_group_id=b"grp1", read_group=b"@RG\tID:grp1\tSM:x")
    names, seqs = [], []
    from conftest import read_fasta
    names, seqs = read_fasta(reads)
    exp = oracle.sam_header(" ".join([cli] + args) + " ", params=p) + oracle.map_batch(names, seqs, params=p)
    assert r.stdout == exp
    assert b"@RG\tID:grp1\tSM:x\n@PG" in r.stdout and b"\tRG:Z:grp1" in r.stdout
    # LF_DEVICES: the batches are spread over several index replicas (here two on device 0); same records, same order
    args = ["--search", fa, "--seq", reads, "--noSamHeader", "-t", "4"]
    r = run(cli, *args, env=dict(os.environ, LF_DEVICES="0,0", LF_CHUNK_READS="9"))
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == golden_sam("default")
    r = run(cli, *args, env=dict(os.environ, LF_DEVICES="0-99"))
    assert r.returncode != 0 and b"no gfx950 device" in r.stderr


@pytest.mark.gpu
def test_index_mode_builds_reference_files(cli, tmp_path):
    """--index writes the reference's index files; --search on a FASTA without index builds it first (src/BWT.cpp:203-208)"""
    fa = str(tmp_path / "genome.fa")
    with gzip.open(os.path.join(GOLDEN, "genome.fa.gz"), "rb") as fi, open(fa, "wb") as fo:
        fo.write(fi.read())
    r = run(cli, "--index", fa)
    assert r.returncode == 0, r.stderr.decode()
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert open(f"{fa}.{ext}", "rb").read() == open(os.path.join(GOLDEN, f"genome.fa.{ext}"), "rb").read(), ext


@pytest.mark.gpu
def test_output_appended_to_a_file_stays_in_order(cli, golden_dir, tmp_path):
    """`lordfast ... >> out.sam`: on an O_APPEND descriptor pwrite ignores its offset, so the writer must not cut a batch's text
    into concurrently written pieces there (the SAM would come out in completion order).  > 8 MB of records, the size from which
    a regular file is written in four pieces."""
    from conftest import read_fasta
    src = str(tmp_path / "src.fa")
    with gzip.open(os.path.join(GOLDEN, "reads.fa.gz"), "rb") as fi, open(src, "wb") as fo:
        fo.write(fi.read())
    names, seqs = read_fasta(src)
    reads = str(tmp_path / "many.fa")
    with open(reads, "wb") as fo:
        for k in range(24):
            for n, s in zip(names, seqs):
                fo.write(b">" + n + b"_c%d\n" % k + s + b"\n")
    fa = os.path.join(golden_dir, "genome.fa")
    plain = str(tmp_path / "plain.sam")
    r = run(cli, "--search", fa, "--seq", reads, "--noSamHeader", "-o", plain, "-t", "4")
    assert r.returncode == 0, r.stderr.decode()
    want = open(plain, "rb").read()
    assert len(want) > (8 << 20) and want.count(b"\n") >= 24 * len(names)
    appended = str(tmp_path / "appended.sam")
    with open(appended, "wb") as fo:
        fo.write(b"@CO\tsomething that was there before\n")
    with open(appended, "ab") as fo:
        r = subprocess.run([cli, "--search", fa, "--seq", reads, "--noSamHeader", "-t", "4"], stdout=fo, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    assert open(appended, "rb").read() == b"@CO\tsomething that was there before\n" + want
