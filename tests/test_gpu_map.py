"""GPU parity, whole path: lf_map_batch (HIP kernels + host glue, through the C ABI) must print the very
SAM records the reference prints (golden vectors), and the oracle's on larger seeded inputs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_CONFIGS, golden_sam
from lordfast_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _product_path_after_each_test():
    """the cross-check selections (lf_debug_crosscheck) never leak into the next test"""
    yield
    import lordfast_amd as la
    la.lib().lf_debug_crosscheck(0)


@pytest.fixture(scope="module")
def lf(golden_dir):
    import lordfast_amd as la
    h = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
    yield h
    h.close()


def first_diff(a: bytes, b: bytes) -> str:
    la_, lb = a.split(b"\n"), b.split(b"\n")
    for i, (x, y) in enumerate(zip(la_, lb)):
        if x != y:
            fx, fy = x.split(b"\t"), y.split(b"\t")
            for j, (u, v) in enumerate(zip(fx, fy)):
                if u != v:
                    return f"line {i} ({fx[0].decode()}) field {j}: {u[:80]!r} != {v[:80]!r}"
            return f"line {i}: field count {len(fx)} vs {len(fy)}"
    return f"line count {len(la_)} vs {len(lb)}"


@pytest.mark.parametrize("cfg", list(GOLDEN_CONFIGS))
def test_sam_golden(lf, golden_reads, cfg):
    import lordfast_amd as la
    names, seqs = golden_reads
    sam, st = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    exp = golden_sam(cfg)
    assert sam == exp, first_diff(sam, exp)
    assert st["n_reads"] == len(seqs) and st["n_edlib_problems"] > 0 and st["n_chain_problems"] > 0
    assert st["n_ksw_problems"] > 0, "fixture must reach the ksw clip/split branch"
    print("tie requests:", st["n_tie_requests"], "of", st["n_chain_problems"])


@pytest.mark.parametrize("cfg", ["default", "clasp"])
def test_sam_golden_with_the_large_request_sort(lf, golden_reads, cfg, monkeypatch):
    """requests above 8192 seeds (a candidate window over a satellite array) are sorted by lf_req_sort_big_kernel (a bitonic network
    over a scratch array in HBM; no library sort on the mapping path).  LF_REQ_SORT_BIG_FROM lowers the bound so that the golden
    reads' requests (a few hundred seeds) take that kernel: records must not change, neither by qPos (dp-n2) nor by tPos (clasp)"""
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_REQ_SORT_BIG_FROM", "65")
    sam, st = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    exp = golden_sam(cfg)
    assert sam == exp, first_diff(sam, exp)


def test_satellite_array_reads_take_the_large_request_sort(tmp_path, oracle_lib):
    """a genome with a tandem satellite array (171 bp monomers, 27 kbp): a read out of the array hits it with every sample, so its
    candidate window's request holds thousands of seeds; records == oracle"""
    import numpy as np
    import lordfast_amd as la
    from lordfast_amd import synth
    rng = np.random.default_rng(5)
    contigs = synth.make_genome(1_500_000, 2, seed=3, repeat_frac=0.0, n_families=0)
    mono = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=171)]
    arr = np.tile(mono, 160)
    mut = rng.random(len(arr)) < 0.01
    arr[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(mut.sum()))]
    s0 = contigs[0][1]
    s0[300000:300000 + len(arr)] = arr
    fa = str(tmp_path / "sat.fa")
    la.index_build(contigs, fa)
    reads = [(f"sat{i}", synth.mutate(s0[300000 + 2000 * i + 500:300000 + 2000 * i + 7500].copy(), 0.10, rng).tobytes()) for i in range(6)]
    reads += synth.make_reads(contigs, 20, 6000, 0.12, seed=9)
    names, seqs = [r[0].encode() for r in reads], [r[1] for r in reads]
    prm = dict()                                         # (a request of n seeds costs the oracle's dp-n2 chainer n^2 / 2 steps: ~10^8 per satellite read)
    lf = la.LordFast(fa, device=0)
    try:
        sam, st = lf.map_batch(names, seqs, params=la.default_params(**prm))
    finally:
        lf.close()
    po = oracle_lib
    exp = po.Oracle(fa).map_batch(names, seqs, params=po.default_params(**prm))
    assert sam == exp, first_diff(sam, exp)
    assert st["n_req_seeds"] > 6 * 8192, "the satellite reads' requests are not above the LDS sort's 8192 seeds"


def test_sam_host_cigar_crosscheck(lf, golden_reads, monkeypatch):
    """lf_debug_crosscheck(2) builds CIGAR / MD on the host from copied-back edit paths instead of lf_render_kernel: both
    must print the reference's records"""
    import lordfast_amd as la
    names, seqs = golden_reads
    la.lib().lf_debug_crosscheck(2)
    sam, st = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS["default"]))
    assert st["render_launches"] == 0
    exp = golden_sam("default")
    assert sam == exp, first_diff(sam, exp)
    la.lib().lf_debug_crosscheck(0)
    sam, st = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS["default"]))
    assert st["render_launches"] >= 1 and st["render_bytes"] > 0
    assert sam == exp, first_diff(sam, exp)


@pytest.mark.parametrize("cfg", ["default", "n30"])
def test_sam_host_vote_crosscheck(lf, golden_reads, monkeypatch, cfg):
    """lf_debug_crosscheck(1) votes / selects / sorts on the host from copied-back hits instead of lf_vote.hip: same records"""
    import lordfast_amd as la
    names, seqs = golden_reads
    la.lib().lf_debug_crosscheck(1)
    sam, st = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    exp = golden_sam(cfg)
    assert sam == exp, first_diff(sam, exp)


def test_map_batch_into_caller_buffer(lf, golden_reads):
    """lf_map_batch_into: same records into a caller-owned buffer; a buffer that is too small is an error, not a truncation"""
    import ctypes as C
    import lordfast_amd as la
    names, seqs = golden_reads
    exp = golden_sam("default")
    buf = np.zeros(len(exp) + 4096, dtype=np.uint8)
    ln, st = lf.map_batch_into(names, seqs, buf.ctypes.data, buf.size, params=la.default_params(**GOLDEN_CONFIGS["default"]))
    assert ln == len(exp) and buf[:ln].tobytes() == exp
    ln2, _ = lf.map_batch_into(names, seqs, buf.ctypes.data, buf.size)            # buffer reuse
    assert buf[:ln2].tobytes() == exp
    buf[:] = 0                                                                    # lf_map_batch_into_lens: lengths supplied
    ln3, _ = lf.map_batch_into(names, seqs, buf.ctypes.data, buf.size, seq_lens=np.array([len(x) for x in seqs], dtype=np.uint32))
    assert ln3 == len(exp) and buf[:ln3].tobytes() == exp
    small = np.zeros(len(exp) // 2, dtype=np.uint8)
    with pytest.raises(RuntimeError, match="too small"):
        lf.map_batch_into(names, seqs, small.ctypes.data, small.size)
    sam, _ = lf.map_batch(names, seqs)                                           # the handle survives the error
    assert sam == exp


@pytest.mark.parametrize("threads", [1, 3])
def test_sam_thread_counts(lf, golden_reads, threads):
    import lordfast_amd as la
    names, seqs = golden_reads
    sam, _ = lf.map_batch(names, seqs, params=la.default_params(threads=threads))
    assert sam == golden_sam("default")


def test_sam_many_chunks(lf, golden_reads, monkeypatch):
    """reads are processed in chunks; chunk boundaries must not show in the output"""
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_CHUNK_READS", "7")
    sam, st = lf.map_batch(names, seqs, params=la.default_params(threads=2))
    assert sam == golden_sam("default")


@pytest.mark.parametrize("lanes,chunk", [(12, 3), (5, 11), (2, 1)])
def test_sam_many_chunks_into_caller_buffer(lf, golden_reads, monkeypatch, lanes, chunk):
    """A caller-owned output buffer: the chunks' SAM texts are copied out asynchronously, out of order, from two device
    buffers per lane; a chunk may stay pending until the chunks of earlier reads have published their sizes.  Many tiny
    chunks on many lanes must still give the one text."""
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_CHUNK_READS", str(chunk))
    monkeypatch.setenv("LF_LANES", str(lanes))
    buf = np.full(64 << 20, 0x7f, dtype=np.uint8)
    for rep in range(2):
        ln, st = lf.map_batch_into(names, seqs, buf.ctypes.data, buf.size, params=la.default_params(threads=16, **GOLDEN_CONFIGS["default"]))
        assert bytes(buf[:ln]) == golden_sam("default")


def test_sam_fastq_and_readgroup(lf, oracle, oracle_lib, golden_reads):
    import lordfast_amd as la
    names, seqs = golden_reads
    names, seqs = names[:30], seqs[:30]
    rng = np.random.default_rng(5)
    quals = [bytes(rng.integers(35, 74, size=len(s)).astype(np.uint8)) for s in seqs]
    sam, _ = lf.map_batch(names, seqs, quals, params=la.default_params(read_group_id=b"grp1"))
    exp = oracle.map_batch(names, seqs, quals, params=oracle_lib.default_params(read_group_id=b"grp1"))
    assert sam == exp, first_diff(sam, exp)


def test_empty_and_tiny(lf, oracle):
    sam, _ = lf.map_batch([], [])
    assert sam == b""
    names = [b"a", b"b"]
    seqs = [b"ACGT", b"N" * 1500]
    sam, _ = lf.map_batch(names, seqs)
    assert sam == oracle.map_batch(names, seqs)


@pytest.fixture(scope="module")
def big_case(tmp_path_factory, oracle_lib):
    """1.2 Mbp genome with repeat families; index built by the compiled reference (checker side)"""
    from conftest import have_ref
    if not have_ref():
        pytest.skip("needs oracle/_ref/liblfref.so to build an index")
    d = tmp_path_factory.mktemp("g1m")
    g = synth.make_genome(1200000, 4, seed=31, n_families=60, repeat_frac=0.12)
    dup = synth.add_duplications(g, 15000, 4, 0.02)
    fa = os.path.join(str(d), "g.fa")
    synth.write_fasta(fa, g)
    ref = oracle_lib.Ref()
    ref.index_build(fa)
    os.remove(fa + ".cache")
    reads = synth.special_reads(g, dup, seed=9) + synth.make_reads(g, 300, 6000, 0.15, seed=10) \
        + synth.make_reads(g, 60, 12000, 0.15, seed=11) + synth.make_reads(g, 40, 3000, 0.10, seed=12, mix=(0.40, 0.25, 0.35))
    return fa, reads


@pytest.mark.parametrize("kw", [dict(), dict(max_map=30), dict(max_map=100), dict(min_anchor_len=17, sampling_count=2000),
                                dict(chain_alg=1), dict(chain_alg=1, max_map=30, min_anchor_len=12, sampling_count=500)])
def test_sam_vs_oracle_big(big_case, oracle_lib, kw):
    import lordfast_amd as la
    fa, reads = big_case
    names = [r[0] for r in reads]
    seqs = [r[1] for r in reads]
    h = la.LordFast(fa, device=0)
    sam, st = h.map_batch(names, seqs, params=la.default_params(**kw))
    h.close()
    orc = oracle_lib.Oracle(fa)
    exp = orc.map_batch(names, seqs, params=oracle_lib.default_params(threads=8, **kw))
    orc.close()
    assert sam == exp, first_diff(sam, exp)


def test_sam_tandem_repeats_tie_order(tmp_path, oracle_lib):
    """Tandem duplications make one read sample hit two places of the same candidate window: equal qPos inside a chain
    request.  std::sort's (unstable) order of those ties reaches the chain DP, so the GPU path must replay libstdc++'s
    introsort for such requests (lf_tie_sort_kernel) -- checked against the oracle, and the replay must really run."""
    import lordfast_amd as la
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs = []
    for ci in range(2):
        parts = []
        for _ in range(12):
            parts.append(acgt[rng.integers(0, 4, size=int(rng.integers(6000, 9000)))])
            unit = acgt[rng.integers(0, 4, size=int(rng.integers(400, 1500)))]
            copies = int(rng.integers(2, 4))
            for k in range(copies):
                u = unit.copy()
                nmut = max(1, len(u) // 200)                       # copies differ a little, like real tandem repeats
                u[rng.integers(0, len(u), size=nmut)] = acgt[rng.integers(0, 4, size=nmut)]
                parts.append(u)
        parts.append(acgt[rng.integers(0, 4, size=7000)])
        contigs.append((f"tr{ci}", np.concatenate(parts)))
    fa = la.index_build(contigs, str(tmp_path / "tandem.fa"))
    reads = synth.make_reads(contigs, 120, 5000, 0.12, seed=3) + synth.make_reads(contigs, 40, 9000, 0.15, seed=4)
    names = [r[0].encode() for r in reads]
    seqs = [r[1] for r in reads]
    h = la.LordFast(fa, device=0)
    sam, st = h.map_batch(names, seqs)
    h.close()
    orc = oracle_lib.Oracle(fa)
    exp = orc.map_batch(names, seqs, params=oracle_lib.default_params(threads=8))
    orc.close()
    assert st["n_tie_requests"] > 0, "the fixture must produce chain requests with equal qPos"
    assert sam == exp, first_diff(sam, exp)


def test_reads_with_junk_ends_and_insertions_replay_in_one_round(tmp_path, oracle_lib, monkeypatch):
    """Reads with random sequence in front, behind and in the middle: alignChain_edlib's clip test fires at both ends and its split test inside
    (src/LordFAST.cpp:1840-1867, 1952-2107, 2172-2199), so a chain needs several ksw_extend results.  The host replay (lf_replay.c) asks for all of
    them in ONE round; the records equal the oracle's, and equal what one request per round (round 4's behaviour, LF_KSW_ONE_PER_ROUND=1) gives."""
    import lordfast_amd as la
    g = synth.make_genome(900000, 3, seed=41, n_families=40, repeat_frac=0.08)
    fa = la.index_build(g, str(tmp_path / "g.fa"))
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    junk = lambda n: acgt[rng.integers(0, 4, size=int(n))].tobytes()
    base = synth.make_reads(g, 48, 7000, 0.10, seed=21, sigma=0.15)
    names, seqs = [], []
    for i, (nm, sq) in enumerate(base):
        kind = i % 4
        if kind == 0:   sq = junk(rng.integers(700, 2500)) + sq + junk(rng.integers(700, 2500))                    # clip tests at both ends
        elif kind == 1: h = len(sq) // 2; sq = sq[:h] + junk(rng.integers(300, 1500)) + sq[h:]                       # split test in the middle
        elif kind == 2: h = len(sq) // 3; sq = junk(900) + sq[:h] + junk(600) + sq[h:2 * h] + junk(400) + sq[2 * h:] + junk(1200)   # all of them
        names.append(nm.encode()); seqs.append(sq)
    orc = oracle_lib.Oracle(fa)
    exp = orc.map_batch(names, seqs, params=oracle_lib.default_params(threads=8))
    orc.close()
    h = la.LordFast(fa, device=0)
    sam, st = h.map_batch(names, seqs)
    assert st["n_ksw_problems"] >= 40, "the fixture must trigger clip / split tests"
    assert sam == exp, first_diff(sam, exp)
    monkeypatch.setenv("LF_KSW_ONE_PER_ROUND", "1")
    sam1, st1 = h.map_batch(names, seqs)
    h.close()
    assert sam1 == exp and st1["n_ksw_problems"] == st["n_ksw_problems"]
    assert st1["n_host_waits"] > st["n_host_waits"], "one request per round must take more rounds (more waits) than all requests at once"


def test_map_file_fasta_gz(lf, golden_dir, tmp_path):
    """lf_map_file: gzip FASTA in (the reference's reader grammar), SAM with header out, batches read ahead of the GPU"""
    import lordfast_amd as la
    out = str(tmp_path / "out.sam")
    reads = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reads.fa.gz")
    p = la.default_params(**GOLDEN_CONFIGS["default"])
    st = lf.map_file(reads, out, params=p, cmdline="lordfast --search x", batch_reads=25)
    got = open(out, "rb").read()
    hdr = lf.sam_header("lordfast --search x", p)
    assert got.startswith(hdr)
    assert got[len(hdr):] == golden_sam("default"), first_diff(got[len(hdr):], golden_sam("default"))
    assert st["n_reads"] > 0
    out2 = str(tmp_path / "out2.sam")
    lf.map_file(reads, out2, params=p, header=False)
    assert open(out2, "rb").read() == golden_sam("default")


def test_sam_very_long_reads(tmp_path, oracle_lib):
    """55-125 kbp reads: chains of hundreds of anchors, gap problems above edlib's leaf size (host-orchestrated Hirschberg
    splits on the helper thread), the n > 2048 wave classes and records of > 10^5 CIGAR pieces"""
    import lordfast_amd as la
    g = synth.make_genome(1200000, 4, seed=31, n_families=60, repeat_frac=0.12)
    fa = la.index_build(g, str(tmp_path / "g.fa"))
    reads = synth.make_reads(g, 3, 130000, 0.12, seed=13, sigma=0.1) + synth.make_reads(g, 2, 60000, 0.15, seed=14, sigma=0.1) \
        + synth.make_reads(g, 2, 90000, 0.20, seed=15, sigma=0.1)
    names = [r[0].encode() for r in reads]
    seqs = [r[1] for r in reads]
    assert max(len(x) for x in seqs) > 100000
    h = la.LordFast(fa, device=0)
    sam, st = h.map_batch(names, seqs)
    h.close()
    orc = oracle_lib.Oracle(fa)
    exp = orc.map_batch(names, seqs, params=oracle_lib.default_params(threads=8))
    orc.close()
    assert sam == exp, first_diff(sam, exp)


def test_sam_lowercase_and_n_bases(lf, oracle, golden_reads):
    """seeding is case-insensitive (nst_nt4_table) but edlib compares raw bytes: lower-case read bases never match the
    upper-case reference; N never seeds and never matches"""
    names, seqs = golden_reads
    rng = np.random.default_rng(123)
    out_names, out_seqs = [], []
    for i, (n, s) in enumerate(zip(names[:40], seqs[:40])):
        b = bytearray(s)
        L = len(b)
        if i % 3 == 0:                                  # a lower-case stretch
            a = int(rng.integers(0, max(1, L - 400))); b[a:a + 300] = bytes(b[a:a + 300]).lower()
        if i % 3 == 1:                                  # runs of N
            for _ in range(3):
                a = int(rng.integers(0, max(1, L - 60))); b[a:a + int(rng.integers(1, 40))] = b"N" * len(b[a:a + int(rng.integers(1, 40))])
        if i % 3 == 2:                                  # scattered lower case + IUPAC codes
            for p in rng.integers(0, L, size=25):
                b[int(p)] = ord("acgtRYn"[int(rng.integers(0, 7))])
        out_names.append(n); out_seqs.append(bytes(b))
    sam, _ = lf.map_batch(out_names, out_seqs)
    exp = oracle.map_batch(out_names, out_seqs)
    assert sam == exp, first_diff(sam, exp)


@pytest.mark.parametrize("limit,lanes", [(3000, 8), (20000, 2), (1, 4)])
def test_chunk_is_cut_when_it_has_too_many_seed_hits(lf, golden_reads, monkeypatch, limit, lanes):
    """The vote sort's 2^30-hit limit per chunk is an implementation detail (the reference has none): a chunk above it is
    cut in two and mapped again (LF_MAX_CHUNK_HITS is the test hook).  Same records, same order, any number of lanes."""
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_MAX_CHUNK_HITS", str(limit))
    monkeypatch.setenv("LF_LANES", str(lanes))
    monkeypatch.setenv("LF_CHUNK_READS", "16")
    sam, st = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS["default"]))
    exp = golden_sam("default")
    assert sam == exp, first_diff(sam, exp)
    assert st["n_reads"] == len(seqs)


@pytest.mark.parametrize("cfg,chunk", [("default", 7), ("clasp_n30", 16), ("default", 0)])
def test_map_batch_multi_replicas_same_output(golden_dir, golden_reads, monkeypatch, cfg, chunk):
    """lf_map_batch_multi over three index replicas (on the one GPU of the test box; normally one per GPU): chunks are
    pulled by the replicas' lanes from one counter, records come out in input order -- the one-device SAM, byte for byte"""
    import lordfast_amd as la
    names, seqs = golden_reads
    hs = [la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=(k != 1)) for k in range(3)]   # one replica locates by LF walk
    try:
        if chunk:
            monkeypatch.setenv("LF_CHUNK_READS", str(chunk))
        sam, st = la.map_batch_multi(hs, names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    finally:
        for h in hs:
            h.close()
    exp = golden_sam(cfg)
    assert sam == exp, first_diff(sam, exp)
    assert st["n_reads"] == len(seqs)


def test_packed_upload_equals_byte_upload(lf, golden_reads, monkeypatch):
    """host batches go up as three bit planes made by the host threads + a list of the bytes that are not upper-case ACGT
    (lf_pack_read, lf_unpack_planes_kernel); LF_UPLOAD_PACKED=0 uploads the bytes.  Same SAM for: the golden reads, reads with
    N runs / IUPAC codes / lower-case stretches at word boundaries, very short reads (several per 64-bit plane word), and a batch
    that is lower case throughout (more exceptions than the list takes: the chunk falls back to bytes)"""
    import lordfast_amd as la
    names, seqs = golden_reads
    rng = np.random.default_rng(41)
    mixed = []
    for i, s in enumerate(seqs[:40]):
        b = bytearray(s)
        for a in (0, 63, 64, 65, 127, 128, len(b) - 70, len(b) - 1):
            if 0 <= a < len(b): b[a:a + int(rng.integers(1, 9))] = b"N" * len(b[a:a + int(rng.integers(1, 9))])
        a = int(rng.integers(0, max(1, len(b) - 200))); b[a:a + 130] = bytes(b[a:a + 130]).lower()
        b[int(rng.integers(0, len(b)))] = ord("R")
        mixed.append(bytes(b)[: len(b) - (i % 7)])
    tiny = [bytes(s[: int(rng.integers(1, 40))]) for s in seqs[:64]]
    lower = [bytes(s).lower() for s in seqs[:30]]
    for batch in (list(seqs), mixed + tiny, lower):
        nm = [b"r%d" % i for i in range(len(batch))]
        monkeypatch.delenv("LF_UPLOAD_PACKED", raising=False)
        sam_p, st_p = lf.map_batch(nm, batch)
        monkeypatch.setenv("LF_UPLOAD_PACKED", "0")
        sam_b, st_b = lf.map_batch(nm, batch)
        assert sam_p == sam_b, first_diff(sam_p, sam_b)
        assert st_p["n_seeds"] == st_b["n_seeds"]


@pytest.mark.parametrize("cfg,lds_max", [("default", 0), ("n30", 600), ("clasp", 3000)])
def test_vote_tables_in_global_memory(lf, golden_reads, monkeypatch, cfg, lds_max):
    """reads with more votes than the largest LDS hash table keep their vote table in a global scratch area
    (lf_vote_hash_kernel<true>); LF_VOTE_LDS_MAX_VOTES lowers the threshold so that the golden reads take that path"""
    import lordfast_amd as la
    names, seqs = golden_reads
    monkeypatch.setenv("LF_VOTE_LDS_MAX_VOTES", str(lds_max))
    sam, st = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    exp = golden_sam(cfg)
    assert sam == exp, first_diff(sam, exp)


@pytest.mark.parametrize("cfg", ["default", "n30", "clasp_n30", "k12c300m20"])
def test_vote_by_cells_equals_vote_by_window_scan(lf, golden_reads, monkeypatch, cfg):
    """lf_vote_cell_kernel (one insert per hit into a table of cells, score(w) = cell(w) + cell(w + 1), every cell's windows
    tested by the thread that created the cell) and lf_vote_hash_kernel (two votes per hit, table scans; LF_VOTE_SCAN=1) must
    select the same candidate windows: same SAM, same request counts"""
    import lordfast_amd as la
    names, seqs = golden_reads
    exp = golden_sam(cfg)
    sam_c, st_c = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    monkeypatch.setenv("LF_VOTE_SCAN", "1")
    sam_s, st_s = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    assert sam_c == exp, first_diff(sam_c, exp)
    assert sam_s == exp, first_diff(sam_s, exp)
    assert st_c["n_req_seeds"] == st_s["n_req_seeds"] > 0


@pytest.mark.parametrize("cfg", ["default", "clasp_n30", "k12c300m20"])
def test_sam_host_walk_crosscheck(lf, golden_reads, monkeypatch, cfg):
    """lf_debug_crosscheck(4) replays every chain with the host walk of lf_pipeline.c instead of lf_walk.hip (which keeps the
    common path of alignChain_edlib on the device): both must print the reference's records, and the device path must
    really have planned the alignments (no descriptors uploaded: n_edlib_problems is the same, counted on the device)"""
    import lordfast_amd as la
    names, seqs = golden_reads
    exp = golden_sam(cfg)
    sam_d, st_d = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    assert sam_d == exp, first_diff(sam_d, exp)
    la.lib().lf_debug_crosscheck(4)
    sam_h, st_h = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    assert sam_h == exp, first_diff(sam_h, exp)
    # the device plans every chain's common path; chains with a clip / split trigger are planned again by the host replay
    assert st_d["n_edlib_problems"] >= st_h["n_edlib_problems"] > 0


@pytest.mark.parametrize("cfg", ["default", "n30", "clasp"])
def test_sam_host_assembly_crosscheck(lf, golden_reads, monkeypatch, cfg):
    """lf_debug_crosscheck(8) formats the SAM lines on the host (two passes over the records) instead of lf_sam.hip (line descriptors
    -> text on the device, one D2H copy into the output buffer): same bytes.  With qualities (FASTQ) and a read group too."""
    import lordfast_amd as la
    names, seqs = golden_reads
    exp = golden_sam(cfg)
    la.lib().lf_debug_crosscheck(8)
    sam_h, _ = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    assert sam_h == exp, first_diff(sam_h, exp)
    if cfg == "default":
        quals = [bytes(33 + (i * 7 + k) % 40 for k in range(len(s))) if i % 3 else b"" for i, s in enumerate(seqs)]
        p = la.default_params(read_group_id=b"grpX", read_group=b"@RG\tID:grpX")
        h, _ = lf.map_batch(names, seqs, quals=quals, params=p)
        la.lib().lf_debug_crosscheck(0)
        d, _ = lf.map_batch(names, seqs, quals=quals, params=p)
        assert d == h, first_diff(d, h)
        assert b"\tRG:Z:grpX" in d


def _pack_to_device(names, seqs, quals=None):
    """the caller's side of lf_map_batch_dev: bases (and qualities) as ONE blob in HBM + host offsets / lengths.  (No torch in
    this process: torch ships its own HIP runtime, and a process that initialises it before liblfgpu.so runs the library on
    THAT runtime -- bench.py does so on purpose, the parity tests stay on the system one.)"""
    import lordfast_amd as la
    blob = b"\0".join(seqs) + b"\0"
    off = np.concatenate([[0], np.cumsum([len(s) + 1 for s in seqs])])[:-1].astype(np.uint64)
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    d_seqs = la.api.DeviceBuffer(len(blob), data=blob)
    d_quals = la.api.DeviceBuffer(len(blob), data=b"\0".join(quals) + b"\0") if quals is not None else None
    return la.api._cstr_array(names), d_seqs, off, lens, d_quals


@pytest.mark.parametrize("cfg", ["default", "clasp_n30", "k12c300m20"])
def test_map_batch_dev_resident_io(lf, golden_reads, cfg, monkeypatch):
    """lf_map_batch_dev: bases already in HBM, SAM text left in HBM -- the records the reference prints, also when the batch is
    cut into many chunks on several lanes (asynchronous device-to-device copies into the caller's buffer)"""
    import lordfast_amd as la
    names, seqs = golden_reads
    exp = golden_sam(cfg)
    na, d_seqs, off, lens, _ = _pack_to_device(names, seqs)
    out = la.api.DeviceBuffer(len(exp) + 4096)
    p = la.default_params(**GOLDEN_CONFIGS[cfg])
    ln, st = lf.map_batch_dev(na, d_seqs.ptr, off, lens, out.ptr, out.nbytes, True, params=p)
    got = out.download(ln)
    assert got == exp, first_diff(got, exp)
    assert st["n_reads"] == len(seqs)
    # host destination from device-resident bases
    hbuf = np.zeros(len(exp) + 4096, dtype=np.uint8)
    ln, _ = lf.map_batch_dev(na, d_seqs.ptr, off, lens, hbuf.ctypes.data, hbuf.size, False, params=p)
    assert hbuf[:ln].tobytes() == exp
    monkeypatch.setenv("LF_CHUNK_READS", "7"); monkeypatch.setenv("LF_LANES", "3")
    ln, _ = lf.map_batch_dev(na, d_seqs.ptr, off, lens, out.ptr, out.nbytes, True, params=p)
    got = out.download(ln)
    assert got == exp, first_diff(got, exp)
    # a buffer that is too small is an error, not a truncation
    with pytest.raises(la.api.LfError):
        lf.map_batch_dev(na, d_seqs.ptr, off, lens, out.ptr, 1000, True, params=p)
    d_seqs.free(); out.free()


def test_map_batch_dev_fastq_short_reads_and_readgroup(lf, oracle, oracle_lib, golden_reads):
    """device-resident qualities (reversed for reverse-strand records), reads shorter than -l (printed from bases fetched on
    demand) and a read group: the oracle's records"""
    import lordfast_amd as la
    names, seqs = golden_reads
    names, seqs = list(names[:24]), list(seqs[:24])
    seqs[3] = seqs[3][:400]; seqs[10] = seqs[10][:99]; seqs[17] = b"ACGT"          # below -l 1000
    rng = np.random.default_rng(5)
    quals = [bytes(rng.integers(35, 74, size=len(s)).astype(np.uint8)) for s in seqs]
    exp = oracle.map_batch(names, seqs, quals, params=oracle_lib.default_params(read_group_id=b"grp1"))
    na, d_seqs, off, lens, d_quals = _pack_to_device(names, seqs, quals)
    out = la.api.DeviceBuffer(len(exp) + 4096)
    ln, _ = lf.map_batch_dev(na, d_seqs.ptr, off, lens, out.ptr, out.nbytes, True, d_quals=d_quals.ptr,
                             params=la.default_params(read_group_id=b"grp1"))
    got = out.download(ln)
    assert got == exp, first_diff(got, exp)
    d_seqs.free(); d_quals.free(); out.free()


def test_reads_at_the_edges_of_the_reference(lf, oracle, golden_dir):
    """reads that start at the very first base / end at the very last base of a contig (and of the whole reference), on both
    strands: the alignment kernels read their target windows 16 symbols at a time and a lane that has not started yet asks
    for symbols in front of its target -- outside the 2-bit array at the edges of the reference.  Same records as the oracle."""
    import gzip
    import lordfast_amd as la
    from conftest import read_fasta, GOLDEN
    cn, cs = read_fasta(os.path.join(GOLDEN, "genome.fa.gz"))
    rng = np.random.default_rng(41)
    names, seqs = [], []
    for ci, c in enumerate(cs):
        a = np.frombuffer(c, dtype=np.uint8)
        for tag, piece in (("head", a[:2600]), ("tail", a[-2600:]), ("head_short", a[:1100]), ("tail_short", a[-1100:])):
            for strand in ("f", "r"):
                x = synth.mutate(piece, 0.12, rng)
                if strand == "r":
                    x = synth.revcomp(x)
                names.append(f"c{ci}_{tag}_{strand}".encode()); seqs.append(x.tobytes())
    sam, st = lf.map_batch(names, seqs)
    exp = oracle.map_batch(names, seqs)
    assert sam == exp, first_diff(sam, exp)
    mapped = [l for l in sam.split(b"\n") if l and not (int(l.split(b"\t")[1]) & 4)]
    assert len(mapped) >= len(names) - 2 and st["n_edlib_problems"] > 0
    assert any(l.split(b"\t")[3] == b"1" for l in mapped), "a record must start at position 1 of a contig"


def test_long_runs_of_matches_and_sparse_edits(lf, golden_dir, oracle_lib):
    """records whose CIGAR carries numbers of five digits and runs that span hundreds of recipe items and many event tiles of the
    renderer: 38 kbp reads with ONE mismatch every 1 500 bases (CIGAR `38000M`, MD `700C1499G...`), the same with a 3-base
    deletion / a 5-base insertion in the middle (an M run of 20 000 is closed inside an event tile: the general digit path),
    both strands.  Expected records: the compiled reference's (error-free stretches are quadratic in the reference's seed
    search and in the oracle's; 1 500 bases keep it at seconds)."""
    from conftest import read_fasta, GOLDEN, have_ref
    if not have_ref():
        pytest.skip("needs oracle/_ref/liblfref.so")
    import shutil
    cn, cs = read_fasta(os.path.join(GOLDEN, "genome.fa.gz"))
    a = np.frombuffer(max(cs, key=len), dtype=np.uint8)
    assert len(a) > 42000
    base = a[2000:40000].copy()
    for p in range(700, len(base), 1500):
        base[p] = ord("A") if base[p] != ord("A") else ord("G")
    names, seqs = [], []

    def add(tag, x):
        names.append((tag + "_f").encode()); seqs.append(bytes(x.tobytes()))
        names.append((tag + "_r").encode()); seqs.append(bytes(synth.revcomp(x).tobytes()))

    add("sparse", base)
    add("sparse_del3", np.concatenate([base[:20000], base[20003:]]))
    add("sparse_ins5", np.concatenate([base[:15000], np.frombuffer(b"ACGTA", dtype=np.uint8), base[15000:]]))
    add("exact1500", a[1000:2500].copy())              # an error-free read: every sample's match runs to the end of the read, all but
                                                       # the first are "contained" (src/BWT.cpp:345): the reference leaves it unmapped
    sam, st = lf.map_batch(names, seqs)
    fa = os.path.join(golden_dir, "genome.fa")
    ref = oracle_lib.Ref()
    made_cache = not os.path.exists(fa + ".cache")
    if made_cache:                                     # the reference reads its k-mer table from a file; the GPU side never does
        d2 = os.path.join(golden_dir, "refidx"); os.makedirs(d2, exist_ok=True)
        shutil.copy(fa, os.path.join(d2, "genome.fa")); fa = os.path.join(d2, "genome.fa")
        ref.index_build(fa)
    ref.load(fa)
    ref.set_params(oracle_lib.default_params(threads=1), "t")            # one thread: records in input order
    exp, _ = ref.map_mem(names, seqs)
    assert sam == exp, first_diff(sam, exp)
    import re
    recs = [l.split(b"\t") for l in sam.split(b"\n") if l]
    assert sum(f[5] == b"38000M" for f in recs) == 2, "mismatches only: one M run"
    assert sum(bool(re.search(rb"\d{5}M\d[ID]", f[5])) for f in recs) == 4, "five-digit runs closed by an indel"


@pytest.mark.parametrize("chunk,lanes", [(0, 8), (5, 8), (9, 2)])
def test_seq_less_egress_into_a_pinned_buffer(lf, oracle, oracle_lib, golden_reads, monkeypatch, chunk, lanes):
    """lf_map_batch_into with PINNED host memory: the SEQ / QUAL columns are left out of the device text, a kernel stores the rest
    of every line straight into the buffer, host threads fill SEQ (reverse-complemented for flag 16) / QUAL from the caller's
    strings (lf_sam.hip HOLES mode) -- same bytes as the whole-line path (LF_SAM_FULL=1) and the oracle: FASTA and FASTQ, a read
    group, records on both strands, secondaries, unmapped and too-short reads, chunks waiting for their place in the output"""
    import ctypes as C
    import lordfast_amd as la
    names, seqs = golden_reads
    L = lf.L
    L.lfg_host_alloc.restype = C.c_void_p
    L.lfg_host_alloc.argtypes = [C.c_size_t]
    L.lfg_host_free.argtypes = [C.c_void_p]
    L.lfg_host_mapped.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    cap = 8 << 20
    buf = L.lfg_host_alloc(cap)
    assert buf and L.lfg_host_mapped(0, buf, cap) == 1
    plain = np.zeros(16, dtype=np.uint8)
    assert L.lfg_host_mapped(0, plain.ctypes.data, plain.size) == 0          # ordinary memory: the whole-line path
    if chunk:
        monkeypatch.setenv("LF_CHUNK_READS", str(chunk))
    monkeypatch.setenv("LF_LANES", str(lanes))
    rng = np.random.default_rng(9)
    quals = [bytes(rng.integers(35, 74, size=len(s)).astype(np.uint8)) for s in seqs]
    try:
        for q, kw in ((None, dict(GOLDEN_CONFIGS["n30"])), (quals, dict(read_group_id=b"grp1"))):
            p = la.default_params(**kw)
            exp = oracle.map_batch(names, seqs, q, params=oracle_lib.default_params(**kw))
            C.memset(buf, 0x23, cap)
            ln, _ = lf.map_batch_into(names, seqs, buf, cap, quals=q, params=p)
            got = C.string_at(buf, ln)
            assert got == exp, first_diff(got, exp)
            assert C.string_at(buf + ln, 1) == b"\0"
            monkeypatch.setenv("LF_SAM_FULL", "1")
            C.memset(buf, 0x23, cap)
            ln2, _ = lf.map_batch_into(names, seqs, buf, cap, quals=q, params=p)
            assert C.string_at(buf, ln2) == exp
            monkeypatch.delenv("LF_SAM_FULL")
    finally:
        L.lfg_host_free(buf)


@pytest.mark.parametrize("chunk,lanes", [(0, 8), (7, 3), (1, 2)])
def test_prepacked_batch_maps_like_the_strings(lf, oracle, oracle_lib, golden_reads, monkeypatch, chunk, lanes):
    """lf_batch_create + lf_map_batch_from (SURVEY 7 step 3's lf_batch_create; the reference's readChunk leaves a mapper-ready chunk outside its
    mapping timer, src/Reads.cpp:84-104): the batch's bit planes are made ONCE, a chunk uploads its bit range of them and the device shifts it into
    place (lf_seed.hip: lf_planes_shift_kernel).  Same bytes as lf_map_batch on the strings: chunks that start at any bit offset (one / seven reads per
    chunk), reads below -l (not packed), N and lower-case bases (exception list), FASTQ, a pinned and an ordinary output buffer, a -l other than the
    batch's (falls back to packing per call)."""
    import ctypes as C
    import lordfast_amd as la
    names, seqs = golden_reads
    names, seqs = list(names), list(seqs)
    rng = np.random.default_rng(31)
    seqs[3] = seqs[3][:500] + b"N" * 7 + seqs[3][507:]
    seqs[5] = seqs[5].lower()
    seqs.insert(4, seqs[0][:650]); names.insert(4, b"short_one")                       # shorter than -l 1000: unmapped record, not in the planes
    quals = [bytes(rng.integers(35, 74, size=len(s)).astype(np.uint8)) for s in seqs]
    L = lf.L
    L.lfg_host_alloc.restype = C.c_void_p
    L.lfg_host_alloc.argtypes = [C.c_size_t]
    L.lfg_host_free.argtypes = [C.c_void_p]
    cap = 8 << 20
    buf = L.lfg_host_alloc(cap)
    plain = np.zeros(cap, dtype=np.uint8)
    if chunk:
        monkeypatch.setenv("LF_CHUNK_READS", str(chunk))
    monkeypatch.setenv("LF_LANES", str(lanes))
    try:
        for q in (None, quals):
            kw = dict(read_group_id=b"grp1") if q is not None else {}
            exp = oracle.map_batch(names, seqs, q, params=oracle_lib.default_params(**kw))
            b = la.ReadBatch(names, seqs, q, min_read_len=1000)
            for ptr in (buf, plain.ctypes.data):
                C.memset(ptr, 0x23, cap)
                ln, st = lf.map_batch_from(b, ptr, cap, params=la.default_params(**kw))
                got = C.string_at(ptr, ln)
                assert got == exp, first_diff(got, exp)
                assert st["n_reads"] == len(seqs)
            # another -l than the one the batch was packed for: same call, per-call packing
            kw2 = dict(kw, min_read_len=600)
            exp2 = oracle.map_batch(names, seqs, q, params=oracle_lib.default_params(**kw2))
            ln, _ = lf.map_batch_from(b, buf, cap, params=la.default_params(**kw2))
            assert C.string_at(buf, ln) == exp2
            b.close()
    finally:
        L.lfg_host_free(buf)
