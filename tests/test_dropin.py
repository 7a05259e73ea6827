"""GPU parity of the DROP-IN entry points: the reference's own function names (src/BWT.h:28-39, src/Chain.h:50-51,
src/LordFAST.h:122-126, lib/edlib/edlib.h:172,190, lib/bwa/ksw.h:107-108) called through ctypes exactly the way the
reference's driver and mapper call them, against the golden vectors of the compiled reference.

The drop-ins keep the reference's process-global state, so the whole module shares one `bwt_load`."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import golden_sam, split_ragged

pytestmark = pytest.mark.gpu


class SeedList(C.Structure):                       # src/LordFAST.h:37-41
    _fields_ = [("list", C.c_void_p), ("num", C.c_uint32)]


class Chain(C.Structure):                          # src/LordFAST.h:65-70
    _fields_ = [("seeds", C.c_void_p), ("chainLen", C.c_uint32), ("score", C.c_float)]


class EdlibAlignConfig(C.Structure):               # lib/edlib/edlib.h:79-104
    _fields_ = [("k", C.c_int), ("mode", C.c_int), ("task", C.c_int)]


class EdlibAlignResult(C.Structure):               # lib/edlib/edlib.h:113-135
    _fields_ = [("editDistance", C.c_int), ("endLocations", C.POINTER(C.c_int)), ("startLocations", C.POINTER(C.c_int)),
                ("numLocations", C.c_int), ("alignment", C.POINTER(C.c_ubyte)), ("alignmentLength", C.c_int),
                ("alphabetLength", C.c_int)]


class Read(C.Structure):                           # src/Reads.h:28-35
    _fields_ = [("length", C.POINTER(C.c_uint32)), ("seq", C.c_char_p), ("qual", C.c_char_p), ("name", C.c_char_p),
                ("isFq", C.POINTER(C.c_uint8))]


@pytest.fixture(scope="module")
def D(golden_dir):
    """liblfgpu.so with the drop-in prototypes declared and the golden index loaded by bwt_load()"""
    import lordfast_amd as la
    L = la.lib()
    assert la.device_count() >= 1, "no gfx950 device: the HIP path has no CPU fallback"
    L.bwt_load.argtypes = [C.c_char_p]
    L.bwt_get_refGenLen.restype = C.c_uint32
    L.getLocs_extend_whole_step.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(SeedList), C.POINTER(SeedList)]
    L.getLocs_extend_whole_step.restype = None
    L.bwt_get_chr_boundaries.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.bwt_get_chr_boundaries.restype = None
    L.bwt_get_intv_info.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.bwt_get_intv_info.restype = None
    L.bwt_str_pac2char.argtypes = [C.c_uint32, C.c_uint32, C.c_char_p]
    L.bwt_str_pac2char.restype = None
    L.bwt_str_pac2int.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    L.bwt_str_pac2int.restype = None
    L.chain_seeds_n2.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Chain)]
    L.chain_seeds_n2.restype = None
    L.chain_seeds_clasp.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Chain)]
    L.edlibNewAlignConfig.argtypes = [C.c_int, C.c_int, C.c_int]
    L.edlibNewAlignConfig.restype = EdlibAlignConfig
    L.edlibAlign.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, EdlibAlignConfig]
    L.edlibAlign.restype = EdlibAlignResult
    L.edlibFreeAlignResult.argtypes = [EdlibAlignResult]
    L.edlibFreeAlignResult.restype = None
    L.ksw_extend2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p] * 5
    L.ksw_extend.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 5
    L.initFASTChunk.argtypes = [C.POINTER(Read), C.c_int]
    L.initFASTChunk.restype = None
    for f in ("initializeFAST", "finalizeFAST", "mapSeqMT"):
        getattr(L, f).restype = None
    gp = la.Params.in_dll(L, "lf_global_params")
    dp = la.default_params()
    C.memmove(C.byref(gp), C.byref(dp), C.sizeof(la.Params))        # the defaults of src/CommandLineParser.cpp:41-55
    assert L.bwt_load(os.path.join(golden_dir, "genome.fa").encode()) == 0
    return L


def _genome(golden_dir):
    from conftest import read_fasta
    return read_fasta(os.path.join(golden_dir, "genome.fa"))


def test_bwt_load_and_reference_fetch(D, golden_dir):
    """bwt_get_refGenLen, bwt_str_pac2char/int, bwt_get_chr_boundaries, bwt_get_intv_info (src/BWT.cpp:305,593-666)"""
    names, seqs = _genome(golden_dir)
    total = sum(len(s) for s in seqs)
    assert D.bwt_get_refGenLen() == total
    off = 0
    rng = np.random.default_rng(3)
    for nm, s in zip(names, seqs):
        up = s.upper()
        for _ in range(20):
            a = int(rng.integers(0, len(s) - 300)); ln = int(rng.integers(1, 300))
            buf = C.create_string_buffer(ln + 1)
            D.bwt_str_pac2char(off + a, ln, buf)
            got = buf.raw[:ln]
            exp = up[a:a + ln]
            # non-ACGT reference bases were replaced by random ones at index time (lib/bwa/bntseq.c:261)
            assert all(g == e or e not in b"ACGT" for g, e in zip(got, exp))
            codes = np.zeros(ln, dtype=np.uint8)
            D.bwt_str_pac2int(off + a, ln, codes.ctypes.data)
            assert bytes(b"ACGT"[c] for c in codes) == got
            cb, ce = C.c_uint32(), C.c_uint32()
            D.bwt_get_chr_boundaries(off + a, off + a + ln, C.byref(cb), C.byref(ce))
            assert (cb.value, ce.value) == (off, off + len(s) - 1)
            cn, cl, b0, b1 = C.c_char_p(), C.c_int32(), C.c_uint32(), C.c_uint32()
            D.bwt_get_intv_info(off + a, off + a + ln, C.byref(cn), C.byref(cl), C.byref(b0), C.byref(b1))
            assert (cn.value, cl.value, b0.value, b1.value) == (nm, len(s), a, a + ln)
        off += len(s)


def test_getLocs_extend_whole_step(D, golden_reads, stages):
    """src/BWT.h:32 one read at a time, caller-allocated SeedLists (src/LordFAST.cpp:134-139 capacities)"""
    from lordfast_amd.api import _seeds_to_triples
    names, seqs = golden_reads
    F = split_ragged(stages["seed_F"], stages["seed_F_n"])
    R = split_ragged(stages["seed_R"], stages["seed_R_n"])
    cap = 1000 * 1000
    bf, br = np.zeros((cap, 2), dtype=np.uint32), np.zeros((cap, 2), dtype=np.uint32)
    for i, s in enumerate(seqs):
        sf, sr = SeedList(bf.ctypes.data, 0), SeedList(br.ctypes.data, 0)
        D.getLocs_extend_whole_step(s, len(s), 1000, C.byref(sf), C.byref(sr))
        assert np.array_equal(_seeds_to_triples(bf[:sf.num].copy()), F[i]), names[i]
        assert np.array_equal(_seeds_to_triples(br[:sr.num].copy()), R[i]), names[i]


def test_chain_seeds_n2(D, stages):
    """src/Chain.h:50: reorders the caller's list like std::sort, fills the caller's Chain_t"""
    from lordfast_amd.api import _seeds_to_triples, _triples_to_seeds
    ins = split_ragged(stages["chain_in"], stages["chain_n"])
    srt = split_ragged(stages["chain_sorted"], stages["chain_n"])
    outs = split_ragged(stages["chain_out"], stages["chain_out_n"])
    for i, (a, b, o, sc) in enumerate(zip(ins, srt, outs, stages["chain_score"])):
        s = _triples_to_seeds(a)
        buf = np.zeros((len(a) + 1, 2), dtype=np.uint32)
        ch = Chain(buf.ctypes.data, 0, 0.0)
        D.chain_seeds_n2(s.ctypes.data, len(a), C.byref(ch))
        assert np.array_equal(_seeds_to_triples(s.reshape(-1)), b), i
        assert ch.chainLen == len(o) and np.array_equal(_seeds_to_triples(buf[:ch.chainLen].copy()), o), i
        assert np.float32(ch.score) == np.float32(sc), i


def test_chain_seeds_clasp(D, stages_clasp):
    """src/Chain.h:51: list untouched, chain in target order, returns 1"""
    from lordfast_amd.api import _seeds_to_triples, _triples_to_seeds
    st = stages_clasp
    ins = split_ragged(st["clasp_in"], st["clasp_n"])
    outs = split_ragged(st["clasp_out"], st["clasp_out_n"])
    for i, (a, o, sc) in enumerate(zip(ins, outs, st["clasp_score"])):
        if len(a) == 0:
            continue
        s = _triples_to_seeds(a)
        keep = s.copy()
        buf = np.zeros((len(a) + 1, 2), dtype=np.uint32)
        ch = Chain(buf.ctypes.data, 0, 0.0)
        assert D.chain_seeds_clasp(s.ctypes.data, len(a), C.byref(ch)) == 1
        assert np.array_equal(s, keep), i
        assert ch.chainLen == len(o) and np.array_equal(_seeds_to_triples(buf[:ch.chainLen].copy()), o), i
        assert np.float32(ch.score) == np.float32(sc), i


def test_edlibAlign_and_free(D, stages):
    """lib/edlib/edlib.h:190,172: {k = -1, NW | SHW, PATH} -> malloc'd result arrays freed by edlibFreeAlignResult"""
    qs = split_ragged(stages["ed_q"].tobytes(), stages["ed_qn"])
    ts = split_ragged(stages["ed_t"].tobytes(), stages["ed_tn"])
    ops = split_ragged(stages["ed_ops"], stages["ed_opsn"])
    step = max(1, len(qs) // 120)                       # one tiny launch per call: a spread sample keeps the test short
    idx = sorted(set(list(range(0, len(qs), step)) + [int(np.argmax(stages["ed_qn"]))]))
    for i in idx:
        cfg = D.edlibNewAlignConfig(-1, int(stages["ed_mode"][i]), 2)
        r = D.edlibAlign(qs[i], len(qs[i]), ts[i], len(ts[i]), cfg)
        assert r.editDistance == int(stages["ed_dist"][i]) and r.numLocations == 1, i
        assert r.endLocations[0] == int(stages["ed_end"][i]) and r.startLocations[0] == 0, i
        got = np.ctypeslib.as_array(r.alignment, shape=(max(r.alignmentLength, 1),))[:r.alignmentLength]
        assert r.alignmentLength == len(ops[i]) and np.array_equal(got, ops[i]), i
        D.edlibFreeAlignResult(r)
    # k >= 0 and the distance above it: "no solution" (editDistance -1, lib/edlib/edlib.cpp:155-160)
    r = D.edlibAlign(b"AAAAAAAAAA", 10, b"CCCCCCCCCC", 10, D.edlibNewAlignConfig(3, 0, 2))
    assert r.editDistance == -1 and not r.alignment
    D.edlibFreeAlignResult(r)
    # task DISTANCE: no path, no start locations
    r = D.edlibAlign(b"ACGTACGT", 8, b"ACGAACGT", 8, D.edlibNewAlignConfig(-1, 0, 0))
    assert r.editDistance == 1 and r.endLocations[0] == 7 and not r.alignment and not r.startLocations
    D.edlibFreeAlignResult(r)


def test_ksw_extend2_and_ksw_extend(D, stages):
    """lib/bwa/ksw.h:107-108 with the clip matrix of src/LordFAST.cpp:178-187"""
    qs = split_ragged(stages["ksw_q"], stages["ksw_qn"])
    ts = split_ragged(stages["ksw_t"], stages["ksw_tn"])
    mat = np.zeros(25, dtype=np.int8)
    for i in range(4):
        for j in range(4):
            mat[i * 5 + j] = 2 if i == j else -16
    for i, (q, t, prm, res) in enumerate(zip(qs, ts, stages["ksw_prm"], stages["ksw_res"])):
        q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
        o_del, e_del, o_ins, e_ins, w, zdrop, h0 = (int(x) for x in prm)
        qle, tle = C.c_int(-7), C.c_int(-7)
        sc = D.ksw_extend2(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, mat.ctypes.data, o_del, e_del, o_ins, e_ins,
                           w, 0, zdrop, h0, C.addressof(qle), C.addressof(tle), None, None, None)
        assert (sc, qle.value, tle.value) == tuple(int(x) for x in res), i
        if o_del == o_ins and e_del == e_ins:
            sc2 = D.ksw_extend(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, mat.ctypes.data, o_del, e_del,
                               w, 0, zdrop, h0, C.addressof(qle), C.addressof(tle), None, None, None)
            assert (sc2, qle.value, tle.value) == tuple(int(x) for x in res), i


def test_chunk_driver_prints_the_reference_sam(D, golden_reads, tmp_path):
    """src/baseFAST.cpp:56-81 as the reference's main() drives it: initializeFAST -> initFASTChunk(Read[]) ->
    mapSeqMT -> finalizeFAST must write header + the records of expected_default.sam.gz"""
    import lordfast_amd as la
    names, seqs = golden_reads
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    out = str(tmp_path / "dropin.sam")
    fp = libc.fopen(out.encode(), b"w")
    assert fp
    C.c_void_p.in_dll(D, "lf_global_output").value = fp
    C.c_int.in_dll(D, "lf_global_no_header").value = 0
    cmd = b"lordfast --search genome.fa --seq reads.fa"
    C.memmove(C.addressof((C.c_char * 2000).in_dll(D, "lf_global_cmdline")), cmd + b"\0", len(cmd) + 1)
    n = len(seqs)
    lens = (C.c_uint32 * n)(*[len(s) for s in seqs])
    isfq = (C.c_uint8 * n)(*([0] * n))
    chunk = (Read * n)()
    for i in range(n):
        chunk[i].length = C.cast(C.byref(lens, 4 * i), C.POINTER(C.c_uint32))
        chunk[i].seq = seqs[i]; chunk[i].qual = b""; chunk[i].name = names[i]
        chunk[i].isFq = C.cast(C.byref(isfq, i), C.POINTER(C.c_uint8))
    D.initializeFAST()
    half = n // 2                                           # two chunks, like two readChunk() rounds
    D.initFASTChunk(chunk, half); D.mapSeqMT()
    D.initFASTChunk(C.cast(C.byref(chunk, C.sizeof(Read) * half), C.POINTER(Read)), n - half); D.mapSeqMT()
    D.finalizeFAST()
    libc.fclose(fp)
    C.c_void_p.in_dll(D, "lf_global_output").value = None
    txt = open(out, "rb").read()
    head = [l for l in txt.split(b"\n") if l.startswith(b"@")]
    body = b"".join(l + b"\n" for l in txt.split(b"\n") if l and not l.startswith(b"@"))
    assert head[0] == b"@HD\tVN:1.5\tSO:unsorted" and head[-1] == b"@PG\tID:lordfast\tPN:lordfast\tVN:0.0.10\tCL:" + cmd
    assert sum(l.startswith(b"@SQ") for l in head) >= 2
    assert body == golden_sam("default")
