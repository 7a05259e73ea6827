"""CPU: the host side of the packed upload (lf_pack_read, lf_samdesc.c) against a numpy restatement of what lf_pack_planes_kernel makes
of the bytes on the device: bit i of word i / 64 of plane x describes base i of the chunk -- code low bit, code high bit (A0 C1 G2
T3), "is one of ACGT" (upper case) -- and every other byte goes to the exception list with its position."""
import ctypes as C

import numpy as np

import lordfast_amd as la


def planes_ref(chunk: bytes, qw: int):
    b = np.frombuffer(chunk, dtype=np.uint8)
    code = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    c = code[b]
    valid = c != 255
    lo = np.zeros(qw * 64, dtype=np.uint8); hi = np.zeros(qw * 64, dtype=np.uint8); va = np.zeros(qw * 64, dtype=np.uint8)
    lo[: len(b)] = valid & ((c & 1) == 1); hi[: len(b)] = valid & ((c & 2) == 2); va[: len(b)] = valid
    pack = lambda bits: np.packbits(bits.reshape(-1, 64), axis=1, bitorder="little").view("<u8").ravel()
    exc = {int(p): int(b[p]) for p in np.nonzero(~valid)[0]}
    return pack(lo), pack(hi), pack(va), exc


def test_pack_read_equals_device_plane_layout():
    L = la.lib()
    L.lf_pack_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.lf_pack_read.restype = C.c_int
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGT" * 12 + b"NacgtRY", dtype=np.uint8)
    lens = [1, 2, 63, 64, 65, 127, 128, 129, 1, 1, 1, 70, 3000, 5, 64, 64, 191, 4097] + [int(x) for x in rng.integers(1, 700, size=40)]
    reads = [bytes(alphabet[rng.integers(0, len(alphabet), size=n)]) if i % 3 else bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)])
             for i, n in enumerate(lens)]
    chunk = b"".join(reads)
    qw = (len(chunk) + 63) // 64 + 2
    planes = np.full(3 * qw, 0xDEADBEEFDEADBEEF, dtype=np.uint64)          # whole words are plain stores: garbage must not survive
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    for x in range(3):                                                       # what the caller zeroes: the words with a read boundary, the slack
        planes[x * qw + (off >> np.uint64(6))] = 0
        planes[x * qw + (int(off[-1]) >> 6): (x + 1) * qw] = 0
    cap = len(chunk) + 16
    xpos = np.zeros(cap, dtype=np.uint64); xbyte = np.zeros(cap, dtype=np.uint8); xn = C.c_uint64(0)
    for k, r in enumerate(reads):
        ok = L.lf_pack_read(planes.ctypes.data, qw, int(off[k]), r, len(r), xpos.ctypes.data, xbyte.ctypes.data, cap, C.byref(xn))
        assert ok == 1
    lo, hi, va, exc = planes_ref(chunk, qw)
    assert np.array_equal(planes[:qw], lo) and np.array_equal(planes[qw:2 * qw], hi) and np.array_equal(planes[2 * qw:], va)
    got = {int(p): int(b) for p, b in zip(xpos[: xn.value], xbyte[: xn.value])}
    assert got == exc
    # a list that is too short: the call says so (the caller uploads the bytes instead) and never writes past the capacity
    xn2 = C.c_uint64(0); small = np.zeros(4, dtype=np.uint64); smallb = np.full(8, 7, dtype=np.uint8)
    oks = [L.lf_pack_read(planes.ctypes.data, qw, int(off[k]), r, len(r), small.ctypes.data, smallb.ctypes.data, 4, C.byref(xn2)) for k, r in enumerate(reads)]
    assert 0 in oks and xn2.value == len(exc) and np.all(smallb[4:] == 7)


def test_prepacked_batch_planes_equal_the_per_call_packing():
    """lf_batch_create (lf_sched.c: lf_read_batch_prepack) packs a whole batch once; a chunk of it is a bit range of those planes.  Every read's bits
    must equal what lf_pack_read makes of the read alone, reads below -l must be left out, the exception list must be sorted."""
    import lordfast_amd as la
    rng = np.random.default_rng(5)
    reads = []
    for i in range(300):
        n = int(rng.integers(200, 3000))
        r = bytearray(b"ACGT"[int(x)] for x in rng.integers(0, 4, n))
        if i % 17 == 0:
            r[n // 2] = ord("N")
        if i % 29 == 0:
            r[3:9] = b"acgtnn"
        reads.append(bytes(r))
    names = [b"r%d" % i for i in range(len(reads))]
    b = la.ReadBatch(names, reads, min_read_len=1000, threads=3)
    assert len(b) == len(reads)
    L = la.lib()

    class Pre(C.Structure):
        _fields_ = [("min_read_len", C.c_int), ("bases", C.c_uint64), ("QW", C.c_uint64), ("planes", C.POINTER(C.c_uint64)), ("pinned", C.c_int),
                    ("boff", C.POINTER(C.c_uint64)), ("exc_pos", C.POINTER(C.c_uint64)), ("exc_byte", C.POINTER(C.c_uint8)), ("n_exc", C.c_uint64)]

    class Batch(C.Structure):
        _fields_ = [("n", C.c_int), ("bases", C.c_uint64), ("names", C.c_void_p), ("seqs", C.c_void_p), ("quals", C.c_void_p), ("lens", C.POINTER(C.c_uint32)), ("rcap", C.c_int),
                    ("blobs", C.c_void_p), ("blob_caps", C.c_void_p), ("nblobs", C.c_int), ("capblobs", C.c_int), ("blob", C.c_void_p), ("blob_n", C.c_size_t), ("blob_cap", C.c_size_t),
                    ("off", C.c_void_p), ("cap", C.c_int), ("pre", C.POINTER(Pre))]

    B = C.cast(b.h, C.POINTER(Batch)).contents
    P = B.pre.contents
    assert P.min_read_len == 1000 and P.pinned == 0                       # (no device here: plain memory)
    mapped = [r for r in reads if len(r) >= 1000]
    assert P.bases == sum(len(r) for r in mapped)
    QW = int(P.QW)
    planes = np.ctypeslib.as_array(P.planes, shape=(3 * QW,)).reshape(3, QW)
    bits = np.unpackbits(planes.view(np.uint8), bitorder="little").reshape(3, QW * 64)
    o = 0
    for i, r in enumerate(reads):
        assert P.boff[i] == o
        if len(r) < 1000:
            continue
        a = np.frombuffer(r, dtype=np.uint8)
        ok = np.isin(a, np.frombuffer(b"ACGT", dtype=np.uint8))
        code = np.select([a == ord("A"), a == ord("C"), a == ord("G"), a == ord("T")], [0, 1, 2, 3], 0)
        assert np.array_equal(bits[2, o:o + len(r)], ok.astype(np.uint8)), i
        assert np.array_equal(bits[0, o:o + len(r)], ((code & 1) * ok).astype(np.uint8)), i
        assert np.array_equal(bits[1, o:o + len(r)], ((code >> 1) * ok).astype(np.uint8)), i
        o += len(r)
    assert P.boff[len(reads)] == o and not bits[:, o:].any()
    pos = np.ctypeslib.as_array(P.exc_pos, shape=(int(P.n_exc),)) if P.n_exc else np.zeros(0, np.uint64)
    byt = np.ctypeslib.as_array(P.exc_byte, shape=(int(P.n_exc),)) if P.n_exc else np.zeros(0, np.uint8)
    assert np.all(np.diff(pos.astype(np.int64)) > 0)
    cat = np.frombuffer(b"".join(mapped), dtype=np.uint8)
    bad = np.nonzero(~np.isin(cat, np.frombuffer(b"ACGT", dtype=np.uint8)))[0]
    assert np.array_equal(pos, bad.astype(np.uint64)) and np.array_equal(byt, cat[bad])
    b.close()
