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
