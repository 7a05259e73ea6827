"""ctypes bindings for the two CHECKERS -- TEST INFRASTRUCTURE ONLY.

  Oracle  -> oracle/liblforacle.so   (our plain-C restatement, lf_oracle.h)
  Ref     -> oracle/_ref/liblfref.so (the real reference compiled by oracle/Makefile)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liblforacle.so")
REF_SO = os.path.join(_HERE, "_ref", "liblfref.so")


def build(quiet: bool = True) -> None:
    """make oracle (+ ref when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class Params(C.Structure):
    _fields_ = [("min_anchor_len", C.c_int), ("sampling_count", C.c_int), ("max_map", C.c_int),
                ("min_read_len", C.c_int), ("max_ref_hits", C.c_int), ("chain_alg", C.c_int),
                ("chain_reward", C.c_double), ("chain_penalty", C.c_double), ("gap_penalty", C.c_double),
                ("threads", C.c_int), ("read_group_id", C.c_char * 256), ("read_group", C.c_char * 1000)]


def default_params(**kw) -> Params:
    p = Params(14, 1000, 10, 1000, 1000, 0, 9.3, 11.4, 0.15, 1, b"", b"")
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _cstr_array(items):
    arr = (C.c_char_p * len(items))()
    arr[:] = [x if isinstance(x, bytes) else x.encode() for x in items]
    return arr


class Oracle:
    def __init__(self, prefix: str | None = None):
        if not os.path.exists(ORACLE_SO):
            build()
        L = self.L = C.CDLL(ORACLE_SO)
        L.lfo_index_load.restype = C.c_void_p
        L.lfo_index_load.argtypes = [C.c_char_p]
        L.lfo_index_free.argtypes = [C.c_void_p]
        L.lfo_map_batch.restype = C.c_void_p
        L.lfo_map_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_char_p),
                                    C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
        L.lfo_sam_header.restype = C.c_void_p
        L.lfo_sam_header.argtypes = [C.c_void_p, C.POINTER(Params), C.c_char_p]
        L.lfo_free.argtypes = [C.c_void_p]
        L.lfo_seed.argtypes = [C.c_void_p, C.POINTER(Params), C.c_char_p, C.c_uint32, C.c_void_p,
                               C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
        L.lfo_chain_n2.argtypes = [C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_void_p,
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.lfo_chain_clasp.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.lfo_edlib.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int),
                                C.c_void_p, C.POINTER(C.c_int)]
        L.lfo_ksw_extend2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7 + \
                                     [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lfo_sort_seeds_by_qpos.argtypes = [C.c_void_p, C.c_size_t]
        L.lfo_pac2char.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p]
        self.idx = None
        if prefix is not None:
            self.load(prefix)

    def load(self, prefix: str):
        self.idx = self.L.lfo_index_load(prefix.encode())
        if not self.idx:
            raise RuntimeError(f"oracle: cannot load index {prefix}")

    def close(self):
        if self.idx:
            self.L.lfo_index_free(self.idx)
            self.idx = None

    def map_batch(self, names, seqs, quals=None, params: Params | None = None) -> bytes:
        p = params or default_params()
        n = len(names)
        q = _cstr_array(quals) if quals is not None else None
        ln = C.c_size_t()
        ptr = self.L.lfo_map_batch(self.idx, C.byref(p), n, _cstr_array(names), _cstr_array(seqs), q, C.byref(ln))
        out = C.string_at(ptr, ln.value)
        self.L.lfo_free(ptr)
        return out

    def sam_header(self, cmdline: str, params: Params | None = None) -> bytes:
        p = params or default_params()
        ptr = self.L.lfo_sam_header(self.idx, C.byref(p), cmdline.encode())
        out = C.string_at(ptr)
        self.L.lfo_free(ptr)
        return out

    def seed(self, seq: bytes, params: Params | None = None, stats: bool = False):
        p = params or default_params()
        cap = p.sampling_count * p.max_ref_hits
        F = np.zeros((cap, 3), dtype=np.uint32)
        R = np.zeros((cap, 3), dtype=np.uint32)
        nF, nR = C.c_uint32(), C.c_uint32()
        st = np.zeros(4, dtype=np.uint64)
        self.L.lfo_seed(self.idx, C.byref(p), seq, len(seq), F.ctypes.data, C.byref(nF), R.ctypes.data,
                        C.byref(nR), st.ctypes.data if stats else None)
        if stats:
            return F[:nF.value].copy(), R[:nR.value].copy(), st
        return F[:nF.value].copy(), R[:nR.value].copy()

    def chain_n2(self, seeds: np.ndarray, params: Params | None = None):
        p = params or default_params()
        s = np.ascontiguousarray(seeds, dtype=np.uint32).copy()
        n = len(s)
        out = np.zeros((max(n, 1), 3), dtype=np.uint32)
        cl, sc = C.c_uint32(), C.c_float()
        self.L.lfo_chain_n2(C.byref(p), s.ctypes.data, n, out.ctypes.data, C.byref(cl), C.byref(sc))
        return s, out[:cl.value].copy(), sc.value

    def chain_clasp(self, seeds: np.ndarray):
        s = np.ascontiguousarray(seeds, dtype=np.uint32).copy()
        n = len(s)
        out = np.zeros((max(n, 1), 3), dtype=np.uint32)
        cl, sc = C.c_uint32(), C.c_float()
        self.L.lfo_chain_clasp(s.ctypes.data, n, out.ctypes.data, C.byref(cl), C.byref(sc))
        return out[:cl.value].copy(), sc.value

    def edlib(self, q: bytes, t: bytes, mode: int):
        ops = np.zeros(len(q) + len(t) + 1, dtype=np.uint8)
        end, nops = C.c_int(), C.c_int()
        ed = self.L.lfo_edlib(q, len(q), t, len(t), mode, C.byref(end), ops.ctypes.data, C.byref(nops))
        return ed, end.value, ops[:nops.value].copy()

    def ksw_extend2(self, q: np.ndarray, t: np.ndarray, o_del, e_del, o_ins, e_ins, w, zdrop, h0):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        qle, tle = C.c_int(), C.c_int()
        sc = self.L.lfo_ksw_extend2(len(q), q.ctypes.data, len(t), t.ctypes.data, o_del, e_del, o_ins, e_ins,
                                    w, zdrop, h0, C.byref(qle), C.byref(tle))
        return sc, qle.value, tle.value

    def sort_seeds(self, seeds: np.ndarray) -> np.ndarray:
        s = np.ascontiguousarray(seeds, dtype=np.uint32).copy()
        self.L.lfo_sort_seeds_by_qpos(s.ctypes.data, len(s))
        return s

    def pac2char(self, beg: int, n: int) -> bytes:
        buf = C.create_string_buffer(n + 1)
        self.L.lfo_pac2char(self.idx, beg, n, buf)
        return buf.raw[:n]


class Ref:
    """The real reference. Process-global state inside the library: one index at a time."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise RuntimeError("oracle/_ref/liblfref.so is missing: run `make -C oracle ref` where /root/reference exists")
        L = self.L = C.CDLL(REF_SO)
        L.ref_map_file.restype = C.c_double
        L.ref_map_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.ref_map_mem.restype = C.c_double
        L.ref_map_mem.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                  C.c_char_p, C.c_int]
        L.ref_set_params.argtypes = [C.c_int] * 6 + [C.c_double] * 3 + [C.c_int]
        L.ref_seed.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32),
                               C.c_void_p, C.POINTER(C.c_uint32)]
        L.ref_chain_n2.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.ref_chain_clasp.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        L.ref_edlib.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int),
                                C.c_void_p, C.POINTER(C.c_int)]
        L.ref_ksw_extend2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7 + \
                                     [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_genome_len.restype = C.c_uint32
        self.params = default_params()

    def index_build(self, fasta: str) -> None:
        if self.L.ref_index_build(fasta.encode()) != 0:
            raise RuntimeError("reference index build failed")

    def load(self, fasta: str) -> None:
        if self.L.ref_load(fasta.encode()) != 0:
            raise RuntimeError("reference index load failed")

    def set_params(self, p: Params, cmdline: str = "") -> None:
        self.params = p
        self.L.ref_set_params(p.min_anchor_len, p.sampling_count, p.max_map, p.min_read_len, p.max_ref_hits,
                              p.chain_alg, p.chain_reward, p.chain_penalty, p.gap_penalty, p.threads)
        self.L.ref_set_cmdline(cmdline.encode())

    def threads(self) -> int:
        return self.L.ref_threads()

    def map_mem(self, names, seqs, quals=None, header: bool = False):
        """returns (SAM bytes, mapSeqMT seconds)"""
        n = len(names)
        q = _cstr_array(quals) if quals is not None else None
        with tempfile.NamedTemporaryFile(suffix=".sam", delete=False) as tf:
            path = tf.name
        try:
            secs = self.L.ref_map_mem(n, _cstr_array(names), _cstr_array(seqs), q, path.encode(), 0 if header else 1)
            with open(path, "rb") as fh:
                sam = fh.read()
        finally:
            os.unlink(path)
        return sam, secs

    def map_file(self, reads_path: str, header: bool = True):
        with tempfile.NamedTemporaryFile(suffix=".sam", delete=False) as tf:
            path = tf.name
        try:
            secs = self.L.ref_map_file(reads_path.encode(), path.encode(), 0 if header else 1)
            with open(path, "rb") as fh:
                sam = fh.read()
        finally:
            os.unlink(path)
        return sam, secs

    def seed(self, seq: bytes):
        p = self.params
        cap = p.sampling_count * p.max_ref_hits
        F = np.zeros((cap, 3), dtype=np.uint32)
        R = np.zeros((cap, 3), dtype=np.uint32)
        nF, nR = C.c_uint32(), C.c_uint32()
        buf = C.create_string_buffer(seq + b"\0" * 64)   # NUL-terminated like a Read block
        self.L.ref_seed(buf, len(seq), p.sampling_count, F.ctypes.data, C.byref(nF), R.ctypes.data, C.byref(nR))
        return F[:nF.value].copy(), R[:nR.value].copy()

    def stage_windows(self, seqs):
        """the reference's own findTopWins_coarse / _fine + alignWin per read (ref_harness.cpp: ref_stage_windows).
        -> list of dicts: mode, coarse (k,4) u32 [tStart,tEnd,isRev,score bits], wins (k,4) u32, maps int32 words
        ({totalScore, n_records, 7 ints per record} per window)"""
        p = self.params
        L = self.L
        L.ref_stage_windows.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                                        C.POINTER(C.c_int), C.c_void_p, C.c_int]
        out = []
        L.ref_stage_begin()
        try:
            for sq in seqs:
                coarse = np.zeros((p.max_map + 1, 4), dtype=np.uint32); wins = np.zeros((p.max_map + 1, 4), dtype=np.uint32)
                maps = np.zeros(1 << 16, dtype=np.int32)
                mode, nc, nw = C.c_int(), C.c_int(), C.c_int()
                buf = C.create_string_buffer(sq + b"\0" * 64)
                words = L.ref_stage_windows(buf, len(sq), C.byref(mode), coarse.ctypes.data, C.byref(nc), wins.ctypes.data, C.byref(nw),
                                            maps.ctypes.data, maps.size)
                out.append(dict(mode=mode.value, coarse=coarse[:nc.value].copy(), wins=wins[:nw.value].copy(), maps=maps[:words].copy()))
        finally:
            L.ref_stage_end()
        return out

    def chain_n2(self, seeds: np.ndarray):
        s = np.ascontiguousarray(seeds, dtype=np.uint32).copy()
        n = len(s)
        out = np.zeros((max(n, 1), 3), dtype=np.uint32)
        cl, sc = C.c_uint32(), C.c_float()
        self.L.ref_chain_n2(s.ctypes.data, n, out.ctypes.data, C.byref(cl), C.byref(sc))
        return s, out[:cl.value].copy(), sc.value

    def chain_clasp(self, seeds: np.ndarray):
        s = np.ascontiguousarray(seeds, dtype=np.uint32).copy()
        n = len(s)
        out = np.zeros((max(n, 1), 3), dtype=np.uint32)
        cl, sc = C.c_uint32(), C.c_float()
        self.L.ref_chain_clasp(s.ctypes.data, n, out.ctypes.data, C.byref(cl), C.byref(sc))
        return out[:cl.value].copy(), sc.value

    def edlib(self, q: bytes, t: bytes, mode: int):
        ops = np.zeros(len(q) + len(t) + 1, dtype=np.uint8)
        end, nops = C.c_int(), C.c_int()
        ed = self.L.ref_edlib(q, len(q), t, len(t), mode, C.byref(end), ops.ctypes.data, C.byref(nops))
        return ed, end.value, ops[:nops.value].copy()

    def ksw_extend2(self, q: np.ndarray, t: np.ndarray, o_del, e_del, o_ins, e_ins, w, zdrop, h0):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        qle, tle = C.c_int(), C.c_int()
        sc = self.L.ref_ksw_extend2(len(q), q.ctypes.data, len(t), t.ctypes.data, o_del, e_del, o_ins, e_ins,
                                    w, zdrop, h0, C.byref(qle), C.byref(tle))
        return sc, qle.value, tle.value
