"""ctypes binding of liblfgpu.so (include/lordfast_amd.h).  No torch types cross this boundary."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LfError(RuntimeError):
    pass


def lib_path() -> str:
    return os.path.join(_HERE, "liblfgpu.so")


def build_library(quiet: bool = True) -> None:
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class Params(C.Structure):
    _fields_ = [("min_anchor_len", C.c_int), ("sampling_count", C.c_int), ("max_map", C.c_int),
                ("min_read_len", C.c_int), ("max_ref_hits", C.c_int), ("chain_alg", C.c_int),
                ("chain_reward", C.c_double), ("chain_penalty", C.c_double), ("gap_penalty", C.c_double),
                ("threads", C.c_int), ("read_group_id", C.c_char * 256), ("read_group", C.c_char * 1000)]


def default_params(**kw) -> Params:
    p = Params(14, 1000, 10, 1000, 1000, 0, 9.3, 11.4, 0.15, 0, b"", b"")
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Seeds(C.Structure):
    _fields_ = [("n_reads", C.c_int), ("offF", C.POINTER(C.c_uint64)), ("offR", C.POINTER(C.c_uint64)),
                ("F", C.c_void_p), ("R", C.c_void_p),
                ("n_cache", C.c_uint64), ("n_occblk", C.c_uint64), ("n_sa", C.c_uint64), ("n_readbytes", C.c_uint64),
                ("ms_search", C.c_float), ("ms_accept", C.c_float), ("ms_locate", C.c_float)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ms_total", "ms_seed", "ms_vote", "ms_chain", "ms_extend", "ms_sam")] + \
               [(n, C.c_float) for n in ("ms_k_search", "ms_k_accept", "ms_k_locate", "ms_k_chain", "ms_k_edlib", "ms_k_ksw")] + \
               [(n, C.c_uint64) for n in ("n_reads", "n_bases", "n_seeds", "n_chain_problems", "n_edlib_problems",
                                          "n_ksw_problems", "n_cache", "n_occblk", "n_sa", "n_readbytes", "ext_bytes",
                                          "edlib_launches", "search_launches", "locate_launches")] + \
               [("ms_render", C.c_double), ("ms_k_render", C.c_float), ("ms_k_vote", C.c_float),
                ("render_bytes", C.c_uint64), ("render_launches", C.c_uint64),
                ("ops_bytes", C.c_uint64), ("n_req_seeds", C.c_uint64), ("n_tie_requests", C.c_uint64), ("dp_block_steps", C.c_uint64),
                ("ksw_bytes", C.c_uint64)] + \
               [(n, C.c_float) for n in ("ms_k_rsweep", "ms_k_tb", "ms_k_hirsch", "ms_k_bin")] + \
               [(n, C.c_uint64) for n in ("hirsch_bytes", "n_host_waits", "n_chunks", "hirsch_max_rows", "hirsch_banded_nodes", "hirsch_unbanded_nodes", "n_stale_first_windows")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def lib():
    """Load liblfgpu.so (fails loudly if it was not built: there is no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise LfError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(make -C lordfast_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(path)
    L.lf_last_error.restype = C.c_char_p
    L.lf_index_load.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
    L.lf_index_free.argtypes = [C.c_void_p]
    L.lf_index_genome_len.argtypes = [C.c_void_p]
    L.lf_index_genome_len.restype = C.c_uint32
    L.lf_seed_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_char_p, C.c_void_p,
                                C.POINTER(C.POINTER(Seeds))]
    L.lf_seeds_free.argtypes = [C.POINTER(Seeds)]
    L.lf_free.argtypes = [C.c_void_p]
    if hasattr(L, "lf_index_build"):
        L.lf_index_build.argtypes = [C.c_char_p, C.c_int]
    if hasattr(L, "lf_edlib_batch"):
        L.lf_edlib_batch.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    if hasattr(L, "lf_chain_n2_batch"):
        L.lf_chain_n2_batch.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int]
    if hasattr(L, "lf_chain_clasp_batch"):
        L.lf_chain_clasp_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    if hasattr(L, "lf_ksw_extend2_batch"):
        L.lf_ksw_extend2_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    if hasattr(L, "lf_map_batch"):
        L.lf_map_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_char_p),
                                   C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t), C.POINTER(Stats)]
        L.lf_map_batch_into.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_char_p),
                                        C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(Stats)]
        L.lf_sam_header.restype = C.c_void_p
        L.lf_sam_header.argtypes = [C.c_void_p, C.POINTER(Params), C.c_char_p]
    _LIB = L
    return L


def read_file(path: str, batch_reads: int = 0, batch_bases: int = 0):
    """FASTA / FASTQ (plain or gzip) -> list of batches [(names, seqs, quals)] parsed by the library's reader
    (quals: b"" for FASTA records)."""
    L = lib()
    L.lf_reads_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.lf_reads_next.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    L.lf_reads_close.argtypes = [C.c_void_p]
    L.lf_read_batch_free.argtypes = [C.c_void_p]
    L.lf_read_batch_size.argtypes = [C.c_void_p]
    for f in ("lf_read_batch_names", "lf_read_batch_seqs", "lf_read_batch_quals"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.POINTER(C.c_char_p)
    h = C.c_void_p()
    _check(L.lf_reads_open(path.encode(), C.byref(h)), "lf_reads_open")
    out = []
    try:
        while True:
            b = C.c_void_p()
            _check(L.lf_reads_next(h, batch_reads, batch_bases, C.byref(b)), "lf_reads_next")
            if not b.value:
                break
            n = L.lf_read_batch_size(b)
            na, sa, qa = L.lf_read_batch_names(b), L.lf_read_batch_seqs(b), L.lf_read_batch_quals(b)
            out.append(([na[i] for i in range(n)], [sa[i] for i in range(n)], [qa[i] for i in range(n)]))
            L.lf_read_batch_free(b)
    finally:
        L.lf_reads_close(h)
    return out


class DeviceBuffer:
    """bytes in HBM of one device (lf_device_alloc): what lf_map_batch_dev reads from / writes to when the caller has no
    HIP-aware framework of its own"""

    def __init__(self, nbytes: int, device: int = 0, data: bytes | None = None):
        L = lib()
        L.lf_device_alloc.restype = C.c_void_p
        L.lf_device_alloc.argtypes = [C.c_int, C.c_size_t]
        L.lf_device_free.argtypes = [C.c_int, C.c_void_p]
        L.lf_device_copy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        self.L, self.device, self.nbytes = L, device, nbytes
        self.ptr = L.lf_device_alloc(device, nbytes)
        if not self.ptr:
            raise LfError(f"lf_device_alloc failed: {L.lf_last_error().decode(errors='replace')}")
        if data is not None:
            self.upload(data)

    def upload(self, data: bytes, offset: int = 0):
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        _check(self.L.lf_device_copy(self.device, C.c_void_p(self.ptr + offset), buf, len(data)), "lf_device_copy")

    def download(self, nbytes: int | None = None, offset: int = 0) -> bytes:
        n = self.nbytes - offset if nbytes is None else nbytes
        buf = (C.c_char * max(1, n))()
        _check(self.L.lf_device_copy(self.device, buf, C.c_void_p(self.ptr + offset), n), "lf_device_copy")
        return bytes(buf[:n])

    def free(self):
        if self.ptr:
            self.L.lf_device_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:                                                # noqa: BLE001
            pass


def device_count() -> int:
    return lib().lf_device_count()


def _check(rc: int, what: str):
    if rc != 0:
        raise LfError(f"{what} failed (rc={rc}): {lib().lf_last_error().decode(errors='replace')}")


def _seeds_to_triples(raw: np.ndarray) -> np.ndarray:
    """Seed_t {u32 tPos; u32 qPos:20, len:12} -> (n,3) u32 [tPos,qPos,len]"""
    raw = raw.reshape(-1, 2)
    return np.stack([raw[:, 0], raw[:, 1] & 0xFFFFF, raw[:, 1] >> 20], axis=1).astype(np.uint32)


def _triples_to_seeds(tr: np.ndarray) -> np.ndarray:
    tr = np.ascontiguousarray(tr, dtype=np.uint32).reshape(-1, 3)
    out = np.empty((len(tr), 2), dtype=np.uint32)
    out[:, 0] = tr[:, 0]
    out[:, 1] = (tr[:, 1] & 0xFFFFF) | ((tr[:, 2] & 0xFFF) << 20)
    return out


def _cstr_array(items):
    arr = (C.c_char_p * len(items))()
    arr[:] = [x if isinstance(x, bytes) else x.encode() for x in items]
    return arr


def _concat(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    return b"".join(seqs), off


def index_build(contigs_or_fasta, fasta_path: str | None = None, device: int = 0) -> str:
    """Build the reference's index files next to a FASTA on the GPU.  `contigs_or_fasta` is either a path or a
    list of (name, uint8 array) that is first written to `fasta_path`."""
    if isinstance(contigs_or_fasta, (str, bytes)):
        fa = contigs_or_fasta if isinstance(contigs_or_fasta, str) else contigs_or_fasta.decode()
    else:
        from . import synth
        fa = fasta_path
        synth.write_fasta(fa, contigs_or_fasta)
    _check(lib().lf_index_build(fa.encode(), device), "lf_index_build")
    return fa


class SamBuffer:
    """SAM text owned by the C library (malloc'd by lf_map_batch)."""

    def __init__(self, L, ptr, n):
        self.L, self.ptr, self.n = L, ptr, n

    def __len__(self):
        return self.n

    def view(self) -> memoryview:
        if self.n == 0:
            return memoryview(b"")
        return memoryview((C.c_char * self.n).from_address(self.ptr.value)).cast("B")

    def tobytes(self) -> bytes:
        return bytes(self.view())

    def head(self, n_bytes: int) -> bytes:
        return bytes(self.view()[:min(n_bytes, self.n)])

    def free(self):
        if self.ptr:
            self.L.lf_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


class ReadBatch:
    """lf_batch_create: the reads as the caller has them (the Python byte strings are kept alive by this object) plus what every mapping call would
    otherwise make per call -- lengths and, for reads of at least min_read_len bases, the bit planes the batch crosses the host link as."""

    def __init__(self, names, seqs, quals=None, min_read_len: int = 1000, threads: int = 0, seq_lens=None):
        L = lib()
        self.names, self.seqs, self.quals = list(names), list(seqs), (list(quals) if quals is not None else None)
        self._na, self._sa = _cstr_array(self.names), _cstr_array(self.seqs)
        self._qa = _cstr_array(self.quals) if self.quals is not None else None
        sl = np.ascontiguousarray(seq_lens, dtype=np.uint32) if seq_lens is not None else None
        L.lf_batch_create.restype = C.c_void_p
        L.lf_batch_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self.h = L.lf_batch_create(len(self.names), self._na, self._sa, self._qa, sl.ctypes.data if sl is not None else None, min_read_len, threads)
        if not self.h:
            raise LfError("lf_batch_create: " + L.lf_last_error().decode(errors="replace"))
        self.L = L

    def __len__(self):
        self.L.lf_batch_size.argtypes = [C.c_void_p]
        return int(self.L.lf_batch_size(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.lf_batch_free.argtypes = [C.c_void_p]
            self.L.lf_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LordFast:
    """An FM-index resident in one GPU's HBM + the batch entry points."""

    FULL_SA = 1

    def __init__(self, prefix: str, device: int = 0, full_sa: bool = True):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.lf_index_load(prefix.encode(), device, self.FULL_SA if full_sa else 0, C.byref(h)), "lf_index_load")
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.L.lf_index_free(self.h)
            self.h = None

    def describe(self) -> str:
        buf = C.create_string_buffer(1024)
        self.L.lf_index_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        _check(self.L.lf_index_describe(self.h, buf, 1024), "lf_index_describe")
        return buf.value.decode()

    def genome_len(self) -> int:
        return self.L.lf_index_genome_len(self.h)

    # ---- stage 1
    def seed_batch(self, seqs, params: Params | None = None):
        """-> (list of F triples, list of R triples, info dict)"""
        p = params or default_params()
        cat, off = _concat(seqs)
        out = C.POINTER(Seeds)()
        _check(self.L.lf_seed_batch(self.h, C.byref(p), len(seqs), cat, off.ctypes.data, C.byref(out)), "lf_seed_batch")
        s = out.contents
        n = s.n_reads
        offF = np.ctypeslib.as_array(s.offF, shape=(n + 1,)).copy()
        offR = np.ctypeslib.as_array(s.offR, shape=(n + 1,)).copy()
        F = np.frombuffer(C.string_at(s.F, int(offF[-1]) * 8), dtype=np.uint32) if offF[-1] else np.zeros(0, np.uint32)
        R = np.frombuffer(C.string_at(s.R, int(offR[-1]) * 8), dtype=np.uint32) if offR[-1] else np.zeros(0, np.uint32)
        Ft, Rt = _seeds_to_triples(F), _seeds_to_triples(R)
        info = dict(n_cache=s.n_cache, n_occblk=s.n_occblk, n_sa=s.n_sa, n_readbytes=s.n_readbytes,
                    ms_search=s.ms_search, ms_accept=s.ms_accept, ms_locate=s.ms_locate)
        self.L.lf_seeds_free(out)
        Fl = [Ft[int(offF[i]):int(offF[i + 1])] for i in range(n)]
        Rl = [Rt[int(offR[i]):int(offR[i + 1])] for i in range(n)]
        return Fl, Rl, info

    # ---- whole path
    def map_batch(self, names, seqs, quals=None, params: Params | None = None, copy: bool = True):
        """-> (SAM records, stats dict).  copy=False returns a SamBuffer (zero-copy view of the C buffer; SAM text of a
        100k-read batch is > 2 GiB)."""
        p = params or default_params()
        sam = C.c_void_p()
        ln = C.c_size_t()
        st = Stats()
        q = _cstr_array(quals) if quals is not None else None
        _check(self.L.lf_map_batch(self.h, C.byref(p), len(names), _cstr_array(names), _cstr_array(seqs), q,
                                   C.byref(sam), C.byref(ln), C.byref(st)), "lf_map_batch")
        buf = SamBuffer(self.L, sam, ln.value)
        if copy:
            out = buf.tobytes()
            buf.free()
            return out, st.as_dict()
        return buf, st.as_dict()

    def map_batch_into(self, names, seqs, out_ptr: int, out_cap: int, quals=None, params: Params | None = None,
                       name_arr=None, seq_arr=None, seq_lens=None):
        """SAM text into a caller-owned buffer (address + capacity), e.g. a pinned torch tensor reused across batches.
        -> (length, stats).  name_arr / seq_arr: pre-built ctypes arrays (saves rebuilding them per call)."""
        p = params or default_params()
        ln = C.c_size_t()
        st = Stats()
        q = _cstr_array(quals) if quals is not None else None
        na = name_arr if name_arr is not None else _cstr_array(names)
        sa = seq_arr if seq_arr is not None else _cstr_array(seqs)
        if seq_lens is not None:              # uint32 numpy array of len(seqs[i]): the library skips its strlen pass
            sl = np.ascontiguousarray(seq_lens, dtype=np.uint32)
            if sl.shape != (len(names),):
                raise LfError(f"seq_lens has shape {sl.shape}, expected ({len(names)},)")
            self.L.lf_map_batch_into_lens.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Stats)]
            _check(self.L.lf_map_batch_into_lens(self.h, C.byref(p), len(names), na, sa, q, sl.ctypes.data, C.c_void_p(out_ptr), out_cap,
                                                 C.byref(ln), C.byref(st)), "lf_map_batch_into_lens")
            return ln.value, st.as_dict()
        _check(self.L.lf_map_batch_into(self.h, C.byref(p), len(names), na, sa, q, C.c_void_p(out_ptr), out_cap,
                                        C.byref(ln), C.byref(st)), "lf_map_batch_into")
        return ln.value, st.as_dict()

    def map_batch_from(self, batch: "ReadBatch", out_ptr: int, out_cap: int, params: Params | None = None):
        """lf_map_batch_from: a mapper-ready batch (ReadBatch: lengths and bit planes made when it was created) -> SAM text in a caller-owned
        buffer.  -> (length, stats)"""
        p = params or default_params()
        ln = C.c_size_t()
        st = Stats()
        self.L.lf_map_batch_from.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Stats)]
        _check(self.L.lf_map_batch_from(self.h, C.byref(p), batch.h, C.c_void_p(out_ptr), out_cap, C.byref(ln), C.byref(st)), "lf_map_batch_from")
        return ln.value, st.as_dict()

    def map_batch_dev(self, name_arr, d_seqs: int, seq_off, seq_lens, out_ptr: int, out_cap: int, out_is_device: bool = True,
                      d_quals: int = 0, params: Params | None = None):
        """lf_map_batch_dev: the bases are already in HBM (address d_seqs, read i at seq_off[i], seq_lens[i] long) and the SAM
        text is written to `out_ptr` (device memory unless out_is_device is False).  name_arr: ctypes array of C strings
        (host).  -> (length, stats)"""
        p = params or default_params()
        so = np.ascontiguousarray(seq_off, dtype=np.uint64)
        sl = np.ascontiguousarray(seq_lens, dtype=np.uint32)
        n = len(sl)
        if so.shape != (n,):
            raise LfError(f"seq_off has shape {so.shape}, expected ({n},)")
        ln, st = C.c_size_t(), Stats()
        self.L.lf_map_batch_dev.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(Stats)]
        _check(self.L.lf_map_batch_dev(self.h, C.byref(p), n, name_arr, C.c_void_p(d_seqs), so.ctypes.data, sl.ctypes.data,
                                       C.c_void_p(d_quals) if d_quals else None, C.c_void_p(out_ptr), out_cap, 1 if out_is_device else 0,
                                       C.byref(ln), C.byref(st)), "lf_map_batch_dev")
        return ln.value, st.as_dict()

    def map_stages(self, seqs, params: Params | None = None):
        """lf_map_stages_batch: per read {mode, wins (k,4) u32 [tStart,tEnd,isRev,score bits], maps int32 words
        ({totalScore, n_records, 7 ints per record} per window)} -- the layout of tests/golden/stages_windows.npz"""
        class Win(C.Structure):
            _fields_ = [("tStart", C.c_uint32), ("tEnd", C.c_uint32), ("isReverse", C.c_uint32), ("score", C.c_float),
                        ("totalScore", C.c_int32), ("n_records", C.c_uint32), ("rec0", C.c_uint32)]

        class Rec(C.Structure):
            _fields_ = [("pos", C.c_uint32), ("posEnd", C.c_uint32), ("qStart", C.c_uint32), ("qEnd", C.c_uint32),
                        ("flag", C.c_int32), ("alnScore", C.c_int32), ("nmCount", C.c_int32)]

        class Stages(C.Structure):
            _fields_ = [("n_reads", C.c_int), ("mode", C.POINTER(C.c_uint8)), ("win0", C.POINTER(C.c_uint32)), ("n_wins", C.c_uint32),
                        ("wins", C.POINTER(Win)), ("n_recs", C.c_uint32), ("recs", C.POINTER(Rec))]
        p = params or default_params()
        out = C.POINTER(Stages)()
        self.L.lf_map_stages_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(Stages))]
        self.L.lf_stages_free.argtypes = [C.POINTER(Stages)]
        _check(self.L.lf_map_stages_batch(self.h, C.byref(p), len(seqs), _cstr_array(seqs), C.byref(out)), "lf_map_stages_batch")
        S = out.contents
        res = []
        for i in range(S.n_reads):
            wins, maps = [], []
            for w in range(S.win0[i], S.win0[i + 1]):
                W = S.wins[w]
                wins.append([W.tStart, W.tEnd, W.isReverse, int(np.float32(W.score).view(np.uint32))])
                maps += [W.totalScore, W.n_records]
                for j in range(W.rec0, W.rec0 + W.n_records):
                    R = S.recs[j]
                    maps += [R.pos, R.posEnd, R.qStart, R.qEnd, R.flag, R.alnScore, R.nmCount]
            res.append(dict(mode=int(S.mode[i]), wins=np.array(wins, dtype=np.uint32).reshape(-1, 4),
                            maps=np.array(maps, dtype=np.int64).astype(np.int32)))
        self.L.lf_stages_free(out)
        return res

    def map_file(self, reads_path: str, out_path: str, params: Params | None = None, header: bool = True, cmdline: str = "",
                 batch_reads: int = 0):
        """reads file -> SAM file (lf_map_file): the reader runs ahead of the GPU.  -> stats dict"""
        p = params or default_params()
        st = Stats()
        self.L.lf_map_file.argtypes = [C.c_void_p, C.POINTER(Params), C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(Stats)]
        _check(self.L.lf_map_file(self.h, C.byref(p), reads_path.encode(), out_path.encode(), 0 if header else 1, cmdline.encode(),
                                  batch_reads, C.byref(st)), "lf_map_file")
        return st.as_dict()

    def sam_header(self, cmdline: str, params: Params | None = None) -> bytes:
        p = params or default_params()
        ptr = self.L.lf_sam_header(self.h, C.byref(p), cmdline.encode())
        out = C.string_at(ptr)
        self.L.lf_free(ptr)
        return out


# ---- index-free batch primitives -------------------------------------------------------------

def edlib_batch(qs, ts, modes, device: int = 0):
    """-> list of (edit_distance, end_location, ops ndarray), kernel ms"""
    L = lib()
    n = len(qs)
    qcat, qoff = _concat(qs)
    tcat, toff = _concat(ts)
    mode = np.ascontiguousarray(modes, dtype=np.uint8)
    ed = np.zeros(n, dtype=np.int32)
    end = np.zeros(n, dtype=np.int32)
    ops = np.zeros(int(qoff[-1] + toff[-1]) + 1, dtype=np.uint8)
    ops_len = np.zeros(n, dtype=np.uint32)
    ms = C.c_float()
    _check(L.lf_edlib_batch(n, qcat, qoff.ctypes.data, tcat, toff.ctypes.data, mode.ctypes.data, ed.ctypes.data,
                            end.ctypes.data, ops.ctypes.data, ops_len.ctypes.data, device, C.byref(ms)), "lf_edlib_batch")
    res = []
    for i in range(n):
        o = int(qoff[i] + toff[i])
        res.append((int(ed[i]), int(end[i]), ops[o:o + int(ops_len[i])].copy()))
    return res, ms.value


def chain_n2_batch(windows, params: Params | None = None, device: int = 0):
    """windows: list of (n,3) triples. -> list of (reordered triples, chain triples, score)"""
    L = lib()
    p = params or default_params()
    off = np.zeros(len(windows) + 1, dtype=np.uint64)
    if len(windows):
        off[1:] = np.cumsum([len(w) for w in windows], dtype=np.uint64)
    allw = np.concatenate([np.asarray(w, dtype=np.uint32).reshape(-1, 3) for w in windows]) if len(windows) else np.zeros((0, 3), np.uint32)
    seeds = _triples_to_seeds(allw)
    nt = max(1, len(allw))
    chain_idx = np.zeros(nt, dtype=np.uint32)
    chain_len = np.zeros(max(1, len(windows)), dtype=np.uint32)
    score = np.zeros(max(1, len(windows)), dtype=np.float32)
    _check(L.lf_chain_n2_batch(C.byref(p), len(windows), seeds.ctypes.data, off.ctypes.data, chain_idx.ctypes.data,
                               chain_len.ctypes.data, score.ctypes.data, device), "lf_chain_n2_batch")
    tr = _seeds_to_triples(seeds.reshape(-1))
    out = []
    for w in range(len(windows)):
        a, b = int(off[w]), int(off[w + 1])
        srt = tr[a:b]
        idx = chain_idx[a:a + int(chain_len[w])]
        out.append((srt, srt[idx], float(score[w])))
    return out


def chain_clasp_batch(windows, device: int = 0):
    """windows: list of (n,3) triples (tPos, qPos, len) in selection order. -> list of (chain triples, score)"""
    L = lib()
    off = np.zeros(len(windows) + 1, dtype=np.uint64)
    if len(windows):
        off[1:] = np.cumsum([len(w) for w in windows], dtype=np.uint64)
    allw = np.concatenate([np.asarray(w, dtype=np.uint32).reshape(-1, 3) for w in windows]) if len(windows) else np.zeros((0, 3), np.uint32)
    seeds = _triples_to_seeds(allw)
    out_seeds = np.zeros_like(seeds) if len(seeds) else np.zeros(1, dtype=seeds.dtype)
    chain_len = np.zeros(max(1, len(windows)), dtype=np.uint32)
    score = np.zeros(max(1, len(windows)), dtype=np.float32)
    _check(L.lf_chain_clasp_batch(len(windows), seeds.ctypes.data, off.ctypes.data, out_seeds.ctypes.data,
                                  chain_len.ctypes.data, score.ctypes.data, device), "lf_chain_clasp_batch")
    tr = _seeds_to_triples(out_seeds.reshape(-1))
    return [(tr[int(off[w]):int(off[w]) + int(chain_len[w])], float(score[w])) for w in range(len(windows))]


def ksw_extend2_batch(qs, ts, prms, device: int = 0):
    L = lib()
    n = len(qs)
    qcat, qoff = _concat([np.ascontiguousarray(q, dtype=np.uint8).tobytes() for q in qs])
    tcat, toff = _concat([np.ascontiguousarray(t, dtype=np.uint8).tobytes() for t in ts])
    prm = np.ascontiguousarray(prms, dtype=np.int32).reshape(n, 7)
    sc = np.zeros(n, np.int32); qle = np.zeros(n, np.int32); tle = np.zeros(n, np.int32)
    _check(L.lf_ksw_extend2_batch(n, qcat, qoff.ctypes.data, tcat, toff.ctypes.data, prm.ctypes.data, sc.ctypes.data,
                                  qle.ctypes.data, tle.ctypes.data, device), "lf_ksw_extend2_batch")
    return [(int(sc[i]), int(qle[i]), int(tle[i])) for i in range(n)]


def map_batch_multi(handles, names, seqs, quals=None, params: Params | None = None):
    """lf_map_batch_multi: one batch spread over several LordFast handles (index replicas, normally one per GPU).
    -> (SAM records in input order, stats dict)"""
    L = lib()
    L.lf_map_batch_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Params), C.c_int, C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats)]
    p = params or default_params()
    hs = (C.c_void_p * len(handles))(*[h.h for h in handles])
    na, sa = _cstr_array(names), _cstr_array(seqs)
    qa = _cstr_array(quals) if quals is not None else None
    out, ln, st = C.c_void_p(), C.c_size_t(), Stats()
    _check(L.lf_map_batch_multi(hs, len(handles), C.byref(p), len(seqs), na, sa, qa, None, None, 0,
                                C.byref(out), C.byref(ln), C.byref(st)), "lf_map_batch_multi")
    sam = C.string_at(out, ln.value) if ln.value else b""
    L.lf_free(out)
    return sam, st.as_dict()


def map_batch_multi_into(handles, names, out_ptr: int, out_cap: int, params: Params | None = None, name_arr=None, seq_arr=None,
                         seq_lens=None, seqs=None):
    """lf_map_batch_multi into a caller-owned (pinned) buffer. -> (length, stats dict)"""
    L = lib()
    L.lf_map_batch_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Params), C.c_int, C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats)]
    p = params or default_params()
    hs = (C.c_void_p * len(handles))(*[h.h for h in handles])
    na = name_arr if name_arr is not None else _cstr_array(names)
    sa = seq_arr if seq_arr is not None else _cstr_array(seqs)
    sl = np.ascontiguousarray(seq_lens, dtype=np.uint32) if seq_lens is not None else None
    ln, st = C.c_size_t(), Stats()
    _check(L.lf_map_batch_multi(hs, len(handles), C.byref(p), len(names), na, sa, None, sl.ctypes.data if sl is not None else None,
                                C.c_void_p(out_ptr), out_cap, None, C.byref(ln), C.byref(st)), "lf_map_batch_multi")
    return ln.value, st.as_dict()
