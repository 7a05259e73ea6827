#!/usr/bin/env python3
"""reader_bench.py [n_reads] -- the library's reader alone (no GPU): reads/s and GB/s of text for plain FASTA / FASTQ, BGZF and
one-stream gzip FASTQ of ~15 kbp reads.  One JSON line per container."""
import ctypes as C
import gzip
import json
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lordfast_amd as la  # noqa: E402


def bgzf(data: bytes, block: int = 0xff00, level: int = 6) -> bytes:
    out = []
    for a in list(range(0, len(data), block)) + [None]:
        chunk = data[a:a + block] if a is not None else b""
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
                   + body + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs_fq, recs_fa = [], []
    for i in range(n):
        ln = int(max(1000, rng.lognormal(np.log(15000) - 0.06, 0.35)))
        s = acgt[rng.integers(0, 4, size=ln, dtype=np.uint8)].tobytes()
        q = (rng.integers(0, 20, size=ln, dtype=np.uint8) + 40).tobytes()         # PacBio-like narrow quality range
        recs_fq.append(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
        recs_fa.append(b">r%d\n" % i + s + b"\n")
    fq, fa = b"".join(recs_fq), b"".join(recs_fa)
    del recs_fq, recs_fa
    d = tempfile.mkdtemp(prefix="lf_rb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    files = {"fasta": fa, "fastq": fq, "fastq.bgzf": bgzf(fq), "fastq.gz": gzip.compress(fq, 6)}
    L = la.lib()
    L.lf_reads_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.lf_reads_next.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    L.lf_reads_close.argtypes = [C.c_void_p]
    L.lf_read_batch_free.argtypes = [C.c_void_p]
    L.lf_read_batch_size.argtypes = [C.c_void_p]
    try:
        for name, data in files.items():
            path = os.path.join(d, "reads." + name)
            with open(path, "wb") as fh:
                fh.write(data)
            best = None
            for rep in range(3):
                h = C.c_void_p()
                assert L.lf_reads_open(path.encode(), C.byref(h)) == 0
                t0 = time.perf_counter(); c0 = sum(os.times()[:2]); got = 0
                while True:
                    b = C.c_void_p()
                    assert L.lf_reads_next(h, 50000, 768 << 20, C.byref(b)) == 0
                    if not b.value:
                        break
                    got += L.lf_read_batch_size(b)
                    L.lf_read_batch_free(b)
                dt = time.perf_counter() - t0; cpu = sum(os.times()[:2]) - c0
                L.lf_reads_close(h)
                assert got == n, (name, got)
                if best is None or dt < best[0]:
                    best = (dt, cpu)
            text = len(fq) if name.startswith("fastq") else len(fa)
            print(json.dumps({"container": name, "reads": n, "file_MB": round(len(data) / 1e6, 1), "text_MB": round(text / 1e6, 1), "seconds": round(best[0], 3),
                              "reads_per_s": round(n / best[0]), "text_GB_per_s": round(text / best[0] / 1e9, 2), "cpu_seconds": round(best[1], 2),
                              "cpus": os.cpu_count()}), flush=True)
            os.remove(path)
    finally:
        os.rmdir(d)


if __name__ == "__main__":
    main()
