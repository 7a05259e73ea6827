#!/usr/bin/env python3
"""cli_bench.py -- end-to-end rate of the process interface: `lordfast --search ref --seq reads.{fa,fq.gz} -o out`.
Uses bench.py's cached synthetic genome / reads (same recipe as config C2).  Prints one JSON line per run."""
import argparse, gzip, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--genome-mbp", type=float, default=3100)
ap.add_argument("--reads", type=int, default=100000)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--gz", action="store_true")
ap.add_argument("--fastq", action="store_true")
ap.add_argument("--bgzf", action="store_true", help="blocked gzip (what bgzip writes): members with a size field, inflated by several threads")
ap.add_argument("--copies", type=int, default=1, help="write the read set this many times (names suffixed): longer files from one cached set")
ap.add_argument("--devnull-only", action="store_true", help="no SAM file on disk (large read sets)")
a = ap.parse_args()
sys.argv = [sys.argv[0], "--genome-mbp", str(a.genome_mbp), "--reads", str(a.reads)]
args = bench.parse()
fa, contigs = bench.ensure_index(args, 0)
names, seqs = bench.make_reads(args, contigs, fa, 0)
d = os.path.dirname(fa)
ext = ("fq" if a.fastq else "fa") + (".gz" if a.gz else "") + (".bgzf.gz" if a.bgzf else "")
rp = os.path.join(d, f"reads_{a.reads}x{a.copies}.{ext}")
if not os.path.exists(rp):
    import struct, zlib
    import numpy as np
    t0 = time.time()
    rng = np.random.default_rng(5)

    class Bgzf:                      # gzip members of <= 0xff00 text bytes with the 'BC' size subfield + the 28-byte end marker
        def __init__(self, path):
            self.fh, self.buf = open(path, "wb"), bytearray()

        def block(self, chunk):
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = c.compress(bytes(chunk)) + c.flush()
            self.fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
                          + body + struct.pack("<II", zlib.crc32(bytes(chunk)), len(chunk)))

        def write(self, b):
            self.buf += b
            while len(self.buf) >= 0xff00:
                self.block(self.buf[:0xff00]); del self.buf[:0xff00]

        def __enter__(self):
            return self

        def __exit__(self, *x):
            if self.buf:
                self.block(self.buf)
            self.block(b""); self.fh.close()
    op = (lambda p_: Bgzf(p_)) if a.bgzf else (lambda p_: gzip.open(p_, "wb", compresslevel=1)) if a.gz else (lambda p_: open(p_, "wb"))
    with op(rp) as fh:
        for c in range(a.copies):
            for n, s in zip(names, seqs):
                nm = n if c == 0 else n + b"_c%d" % c
                if a.fastq:          # qualities: 20 values, like a PacBio / ONT file (not one letter: that would inflate at memset speed)
                    q = (rng.integers(0, 20, size=len(s), dtype=np.uint8) + 40).tobytes()
                    fh.write(b"@" + nm + b"\n" + s + b"\n+\n" + q + b"\n")
                else:
                    fh.write(b">" + nm + b"\n" + s + b"\n")
    print(f"[cli_bench] wrote {rp} in {time.time() - t0:.1f}s", file=sys.stderr)
bases = sum(len(s) for s in seqs) * a.copies
a.reads = a.reads * a.copies
exe = os.path.join(ROOT, "lordfast_amd", "lordfast")
for out in (("/dev/null",) if a.devnull_only else ("/dev/null", os.path.join(d, "cli_out.sam"))):
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([exe, "--search", fa, "--seq", rp, "-o", out, "-t", str(a.threads)], capture_output=True, text=True)
        if os.environ.get("LF_TIMING"):
            print("\n".join(l for l in r.stderr.splitlines() if "file batch" in l or "setup" in l), file=sys.stderr)
        el = time.time() - t0
        note = [l for l in r.stderr.splitlines() if "processed" in l or "search wall" in l]
        sw = [l for l in r.stderr.splitlines() if "search wall time" in l]
        search_rate = float(sw[-1].split(":")[-1].split()[0]) if sw else None
        # the CLI's own timer covers mapping only when the index is resident; we want the steady-state rate: subtract the index load
        load = 0.0
        for l in r.stderr.splitlines():
            if "index was loaded in" in l:
                load = float(l.split("loaded in")[1].split()[0])
        size = os.path.getsize(out) if out != "/dev/null" else None
        print(json.dumps(dict(input=os.path.basename(rp), output=out, rc=r.returncode, wall_s=round(el, 2), index_load_s=load,
                              reads_per_s_search_loop=search_rate, gbases_per_s=round(bases * (search_rate or 0) / a.reads / 1e9, 3),
                              sam_bytes=size, notes=note[-2:])), flush=True)
