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
ap.add_argument("--devnull-only", action="store_true", help="no SAM file on disk (large read sets)")
a = ap.parse_args()
sys.argv = [sys.argv[0], "--genome-mbp", str(a.genome_mbp), "--reads", str(a.reads)]
args = bench.parse()
fa, contigs = bench.ensure_index(args, 0)
names, seqs = bench.make_reads(args, contigs, fa, 0)
d = os.path.dirname(fa)
ext = ("fq" if a.fastq else "fa") + (".gz" if a.gz else "")
rp = os.path.join(d, f"reads_{a.reads}.{ext}")
if not os.path.exists(rp):
    t0 = time.time()
    op = gzip.open if a.gz else open
    with op(rp, "wb", **({"compresslevel": 1} if a.gz else {})) as fh:
        for n, s in zip(names, seqs):
            if a.fastq:
                fh.write(b"@" + n + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
            else:
                fh.write(b">" + n + b"\n" + s + b"\n")
    print(f"[cli_bench] wrote {rp} in {time.time() - t0:.1f}s", file=sys.stderr)
bases = sum(len(s) for s in seqs)
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
