"""Many equal Hirschberg roots in one call (stage API): the throughput side of the level kernels.  LF_HIRSCH_DEBUG=1 prints the levels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lordfast_amd as la
from lordfast_amd import synth
rng = np.random.default_rng(5)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
def rseq(n): return ACGT[rng.integers(0, 4, n)].tobytes()
la.edlib_batch([rseq(3000)], [rseq(3000)], [0])
sets = []
for name, cnt, n, mode, rel in (("nw3000junk", 2000, 3000, 0, 0), ("nw3000rel", 2000, 3000, 0, 1), ("nw6000junk", 600, 6000, 0, 0), ("shw6000junk", 600, 6000, 1, 0), ("nw6000rel", 600, 6000, 0, 1), ("shw16000junk", 60, 16000, 1, 0)):
    qs = [rseq(n) for _ in range(cnt)]
    if rel: ts = [synth.mutate(np.frombuffer(q, dtype=np.uint8), 0.15, rng).tobytes() for q in qs]
    else: ts = [rseq(n + (n // 5 if mode else 0)) for _ in qs]
    sets.append((name, qs, ts, [mode] * cnt))
os.environ["LF_HIRSCH_TRIAL"] = "0,0"
for band in (1, 0, 1, 0):
    os.environ["LF_HIRSCH_BAND"] = str(band)
    for name, qs, ts, modes in sets:
        sys.stderr.write("== band %d %s\n" % (band, name)); sys.stderr.flush()
        t0 = time.time()
        res, ms = la.edlib_batch(qs, ts, modes)
        print("band %d %-14s %6.1f ms kernels" % (band, name, ms), flush=True)
