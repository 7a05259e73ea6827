"""Latency of ONE large edlib problem through the Hirschberg levels, alone on the GPU (stage API, lf_edlib_batch): banded against unbanded
sweeps, related (15 %) and unrelated strings.  LF_HIRSCH_DEBUG=1 prints the levels of each call on stderr."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lordfast_amd as la
from lordfast_amd import synth

rng = np.random.default_rng(5)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rseq(n):
    return ACGT[rng.integers(0, 4, n)].tobytes()


cases = []
for n in (6000, 12000, 24000, 48000):
    q = rseq(n)
    cases.append(("related15", n, 0, q, synth.mutate(np.frombuffer(q, dtype=np.uint8), 0.15, rng).tobytes()))
    cases.append(("unrelated", n, 0, q, rseq(n)))
    cases.append(("related15", n, 1, q, synth.mutate(np.frombuffer(q, dtype=np.uint8), 0.15, rng).tobytes() + rseq(n // 5)))
    cases.append(("unrelated", n, 1, q, rseq(n + n // 5)))
la.edlib_batch([rseq(3000)], [rseq(3000)], [0])      # warm up: code objects, slots
for band in (1, 0):
    os.environ["LF_HIRSCH_BAND"] = str(band)
    for name, n, mode, q, t in cases:
        best = 1e9
        for rep in range(2):
            sys.stderr.write("== band %d %s n %d mode %d\n" % (band, name, n, mode)); sys.stderr.flush()
            t0 = time.time()
            res, ms = la.edlib_batch([q], [t], [mode])
            best = min(best, (time.time() - t0) * 1e3)
        print("band %d  %-10s n %6d m %6d mode %d  ed %6d  %8.2f ms wall (kernels %.2f ms)" % (band, name, n, len(t), mode, res[0][0], best, ms), flush=True)
