"""One NW problem per call, alone on the GPU: the banded level kernel with a band that covers (almost) the whole matrix (LF_HIRSCH_TRIAL=16,16: the root's
trial bound is its row count) against the unbanded kernel -- same cells, same chain; what differs is the kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lordfast_amd as la
rng = np.random.default_rng(5)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
def rseq(n): return ACGT[rng.integers(0, 4, n)].tobytes()
la.edlib_batch([rseq(3000)], [rseq(3000)], [0])
for n in (5000, 8000, 12000):
    q, t = rseq(n), rseq(n)
    for env in ({"LF_HIRSCH_BAND": "0"}, {"LF_HIRSCH_BAND": "1", "LF_HIRSCH_TRIAL": "16,16"}, {"LF_HIRSCH_BAND": "1", "LF_HIRSCH_TRIAL": "0,0"}):
        for k in ("LF_HIRSCH_BAND", "LF_HIRSCH_TRIAL"): os.environ.pop(k, None)
        os.environ.update(env)
        for rep in range(2):
            sys.stderr.write("== n %d %s\n" % (n, env)); sys.stderr.flush()
            res, ms = la.edlib_batch([q], [t], [0])
        print(n, env, res[0][0], "%.2f ms" % ms, flush=True)
