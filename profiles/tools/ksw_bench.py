#!/usr/bin/env python3
"""ksw_bench.py -- lf_ksw_extend2_batch on clip-test-like problems (query = a read's tail, target = the reference behind the
chain, ~15 % errors, the reference's two parameter sets): wall time of a batch of N problems of L rows with the default kernel
(lf_ksw_mw_kernel) and with LF_KSW_1WAVE=1 (round 3's lf_ksw_kernel).  N = 1: the latency of a row; N large: throughput."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lordfast_amd as la
from lordfast_amd import synth

def problems(n, L, seed):
    rng = np.random.default_rng(seed)
    code = np.zeros(256, dtype=np.uint8); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
    qs, ts = [], []
    for i in range(n):
        qa = rng.integers(0, 4, size=L).astype(np.uint8)
        qb = np.frombuffer(b"ACGT", dtype=np.uint8)[qa]
        ta = synth.mutate(qb, 0.15, rng)
        qs.append(qa); ts.append(code[ta])
    return qs, ts

def main():
    out = []
    for L in (2000, 10000):
        for n in (1, 64, 512):
            qs, ts = problems(n, L, 7 + n)
            for name, prm in (("w40", (0, 1, 0, 1, 40, 40, L)), ("w100", (8, 1, 4, 1, 100, 200, L))):
                prms = [prm] * n
                row = {"rows": L, "problems": n, "prm": name}
                for mode in ("default", "mw", "onewave"):      # default: lf_ksw_r4_kernel (round 5); mw: LF_KSW_R4=0 (round 4's four-wavefront kernel); onewave: round 3's
                    os.environ.pop("LF_KSW_1WAVE", None); os.environ.pop("LF_KSW_R4", None)
                    if mode == "onewave": os.environ["LF_KSW_1WAVE"] = "1"
                    if mode == "mw": os.environ["LF_KSW_R4"] = "0"
                    la.ksw_extend2_batch(qs, ts, prms)
                    t0 = time.perf_counter(); res = la.ksw_extend2_batch(qs, ts, prms); dt = time.perf_counter() - t0
                    rows_done = sum(r[2] for r in res)
                    row[mode + "_ms"] = round(dt * 1e3, 3); row[mode + "_us_per_row_per_problem"] = round(dt * 1e6 / max(1, rows_done / n), 3)
                    row["rows_done_mean"] = rows_done / n
                print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()
