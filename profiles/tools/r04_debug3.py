#!/usr/bin/env python3
"""full-size host-boundary steps under different settings in ONE process (index built / loaded once); a hang aborts through LF_WATCHDOG"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["LF_WATCHDOG"] = "60"
import numpy as np
import torch
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
fa, contigs = bench.ensure_index(args, 0)
import lordfast_amd as la
lf = la.LordFast(fa, device=0, full_sa=True)
names, seqs = bench.make_reads(args, contigs, fa, 0)
params = la.default_params(min_anchor_len=14, sampling_count=1000, threads=16)
bases = sum(len(s) for s in seqs)
cap = int(3.3 * bases) + len(seqs) * 2048
host_out = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
arrs = (la.api._cstr_array(names), la.api._cstr_array(seqs))
lens = np.array([len(x) for x in seqs], dtype=np.uint32)
import xxhash
def run(tag, env):
    for k in ("LF_SAM_FULL", "LF_LANES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    print("==", tag, env, flush=True)
    for it in range(3):
        t0 = time.perf_counter()
        ln, st = lf.map_batch_into(names, None, host_out.data_ptr(), host_out.numel(), params=params, name_arr=arrs[0], seq_arr=arrs[1], seq_lens=lens)
        dt = time.perf_counter() - t0
        print(f"   {tag} step {it}: {dt * 1e3:.1f} ms, {ln} bytes", flush=True)
    h = xxhash.xxh3_128(); h.update(host_out[:ln].numpy().tobytes()); print("   digest", h.hexdigest(), flush=True)
run("full-text egress, 8 lanes", {"LF_SAM_FULL": "1"})
run("holes, 1 lane", {"LF_LANES": "1"})
run("holes, 2 lanes", {"LF_LANES": "2"})
run("holes, 8 lanes", {})
