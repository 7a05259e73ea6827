#!/usr/bin/env python3
"""chunk_matrix.py -- one process, index loaded once: step time of the host-boundary and the HBM-resident call for several batch
sizes x chunk sizes (LF_CHUNK_READS) x SAM egress modes.  One JSON line per cell."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch
import bench
sys.argv = ["bench.py"]
args = bench.parse()
fa, contigs = bench.ensure_index(args, 0)
import lordfast_amd as la
from lordfast_amd import dist as lfd
lf = la.LordFast(fa, device=0, full_sa=True)
print(json.dumps({"index": lf.describe()}), flush=True)
names, seqs = bench.make_reads(args, contigs, fa, 0)
params = la.default_params(min_anchor_len=14, sampling_count=1000, threads=bench.host_budget())
dev = torch.device("cuda", 0)
bases = sum(len(s) for s in seqs)
cap = int(3.3 * bases) + len(seqs) * 2048
host_out = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
dev_out = torch.empty(cap, dtype=torch.uint8, device=dev)

def cell(tag, n, env, host, steps=5):
    for k in ("LF_SAM_FULL", "LF_LANES", "LF_CHUNK_READS"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    nm, sq = names[:n], seqs[:n]
    if host:
        arrs = (la.api._cstr_array(nm), la.api._cstr_array(sq))
        lens = np.array([len(x) for x in sq], dtype=np.uint32)
        fn = lambda: lf.map_batch_into(nm, None, host_out.data_ptr(), host_out.numel(), params=params, name_arr=arrs[0], seq_arr=arrs[1], seq_lens=lens)
    else:
        shard = lfd.make_shards(torch, nm, sq, 1, dev)[0][0]
        na = shard.name_array(torch)
        fn = lambda: lf.map_batch_dev(na, shard.blob.data_ptr(), shard.seq_off[:-1], shard.seq_lens, dev_out.data_ptr(), dev_out.numel(), True, params=params)
    fn(); fn()
    torch.cuda.synchronize()
    c0 = sum(os.times()[:4]); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    dt = (time.perf_counter() - t0) / steps; cpu = (sum(os.times()[:4]) - c0) / steps
    print(json.dumps({"cell": tag, "reads": n, "io": "host" if host else "hbm", **env, "ms_per_step": round(dt * 1e3, 2), "reads_per_s": round(n / dt), "host_cpu_s_per_step": round(cpu, 3)}), flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "host"):
    for cr in (4167, 6250, 8334, 12500, 25000):
        cell("host boundary, holes", 100000, {"LF_CHUNK_READS": cr}, True)
    for cr in (4167, 12500):
        cell("host boundary, whole lines", 100000, {"LF_CHUNK_READS": cr, "LF_SAM_FULL": 1}, True)
if which in ("all", "hbm"):
    for n, crs in ((12500, (1563, 3125, 6250, 12500)), (25000, (3125, 6250, 12500, 25000)), (50000, (6250, 12500, 25000)), (100000, (12500, 25000, 50000))):
        for cr in crs:
            cell("hbm resident", n, {"LF_CHUNK_READS": cr}, False, steps=8 if n < 50000 else 5)
