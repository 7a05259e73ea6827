#!/usr/bin/env python3
"""small segdup workload: our records vs the compiled reference's, read by read; prints the differing reads"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
sys.argv = ["bench.py", "--config", "c4", "--genome-mbp", "120", "--reads", "8000"] + sys.argv[1:]
args = bench.parse()
fa, contigs = bench.ensure_index(args, 0)
import lordfast_amd as la
from oracle import pyoracle as po
lf = la.LordFast(fa, device=0, full_sa=True)
names, seqs = bench.make_reads(args, contigs, fa, 0)
ref = po.Ref(); ref.load(fa)
for alg, nm in ((1, 30), (0, 30), (1, 10)):
    p = la.default_params(chain_alg=alg, max_map=nm)
    ref.set_params(po.default_params(threads=0, chain_alg=alg, max_map=nm), "dbg")
    want, _ = ref.map_mem(names, seqs)
    got, st = lf.map_batch(names, seqs, params=p)
    def by_read(txt):
        d = {}
        for l in txt.split(b"\n"):
            if l:
                d.setdefault(l.split(b"\t", 1)[0], []).append(l)
        return d
    W, G = by_read(want), by_read(got)
    bad = [n for n in W if W[n] != G.get(n)]
    print(f"chain_alg {alg} -n {nm}: {len(bad)} of {len(W)} reads differ", flush=True)
    for n in bad[:4]:
        w, g = W[n], G.get(n, [])
        print("  read", n.decode(), "reference lines", len(w), "ours", len(g))
        for i in range(max(len(w), len(g))):
            a = w[i].split(b"\t") if i < len(w) else None
            b = g[i].split(b"\t") if i < len(g) else None
            if a != b:
                def short(x):
                    return None if x is None else [f[:60] for k, f in enumerate(x) if k not in (9, 10)]
                print("   line", i, "\n     ref ", short(a), "\n     ours", short(b))
                if a and b:
                    print("     differing fields:", [k for k in range(min(len(a), len(b))) if a[k] != b[k]])
                break

# ---- is the differing read the reference's stale-chain case (src/Chain.cpp:68,92: chain_seeds_clasp with zero fragments leaves the
# previous call's chain in place)?  then the reference's own output depends on what its thread mapped before
bad_name = b"r2048_chr13_2794749_-"
if bad_name in names:
    i = names.index(bad_name)
    orc = po.Oracle(fa)
    p1 = po.default_params(chain_alg=1, max_map=30)
    print("oracle restatement on the read alone:", [l.split(b"\t")[:6] for l in orc.map_batch([names[i]], [seqs[i]], params=p1).split(b"\n") if l])
    ours, _ = lf.map_batch([names[i]], [seqs[i]], params=la.default_params(chain_alg=1, max_map=30))
    print("ours on the read alone:", [l.split(b"\t")[:6] for l in ours.split(b"\n") if l])
    for th, lo in ((1, i), (1, max(0, i - 40)), (16, 0)):
        ref.set_params(po.default_params(threads=th, chain_alg=1, max_map=30), "dbg")
        w, _ = ref.map_mem(names[lo:i + 1], seqs[lo:i + 1])
        ls = [l.split(b"\t") for l in w.split(b"\n") if l.startswith(bad_name)]
        print(f"reference, --threads {th}, reads {lo}..{i}:", [[f[:40] for f in l[:6]] for l in ls])
