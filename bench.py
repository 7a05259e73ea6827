#!/usr/bin/env python3
"""bench.py -- aligned reads/s of the GPU seed->chain->extend path on synthetic PacBio-error reads.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" maps one batch of synthetic reads (inputs in host memory -> SAM records in host memory, index
resident in HBM, index load excluded -- the reference's own timer, src/baseFAST.cpp:69-75).  Weak scaling:
every GPU gets the same number of reads; rank 0 scatters the packed batch and gathers the SAM records over
RCCL.  Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the dominant
kernel and `cpu_baseline` (the real reference, compiled into oracle/_ref, on all host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("LF_BENCH_GENOME_MBP", "3100")))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("LF_BENCH_READS", "100000")), help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--err", type=float, default=0.15)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workdir", default=os.environ.get("LF_BENCH_DIR", "/tmp/lf_bench"))
    ap.add_argument("--check", action="store_true", help="also compare a sample of the SAM with the oracle")
    return ap.parse_args()


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def ensure_index(args, rank):
    """synthetic genome + index files under workdir (rank 0 builds, others wait on the done marker)"""
    from lordfast_amd import synth
    tag = f"g{args.genome_mbp:g}"
    d = os.path.join(args.workdir, tag)
    fa = os.path.join(d, "genome.fa")
    done = os.path.join(d, "DONE")
    if rank == 0 and not os.path.exists(done):
        os.makedirs(d, exist_ok=True)
        t0 = time.time()
        total = int(args.genome_mbp * 1e6)
        n_contigs = max(2, min(24, total // 2_000_000))
        fams = max(0, min(1000, total // 100_000))
        contigs = synth.make_genome(total, n_contigs, seed=11, repeat_frac=0.10, n_families=fams)
        log(f"genome {total} bp, {n_contigs} contigs, {fams} repeat families: {time.time() - t0:.1f}s")
        t0 = time.time()
        import lordfast_amd as la
        if hasattr(la.lib(), "lf_index_build"):
            la.index_build(contigs, fa)                       # GPU indexer (writes the reference's formats)
        else:
            synth.write_fasta(fa, contigs)
            from oracle import pyoracle as po                 # checker-side indexer (the reference's own)
            po.Ref().index_build(fa)
        log(f"index built in {time.time() - t0:.1f}s")
        np.save(os.path.join(d, "contig_lens.npy"), np.array([len(s) for _, s in contigs]))
        open(done, "w").write("ok")
        return fa, contigs
    while not os.path.exists(done):
        time.sleep(1.0)
    return fa, None


def make_reads(args, contigs, fa, n_total):
    """seeded reads; cached on disk so that repeated bench runs on one box skip generation"""
    from lordfast_amd import synth
    key = hashlib.md5(f"{fa}|{n_total}|{args.read_len}|{args.err}".encode()).hexdigest()[:12]
    path = os.path.join(os.path.dirname(fa), f"reads_{key}.npz")
    if os.path.exists(path):
        z = np.load(path)
        blob, off, nblob, noff = z["blob"].tobytes(), z["off"], z["nblob"].tobytes(), z["noff"]
        seqs = [blob[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
        names = [nblob[int(noff[i]):int(noff[i + 1])] for i in range(len(noff) - 1)]
        return names, seqs
    if contigs is None:
        from tests.conftest import read_fasta
        cn, cs = read_fasta(fa)
        contigs = [(a.decode(), np.frombuffer(b, dtype=np.uint8)) for a, b in zip(cn, cs)]
    t0 = time.time()
    reads = synth.make_reads(contigs, n_total, args.read_len, args.err, seed=2024)
    names = [r[0].encode() for r in reads]
    seqs = [r[1] for r in reads]
    log(f"{n_total} reads generated in {time.time() - t0:.1f}s")
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    noff = np.concatenate([[0], np.cumsum([len(s) for s in names])]).astype(np.int64)
    np.savez(path, blob=np.frombuffer(b"".join(seqs), dtype=np.uint8), off=off,
             nblob=np.frombuffer(b"".join(names), dtype=np.uint8), noff=noff)
    return names, seqs


def cpu_baseline(args, fa, names, seqs):
    """the REAL reference (oracle/_ref/liblfref.so) on all host cores, on a bounded sample"""
    from oracle import pyoracle as po
    if not os.path.exists(po.REF_SO):
        return None
    if not os.path.exists(fa + ".cache"):
        return None
    ref = po.Ref()
    ref.load(fa)
    # "--threads = all cores": the reference takes every online CPU (capped at 255); inside a CPU-quota'd
    # container that oversubscribes, so also try the quota-sized pool and keep the FASTER of the two
    cands = [0]
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cands.append(max(1, -(-int(q) // int(per))))
    except Exception:                                                    # noqa: BLE001
        pass
    best = None
    for th in cands:
        ref.set_params(po.default_params(threads=th), "bench")
        c = ref.threads()
        pilot = min(len(seqs), max(4 * c, 256))
        _, secs = ref.map_mem(names[:pilot], seqs[:pilot])
        r = pilot / max(secs, 1e-6)
        log(f"cpu baseline pilot: --threads {c}: {r:.1f} reads/s")
        if best is None or r > best[0]:
            best = (r, th, c)
    rate, th, cores = best
    ref.set_params(po.default_params(threads=th), "bench")
    n = int(min(len(seqs), max(pilot, rate * args.cpu_seconds)))
    sam, secs = ref.map_mem(names[:n], seqs[:n])
    bases = sum(len(s) for s in seqs[:n])
    return dict(value=n / secs, unit="reads/s", cores=cores, kind="reference",
                sample=f"first {n} reads of the same batch ({bases / 1e6:.1f} Mbp), mapSeqMT only, {secs:.1f}s, "
                       f"--threads {cores}", bp_per_s=bases / secs), sam, n


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
    dev = torch.device("cuda", local)

    import lordfast_amd as la
    from lordfast_amd import dist as lfd
    if la.device_count() < 1:
        sys.exit("bench.py: no gfx950 device visible; the HIP path has no CPU fallback")

    fa, contigs = ensure_index(args, rank)
    if dist:
        dist.barrier()
    t0 = time.time()
    lf = la.LordFast(fa, device=local, full_sa=True)
    log(f"rank {rank}: index resident in HBM after {time.time() - t0:.1f}s")
    params = la.default_params(min_anchor_len=14, sampling_count=1000)       # -k 14 -c 1000 --chainAlg dp-n2

    n_total = args.reads * world
    if rank == 0:
        names, seqs = make_reads(args, contigs, fa, n_total)
    else:
        names, seqs = None, None

    def step():
        if dist:
            my_names, my_seqs, _ = lfd.scatter_reads(dist, torch, names, seqs, dev)
        else:
            my_names, my_seqs = names, seqs
        t_call = time.perf_counter()
        sam, st = lf.map_batch(my_names, my_seqs, params=params, copy=False)
        st["ms_python_call"] = (time.perf_counter() - t_call) * 1e3
        if dist:
            sam = lfd.gather_sam(dist, torch, sam.view(), dev)
        return sam, st

    for _ in range(args.warmup):
        step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agg = None
    sam = None
    for _ in range(args.steps):
        sam, st = step()
        if agg is None:
            agg = dict(st)
        else:
            for k, v in st.items():
                agg[k] += v
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        K = args.steps
        bases = sum(len(s) for s in seqs)
        value = n_total * K / elapsed
        # ---- roofline of the dominant kernel (HIP-event time summed over the timed steps, rank 0) ----
        kernels = {
            "lf_seed_search_kernel": (agg["ms_k_search"], 16 * agg["n_cache"] + 64 * agg["n_occblk"] + agg["n_readbytes"], agg["search_launches"]),
            "lf_seed_locate_kernel": (agg["ms_k_locate"], 8 * agg["n_sa"] + 9 * agg["n_sa"], agg["locate_launches"]),
            "lf_edlib_kernel": (agg["ms_k_edlib"], agg["ext_bytes"], max(1, agg["edlib_launches"])),
            "lf_chain_n2_kernel": (agg["ms_k_chain"], 16 * agg["n_chain_problems"], max(1, K)),
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, alg_bytes, launches = kernels[dom]
        achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=8000.0, unit="GB/s", frac=achieved / 8000.0,
                        traffic=None, launches=int(launches), avg_launch_ms=ms / max(1, launches),
                        algorithmic_bytes_per_launch=alg_bytes / max(1, launches),
                        per_kernel_ms={k: round(v[0], 3) for k, v in kernels.items()})
        out = {
            "metric": "aligned reads/s", "value": value, "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.reads} synthetic PacBio reads per GPU (~{args.read_len} bp, {args.err:.0%} err) vs "
                                   f"{args.genome_mbp:g} Mbp synthetic genome, -k 14 -c 1000 --chainAlg dp-n2",
                       "reads_per_gpu": args.reads, "mean_read_len": bases / max(1, n_total), "genome_mbp": args.genome_mbp,
                       "parallelism": f"read-sharded x{world}", "index": "FM-index + full SA resident in HBM"},
            "gbp_per_s": bases * K / elapsed / 1e9,
            "host_ms_per_step": {k: agg[k] / K for k in ("ms_total", "ms_python_call", "ms_seed", "ms_vote", "ms_chain", "ms_extend", "ms_sam")},
            "per_read": {"seeds": agg["n_seeds"] / (n_total * K / world), "edlib_problems": agg["n_edlib_problems"] / (n_total * K / world),
                         "seed_bytes": (16 * agg["n_cache"] + 64 * agg["n_occblk"] + 8 * agg["n_sa"] + agg["n_readbytes"]) / (n_total * K / world)},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                cb = cpu_baseline(args, fa, names, seqs)
            except Exception as e:                                          # noqa: BLE001
                log("cpu baseline failed:", e)
                cb = None
            if cb:
                base, ref_sam, n = cb
                out["cpu_baseline"] = base
                out["speedup_vs_cpu_baseline"] = value / base["value"]
                # CIGAR bit-match rate against the reference on the sampled reads (primary records)
                # the reference sample is a prefix of the batch: compare the head of our SAM (records are in read order)
                head = sam.head(4 * len(ref_sam) + (1 << 20)) if hasattr(sam, "head") else sam[:4 * len(ref_sam) + (1 << 20)]
                mine = [l for l in head.split(b"\n")[:-1] if l]
                want = {}
                for l in ref_sam.split(b"\n"):
                    f = l.split(b"\t")
                    if len(f) > 5 and not (int(f[1]) & (256 | 2048)):
                        want[f[0]] = l
                hit = tot = 0
                for l in mine:
                    f = l.split(b"\t")
                    if f[0] in want and not (int(f[1]) & (256 | 2048)):
                        tot += 1
                        hit += (l == want[f[0]])
                out["primary_record_match_rate"] = hit / max(1, tot)
                out["primary_records_compared"] = tot
        print(json.dumps(out), flush=True)
    lf.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
