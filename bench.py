#!/usr/bin/env python3
"""bench.py -- aligned reads/s of the GPU seed->chain->extend path on synthetic PacBio-error reads.

    python bench.py --gpus N --steps K --warmup W
        N > 1 from a bare shell: this process starts the N ranks itself (python -m torch.distributed.run ... bench.py) and relays
        rank 0's line; already under torchrun (RANK / WORLD_SIZE set, the driver's form) it is one of the ranks.

A "step" maps one batch of synthetic reads with the index resident in HBM (index load excluded -- the reference's own timer,
src/baseFAST.cpp:69-75).
N = 1: `value` / `ms_per_step` = the boundary SURVEY 8(d) defines and the reference has: reads in host memory -> SAM records in
host memory (lf_map_batch_into_lens; H2D and D2H are inside the step).  `value_hbm_resident`: the same steps with the read bases
already in HBM and the SAM text left there (lf_map_batch_dev) -- what one rank of the N-GPU deployment does.
N > 1: one process per GPU, index replicated, reads are independent.  --scaling strong (default for N > 1): the SAME --reads set
cut by bases over the ranks (BASELINE config C3); weak: --reads per GPU.  `value` = the rate with the exchange (--exchange on,
default): rank 0 owns every step's reads and ends with its SAM records; both move over RCCL point-to-point (xGMI), pipelined one
step ahead / behind the mapping (lordfast_amd/dist.py: PipelinedExchange), inside the timed region.  The rates without any
exchange (HBM-resident shards; every rank's own host buffers) and the weak-scaling rate are reported next to it.
The records the TIMED steps wrote are snapshotted (digest + head) before anything else runs; that snapshot is what is compared
with the reference, and it must equal the output of the exclusive (one lane, one chunk) pass.
Prints ONE JSON line on rank 0 carrying `roofline` for the single kernel with the largest exclusive time (all kernels in
`by_kernel`) and `cpu_baseline` (the real reference, compiled into oracle/_ref, on the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # before the HIP runtime initialises (see lf_host.c)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 4 (c4: 10 -- 10 x 100 k = the 1 M reads of BASELINE config C4)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("LF_BENCH_GENOME_MBP", "3100")))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("LF_BENCH_READS", "100000")), help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--err", type=float, default=0.15)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the exclusive pass (profiler counter runs: exactly the timed steps' launches)")
    ap.add_argument("--no-host-region", action="store_true", help="skip the host-buffer region (kernel-trace runs that measure the GPU-busy fraction of the HBM-resident steps)")
    ap.add_argument("--workdir", default=os.environ.get("LF_BENCH_DIR", "/tmp/lf_bench"))
    ap.add_argument("--chain-alg", choices=["dp-n2", "clasp"], default="dp-n2",
                    help="BASELINE config C2 (the headline) is dp-n2; clasp + --max-map 30 is config C4's option set")
    ap.add_argument("--max-map", type=int, default=10, help="-n (config C4: 30)")
    ap.add_argument("--single-output", action="store_true", help="(kept for round-1 command lines) same as --exchange on")
    ap.add_argument("--exchange", choices=["auto", "on", "off"], default="auto",
                    help="N>1: rank 0 owns every step's read batch (in HBM); its shards go to the ranks and their SAM records come back "
                         "over RCCL point-to-point, pipelined with the mapping, inside the timed region.  auto = on for N>1.  The rate "
                         "without the exchange is reported too")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong (default for N > 1): the same --reads set cut by bases over the N ranks (config C3); weak: --reads per GPU "
                         "(work grows with N; reported as value_weak_* next to the strong line)")
    ap.add_argument("--dup-frac", type=float, default=None,
                    help="fraction of the reads drawn from segmental duplications (2-4 copies of 40-80 kbp segments at 1-3 %% divergence, "
                         "--segdups of them pasted into the genome): those reads have several near-equal candidate windows, so mapSeq "
                         "takes its fine branch (src/LordFAST.cpp:542-562) and -n windows are chained and extended.  c4 default 0.3, else 0")
    ap.add_argument("--segdups", type=int, default=2000, help="duplicated segments in the genome when --dup-frac > 0")
    ap.add_argument("--inflight", type=int, default=None,
                    help="steps in flight at once (threads calling the library concurrently; its lane allocator shares the device's 8 lanes "
                         "between them).  1 = one step after the other.  Small shards (a rank's share under strong scaling) are bound by the "
                         "latency of a chunk's launch chain, which the next steps' kernels hide; every step is still a complete call.  "
                         "Default: 2 for N > 1 under strong scaling, 4 when a rank's shard is at most 32 k reads (a rank's shard is 1 / N of the batch; the "
                         "exchange loop keeps that many of a rank's steps in flight too, lordfast_amd/dist.py: run_pipeline), else 1")
    ap.add_argument("--mode", choices=["ranks", "inproc"], default="ranks",
                    help="ranks: one process per GPU (torch.distributed); inproc: ONE process drives N devices through lf_map_batch_multi "
                         "(chunks pulled from one counter; every device copies through its own PCIe link)")
    ap.add_argument("--exchange-timeout", type=float, default=float(os.environ.get("LF_BENCH_EXCHANGE_TIMEOUT", "240")),
                    help="seconds after which a stuck exchange phase is abandoned: the line is printed with the no-exchange rate as value")
    ap.add_argument("--ref-fasta", default=os.environ.get("LF_BENCH_REF_FASTA"),
                    help="a real reference FASTA (plain or .gz) to index and draw the synthetic-error reads from (SURVEY 8d: 'real FASTA if present on the "
                         "box, else synthetic').  Default: $LF_BENCH_REF_FASTA, else the first existing one of a few conventional paths, else the synthetic genome")
    ap.add_argument("--repeat-profile", choices=["default", "grch38like", "t2tlike"], default=None,
                    help="default: 10 %% of the genome from 1000 low-copy families (SURVEY 8d); grch38like: ~50 %% repeats incl. a "
                         "300 bp family with ~10^6 copies per 3 Gbp and truncated 1-6 kbp families; t2tlike (default of --config c5, 'vs CHM13 T2T'): "
                         "grch38like + a centromeric satellite array (171 bp monomers in higher-order repeats) per contig and simple-sequence arrays")
    ap.add_argument("--config", choices=["c2", "c4", "c5"], default="c2",
                    help="BASELINE.json config: c2 = 15 kbp / 15 %% PacBio, -k 14 -c 1000 dp-n2 (headline); c4 = clasp -n 30; "
                         "c5 = 50 kbp / 10 %% ONT error mix, -k 17 -c 2000")
    a = ap.parse_args()
    if a.single_output:
        a.exchange = "on"
    if a.config == "c4":
        a.chain_alg, a.max_map = "clasp", 30
        if a.dup_frac is None:
            a.dup_frac = 0.3
    if a.dup_frac is None:
        a.dup_frac = 0.0
    if a.steps is None:
        a.steps = 10 if a.config == "c4" else 4
    if a.scaling is None:
        a.scaling = "strong" if a.gpus > 1 else "weak"
    if a.inflight is None:
        # a rank's shard under strong scaling: 2 steps in flight, 4 when the shard is small (12.5 k reads at N = 8: 9.2 -> 8.3 ms per step, 25 k at
        # N = 4: 18.0 -> 17.4 ms on the one-GPU proxy, profiles/r05_shard_sweep/); one step after the other everywhere else
        strong_ranks = a.gpus > 1 and a.scaling == "strong" and a.mode == "ranks"
        a.inflight = (4 if a.reads // a.gpus <= 32768 else 2) if strong_ranks else 1
    if a.repeat_profile is None:
        a.repeat_profile = "t2tlike" if a.config == "c5" else "default"
    if not a.ref_fasta:
        for cand in ("/data/reference/GRCh38.fa", "/data/GRCh38.fa", "/data/chm13v2.0.fa", "/opt/data/GRCh38.fa", os.path.expanduser("~/GRCh38.fa")):
            if os.path.exists(cand) or os.path.exists(cand + ".gz"):
                a.ref_fasta = cand if os.path.exists(cand) else cand + ".gz"
                break
    if a.ref_fasta and not os.path.exists(a.ref_fasta):
        sys.exit(f"bench.py: --ref-fasta {a.ref_fasta}: no such file")
    if a.config == "c5":
        if a.read_len == 15000:
            a.read_len = 50000
        if a.err == 0.15:
            a.err = 0.10
        if "LF_BENCH_READS" not in os.environ and a.reads == 100000:
            a.reads = 30000
    return a


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def genome_recipe(args):
    total = int(args.genome_mbp * 1e6)
    return total, max(2, min(24, total // 2_000_000)), max(0, min(1000, total // 100_000))


def make_contigs(args):
    """the synthetic genome is a pure function of its recipe: any rank can regenerate it instead of parsing the FASTA.
    -> (contigs, segdup families or None)"""
    from lordfast_amd import synth
    if args.ref_fasta:
        contigs = synth.read_fasta(args.ref_fasta)
        if not contigs:
            sys.exit(f"bench.py: no contig of >= 2000 bases in {args.ref_fasta}")
        total = sum(len(c[1]) for c in contigs)
    else:
        total, n_contigs, fams = genome_recipe(args)
        contigs = synth.make_genome(total, n_contigs, seed=11, repeat_frac=0.10, n_families=fams, profile=args.repeat_profile)
    segd = None
    if args.dup_frac > 0:
        nseg = max(1, min(args.segdups, total // 400_000))
        seg_len = (40000, 80000) if total >= 50_000_000 else (12000, 20000)
        segd = synth.add_segdups(contigs, nseg, seg_len=seg_len, seed=7)
    return contigs, segd


def ensure_index(args, rank):
    """synthetic genome + index files under workdir (rank 0 builds, others wait on the done marker)"""
    from lordfast_amd import synth
    tag = f"g{args.genome_mbp:g}" + ("" if args.repeat_profile == "default" else "_" + args.repeat_profile) + (f"_sd{args.segdups}" if args.dup_frac > 0 else "")
    if args.ref_fasta:
        st = os.stat(args.ref_fasta)
        tag = "ref_" + hashlib.md5(f"{os.path.abspath(args.ref_fasta)}|{st.st_size}".encode()).hexdigest()[:10] + (f"_sd{args.segdups}" if args.dup_frac > 0 else "")
    d = os.path.join(args.workdir, tag)
    fa = os.path.join(d, "genome.fa")
    done = os.path.join(d, "DONE")
    if rank == 0 and not os.path.exists(done):
        os.makedirs(d, exist_ok=True)
        t0 = time.time()
        contigs = make_contigs(args)
        log((f"reference {args.ref_fasta}: " if args.ref_fasta else f"synthetic genome ({args.repeat_profile} repeats): ") + f"{sum(len(c[1]) for c in contigs[0])} bp, {len(contigs[0])} contigs"
            + (f", {len(contigs[1])} segmental duplications" if contigs[1] else "") + f": {time.time() - t0:.1f}s")
        t0 = time.time()
        import lordfast_amd as la
        la.index_build(contigs[0], fa)                        # GPU indexer (writes the reference's file formats)
        log(f"index built in {time.time() - t0:.1f}s")
        open(done, "w").write("ok")
        return fa, contigs
    while not os.path.exists(done):
        time.sleep(1.0)
    return fa, None


def make_reads(args, contigs, fa, rank):
    """this rank's seeded reads (seed 2024 + rank: every GPU maps DIFFERENT reads); cached on disk so that repeated
    bench runs on one box skip generation"""
    from lordfast_amd import synth
    key = hashlib.md5((f"{fa}|{args.reads}|{args.read_len}|{args.err}|{rank}|{args.config == 'c5'}" + (f"|dup{args.dup_frac:g}" if args.dup_frac > 0 else "")).encode()).hexdigest()[:12]
    path = os.path.join(os.path.dirname(fa), f"reads_{key}.npz")
    if os.path.exists(path):
        z = np.load(path)
        blob, off, nblob, noff = z["blob"].tobytes(), z["off"], z["nblob"].tobytes(), z["noff"]
        seqs = [blob[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
        names = [nblob[int(noff[i]):int(noff[i + 1])] for i in range(len(noff) - 1)]
        return names, seqs
    if contigs is None:
        contigs = make_contigs(args)
    t0 = time.time()
    mix = (0.40, 0.25, 0.35) if args.config == "c5" else (0.15, 0.50, 0.35)          # ONT / PacBio CLR profile
    reads = synth.make_reads(contigs[0], args.reads, args.read_len, args.err, seed=2024 + rank, mix=mix, segdups=contigs[1], dup_frac=args.dup_frac,
                             acgt_only=bool(args.ref_fasta))
    names = [(r[0] if rank == 0 else f"g{rank}_{r[0]}").encode() for r in reads]
    seqs = [r[1] for r in reads]
    log(f"rank {rank}: {args.reads} reads generated in {time.time() - t0:.1f}s")
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    noff = np.concatenate([[0], np.cumsum([len(s) for s in names])]).astype(np.int64)
    tmp = path + f".tmp{os.getpid()}.npz"
    np.savez(tmp, blob=np.frombuffer(b"".join(seqs), dtype=np.uint8), off=off,
             nblob=np.frombuffer(b"".join(names), dtype=np.uint8), noff=noff)
    os.replace(tmp, path)
    return names, seqs


def sample_order(n_total, n_want):
    """read indices for the parity / CPU-baseline sample: a golden-ratio stride over the WHOLE batch, so that any prefix of the
    list is spread over every chunk and lane of the step (not the first reads only)"""
    n_want = max(1, min(n_total, n_want))
    g = max(1, int(n_total * 0.6180339887))
    while np.gcd(g, n_total) != 1:
        g += 1
    return [int((j * g) % n_total) for j in range(n_want)]


def records_of(view, n_bytes, wanted):
    """-> {read name: [its SAM lines]} for the names in `wanted`, out of n_bytes of SAM text (numpy uint8 view); records of a read
    are consecutive lines that start with its name"""
    out = {}
    wanted = set(wanted)
    CH = 1 << 28
    carry = b""
    for a in range(0, n_bytes, CH):
        blk = carry + bytes(memoryview(view[a:min(n_bytes, a + CH)]))
        cut = blk.rfind(b"\n") + 1
        carry = blk[cut:]
        for l in blk[:cut].split(b"\n"):
            if l:
                nm = l.split(b"\t", 1)[0]
                if nm in wanted:
                    out.setdefault(nm, []).append(l)
    return out


def cpu_baseline(args, fa, names, seqs):
    """the REAL reference (oracle/_ref/liblfref.so) on all host cores, on a bounded sample (the caller hands over the sample: reads
    spread over the whole batch)"""
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
        kk, cc = (17, 2000) if args.config == "c5" else (14, 1000)
        ref.set_params(po.default_params(threads=th, chain_alg=1 if args.chain_alg == 'clasp' else 0, max_map=args.max_map,
                                         min_anchor_len=kk, sampling_count=cc), "bench")
        c = ref.threads()
        pilot = min(len(seqs), max(4 * c, 256))
        _, secs = ref.map_mem(names[:pilot], seqs[:pilot])
        r = pilot / max(secs, 1e-6)
        log(f"cpu baseline pilot: --threads {c}: {r:.1f} reads/s")
        if best is None or r > best[0]:
            best = (r, th, c)
    rate, th, cores = best
    ref.set_params(po.default_params(threads=th, chain_alg=1 if args.chain_alg == 'clasp' else 0, max_map=args.max_map,
                                     min_anchor_len=kk, sampling_count=cc), "bench")
    n = int(min(len(seqs), max(pilot, rate * args.cpu_seconds)))
    sam, secs = ref.map_mem(names[:n], seqs[:n])
    bases = sum(len(s) for s in seqs[:n])
    granted = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            granted = min(granted, max(1, -(-int(q) // int(per))))
    except Exception:                                                    # noqa: BLE001
        pass
    # `cores` = the CPUs this container may use (cgroup quota), `threads` = the reference's --threads value that was fastest
    return dict(value=n / secs, unit="reads/s", cores=granted, cpus_granted=granted, cpus_online=os.cpu_count(), threads=cores,
                kind="reference",
                sample=f"{n} reads of the same batch, taken with a golden-ratio stride across ALL of it ({bases / 1e6:.1f} Mbp), mapSeqMT only, {secs:.1f}s, "
                       f"--threads {cores} on {granted} granted CPUs", bp_per_s=bases / secs), sam, n


def host_budget():
    """CPUs this container may use (cgroup quota or online CPUs)"""
    budget = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            budget = min(budget, max(1, -(-int(q) // int(per))))
    except Exception:                                                    # noqa: BLE001
        pass
    return budget


def spawn_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as FRESH child processes (this process has touched
    neither HIP nor torch) and exit with their status; rank 0's JSON line goes to our stdout unchanged."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting", args.gpus, "ranks:", " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if os.environ.get("LF_BENCH_STACKS"):      # debugging aid: the Python stacks of all threads every N seconds (where a hung run is)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LF_BENCH_STACKS"]), repeat=True, file=sys.stderr)
    under_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and args.mode == "ranks" and not under_torchrun:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0")) if args.mode == "ranks" else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if args.mode == "ranks" else 1
    local = int(os.environ.get("LOCAL_RANK", "0")) if args.mode == "ranks" else 0
    if args.mode == "ranks" and world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import threading

    # how long a lane driver polls for its stream before it sleeps (lf_gpu_common.h: lf_stream_wait): 200 us on a box of its own; with several
    # ranks on one node the polling threads of all of them share the node's CPUs, so the budget is split
    if "LF_SPIN_US" not in os.environ and args.mode == "ranks":
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        if lw > 1:
            os.environ["LF_SPIN_US"] = str(max(50, 200 // lw))

    import torch
    dist, ctl, backend = None, None, None
    if world > 1:
        import datetime

        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # LF_BENCH_BACKEND=gloo is a test hook: it lets the N > 1 code path run on a box with fewer GPUs than ranks
        # (ranks then share devices, bulk data goes through host memory); the driver's runs use RCCL ("nccl"), one rank per GPU
        backend = os.environ.get("LF_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=max(600, int(args.exchange_timeout) * 2)))
        ctl = dist.new_group(backend="gloo") if backend == "nccl" else dist.group.WORLD      # metadata: host tensors
    dev = torch.device("cuda", local)
    bulk = dev if (not dist or backend == "nccl") else torch.device("cpu")      # where scattered reads / gathered SAM live
    rdev = bulk                                                                 # where the reduced scalars live

    import lordfast_amd as la
    from lordfast_amd import dist as lfd
    if la.device_count() < 1:
        sys.exit("bench.py: no gfx950 device visible; the HIP path has no CPU fallback")
    n_dev = args.gpus if args.mode == "inproc" else 1
    if args.mode == "inproc" and la.device_count() < n_dev and not os.environ.get("LF_BENCH_SHARE_DEVICES"):
        sys.exit(f"bench.py --mode inproc --gpus {n_dev}: only {la.device_count()} devices visible")

    fa, contigs = ensure_index(args, rank)
    if dist:
        dist.barrier()
    t0 = time.time()
    lf = la.LordFast(fa, device=local, full_sa=True)
    # the product path and nothing else: no host cross-check implementation may be switched in (they live in a test library, liblfxcheck.so)
    lf.L.lf_debug_crosscheck.restype = C.c_uint
    lf.L.lf_debug_crosscheck.argtypes = [C.c_uint]
    xmask = lf.L.lf_debug_crosscheck(0)
    assert xmask == 0, f"lf_debug_crosscheck mask was {xmask}: a cross-check implementation was switched in"
    lf.L.lf_device_copy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    replicas = [lf] + [la.LordFast(fa, device=d % la.device_count(), full_sa=True) for d in range(1, n_dev)]
    log(f"rank {rank}: index resident in HBM after {time.time() - t0:.1f}s")
    kk, cc = (17, 2000) if args.config == "c5" else (14, 1000)             # -k 14 -c 1000 --chainAlg dp-n2 (C2); -k 17 -c 2000 (C5)
    params = la.default_params(min_anchor_len=kk, sampling_count=cc,
                               chain_alg=1 if args.chain_alg == "clasp" else 0, max_map=args.max_map)

    # ---- the job's reads.  weak: every rank has its own --reads (seed 2024 + rank); strong: ONE --reads set, cut by bases
    strong = args.scaling == "strong" and world * n_dev > 1
    exchange = world > 1 and args.exchange in ("auto", "on")
    if strong:
        all_names, all_seqs = make_reads(args, contigs, fa, 0)
        bounds = lfd.shard_bounds([len(x) for x in all_seqs], world)
        lo, hi = bounds[rank]
        names, seqs = all_names[lo:hi], all_seqs[lo:hi]
        n_total = len(all_seqs)
    else:
        names, seqs = make_reads(args, contigs, fa, rank)
        for r in range(1, n_dev):                                         # inproc: the one process holds every device's share
            nr, sr = make_reads(args, contigs, fa, r)
            names, seqs = names + nr, seqs + sr
        n_total = args.reads * world * n_dev
    own = lfd.make_shards(torch, names, seqs, 1, bulk)[0][0]                  # this rank's shard, resident on the bulk device
    own_na = own.name_array(torch)                                             # (built here, once: the steps in flight share it)
    job_shards = None
    if exchange and rank == 0:
        # rank 0 OWNS the whole job's read batch, resident in its HBM, packed once outside the timed region (the way a reader
        # leaves a chunk in memory): shard r = what rank r would map by itself
        if strong:
            job_shards = [lfd.make_shards(torch, all_names[a:b], all_seqs[a:b], 1, bulk)[0][0] for a, b in bounds]
        else:
            job_shards = [own]
            for r in range(1, world):
                nr, sr = make_reads(args, contigs, fa, r)
                job_shards.append(lfd.make_shards(torch, nr, sr, 1, bulk)[0][0])
        log(f"rank 0 owns the job: {sum(len(s) for s in job_shards)} reads, {sum(s.nbytes for s in job_shards) / 1e9:.2f} GB in HBM")
    if strong:
        del all_names, all_seqs
    contigs = None
    bases_local = sum(len(s) for s in seqs)

    # host thread budget: the cgroup quota (or the online CPUs) split between the ranks of this node
    budget = host_budget()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    params.threads = max(2, min(255, budget // max(1, local_world)))
    if os.environ.get("LF_BENCH_THREADS"):
        params.threads = int(os.environ["LF_BENCH_THREADS"])
    log(f"rank {rank}: {params.threads} host threads (budget {budget}, {local_world} ranks on this node)")

    # SAM buffers, reused by every step.  dev_out: HBM (value); host_out: pinned host memory (PCIe-inclusive rate)
    per_rank_reads = max(len(seqs), -(-n_total // world))
    # a record is ~1.7 x its read (SEQ + CIGAR + MD); -n > 1 on duplicated reads prints every candidate's record with the full SEQ
    cap_one = int(3.0 * (1.0 + 4.0 * args.dup_frac) * max(bases_local, per_rank_reads * args.read_len) * 1.1) + per_rank_reads * 2048 + (1 << 20)
    read_cap = int(1.25 * max(bases_local, per_rank_reads * args.read_len) * 1.1) + per_rank_reads * 64 + (1 << 20)
    dev_out = torch.empty(cap_one, dtype=torch.uint8, device=dev)
    try:
        host_out = torch.empty(cap_one * n_dev, dtype=torch.uint8, pin_memory=True)
    except RuntimeError:
        host_out = torch.empty(cap_one * n_dev, dtype=torch.uint8)
    fixed_arrays = (la.api._cstr_array(names), la.api._cstr_array(seqs))
    seq_lens = np.array([len(x) for x in seqs], dtype=np.uint32)          # Read.length of the reference's records
    stage_in = torch.empty(read_cap, dtype=torch.uint8, device=dev) if bulk.type == "cpu" and dist else None

    class _Sam:                                                          # head()/len() like api.SamBuffer
        def __init__(self, t, n):
            self.t, self.n = t, n

        def __len__(self):
            return self.n

        def head(self, k):
            k = min(k, self.n)
            if self.t.device.type == "cpu":
                return bytes(self.t[:k].numpy().tobytes())
            buf = (C.c_char * k)()                           # through the library's own copy, not torch's (see digest)
            la.api._check(lf.L.lf_device_copy(local, buf, C.c_void_p(self.t.data_ptr()), C.c_size_t(k)), "lf_device_copy")
            return bytes(buf)

    # --inflight D: slot j > 0 has output (and, gloo hook, staging) buffers of its own (slot 0: dev_out / host_out / stage_in)
    D = max(1, args.inflight)
    dev_outs = [dev_out] + [torch.empty(cap_one, dtype=torch.uint8, device=dev) for _ in range(D - 1)]
    host_outs = [host_out] + [torch.empty(cap_one * n_dev, dtype=torch.uint8, pin_memory=True) for _ in range(D - 1)]
    stage_ins = [stage_in] + [torch.empty(read_cap, dtype=torch.uint8, device=dev) if stage_in is not None else None for _ in range(D - 1)]

    def map_shard(shard, out, j=0):
        """one step of the product path on a device-resident shard -> SAM text in `out` (HBM, or -- gloo hook -- host); j: the
        calling thread's slot when several steps are in flight"""
        na = shard.name_array(torch)
        blob = shard.blob
        if blob.device.type == "cpu":                                   # gloo hook: what arrived in host memory is staged into HBM
            if stage_ins[j].numel() < shard.nbytes:
                stage_ins[j] = torch.empty(shard.nbytes + (1 << 20), dtype=torch.uint8, device=dev)
            stage_ins[j][:shard.nbytes].copy_(blob[:shard.nbytes]); torch.cuda.current_stream().synchronize()
            blob = stage_ins[j]
        target = out if out.device.type == "cuda" else dev_outs[j]
        t_call = time.perf_counter()
        ln, st = lf.map_batch_dev(na, blob.data_ptr(), shard.seq_off[:-1], shard.seq_lens, target.data_ptr(), target.numel(), True, params=params)
        st["ms_python_call"] = (time.perf_counter() - t_call) * 1e3
        if target is not out:
            out[:ln].copy_(target[:ln])
        return ln, st

    def step_hbm(j=0):
        t_call = time.perf_counter()
        ln, st = lf.map_batch_dev(own_na, own.blob.data_ptr(), own.seq_off[:-1], own.seq_lens, dev_outs[j].data_ptr(), dev_outs[j].numel(), True, params=params) \
            if (j > 0 and own.blob.device.type == "cuda") else map_shard(own, dev_outs[j], j)
        st.setdefault("ms_python_call", (time.perf_counter() - t_call) * 1e3)
        return _Sam(dev_outs[j], ln), st

    def step_host(j=0):
        t_call = time.perf_counter()
        ho = host_outs[j]
        if n_dev > 1:
            ln, st = la.api.map_batch_multi_into(replicas, names, ho.data_ptr(), ho.numel(), params=params,
                                                 name_arr=fixed_arrays[0], seq_arr=fixed_arrays[1], seq_lens=seq_lens)
        else:
            ln, st = lf.map_batch_into(names, None, ho.data_ptr(), ho.numel(), params=params, name_arr=fixed_arrays[0],
                                       seq_arr=fixed_arrays[1], seq_lens=seq_lens)
        st["ms_python_call"] = (time.perf_counter() - t_call) * 1e3
        return _Sam(ho, ln), st

    def add(agg, st):
        if agg is None:
            return dict(st)
        for k, v in st.items():
            agg[k] += v
        return agg

    def run_steps(fn, n_steps):
        """n_steps complete calls, D of them in flight; -> (summed stats, last SAM)"""
        agg, sam = None, None
        if D == 1:
            for _ in range(n_steps):
                sam, st = fn()
                agg = add(agg, st)
        else:
            # D steps in flight: thread j runs steps j, j + D, ... (the calls release the GIL); EXACTLY n_steps complete calls
            from concurrent.futures import ThreadPoolExecutor

            def worker(j):
                a, last = None, None
                for _k in range(j, n_steps, D):
                    last, st = fn(j)
                    a = add(a, st)
                return a, last
            with ThreadPoolExecutor(D) as ex:
                for a, last in ex.map(worker, range(D)):
                    if a is not None:
                        for k in ("ms_python_call",):
                            a.setdefault(k, 0.0)
                        agg = dict(a) if agg is None else {k: agg.get(k, 0) + v for k, v in a.items()}
                    sam = sam if sam is not None else last
        return agg, sam

    def timed(fn, n_steps):
        """EXACTLY n_steps steps between barrier + synchronize on both sides; elapsed = max over ranks"""
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cpu0 = sum(os.times()[:4])
        agg, sam = run_steps(fn, n_steps)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        cpu_s = sum(os.times()[:4]) - cpu0
        if dist:
            tmax = torch.tensor([el], dtype=torch.float64, device=rdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, cpu_s, agg, sam

    def digest(sam):
        """(xxh3-128 of the whole text, its length).  Device text is brought over by the LIBRARY's copy (lf_device_copy, one
        D2H into the pinned buffer) -- not by torch: a torch-side `.cpu()` of a few GB between two regions left the next
        library call waiting for the GPU forever (two runs of two; the runtime's device-wide wait with the lanes' ~10^2
        streams alive, see DESIGN.md 'runtime trap')."""
        import xxhash
        n = len(sam)
        if sam.t.device.type != "cpu":
            if n > host_out.numel():
                return None
            la.api._check(lf.L.lf_device_copy(local, C.c_void_p(host_out.data_ptr()), C.c_void_p(sam.t.data_ptr()), C.c_size_t(n)), "lf_device_copy")
            view = host_out.numpy()
        else:
            view = sam.t.numpy()
        h = xxhash.xxh3_128()
        for a in range(0, n, 1 << 28):
            h.update(memoryview(view[a:min(n, a + (1 << 28))]))
        return h.hexdigest(), n

    # ---- 1. the shard resident in this rank's HBM, SAM text left in HBM, no exchange (`value_hbm_resident`) ----
    hbm_step = step_host if args.mode == "inproc" else step_hbm
    if args.warmup:
        run_steps(hbm_step, args.warmup * D)            # every slot in flight warms its lanes' buffers
    elapsed_nx, cpu_s_hbm, agg_hbm, sam_hbm = timed(hbm_step, args.steps)
    snap_hbm = digest(sam_hbm) if (rank == 0 and world == 1 and not args.no_exclusive) else None
    # ---- 2. the same steps through the host-buffer boundary the reference has: reads in host memory -> SAM records in host memory
    #         (H2D and D2H inside the step).  N = 1: this is `value` ----
    elapsed_host = cpu_s_host = agg_host = sam_host = None
    snap_host = head_host = None
    order = None
    if args.mode != "inproc" and not args.no_host_region:
        run_steps(step_host, max(1, args.warmup) * D)
        elapsed_host, cpu_s_host, agg_host, sam_host = timed(step_host, args.steps)
        if rank == 0 and world == 1:
            # the records the TIMED steps wrote, before anything else runs: a digest of all of them and the head that is compared with the reference
            snap_host = digest(sam_host)
            # (the reference maps ~ 1 k reads/s on the box's cores: room for 1.5 x 1.2 k reads/s x the sample time, ~1.75 x read_len + header per record, x candidates)
            # the reads the reference will be timed on: a golden-ratio stride over the WHOLE batch (any prefix of the list covers every
            # chunk and lane); their records are taken out of the timed output now, before any other pass touches the buffer
            if not args.no_cpu_baseline:
                order = sample_order(len(seqs), int(1.5 * args.cpu_seconds * 1200 * 15000 / max(1000, args.read_len)) + 256)
                t_x = time.time()
                head_host = records_of(sam_host.t.numpy(), len(sam_host), [names[i] for i in order])
                log(f"records of {len(head_host)} sampled reads taken out of the timed output in {time.time() - t_x:.1f}s")
    elif args.mode == "inproc":
        elapsed_host, cpu_s_host, agg_host, sam_host = elapsed_nx, cpu_s_hbm, agg_hbm, sam_hbm
    # ---- 2b. the same host boundary with a MAPPER-READY batch (lf_batch_create / lf_map_batch_from): lengths and the bit planes the batch crosses the link as
    #          are made once, when the batch is made -- the analogue of the reference's readChunk output, which its mapping timer excludes
    #          (src/Reads.cpp:84-104, src/baseFAST.cpp:59-75).  `value` stays the region above (the caller hands over plain strings every step). ----
    elapsed_pre = cpu_s_pre = snap_pre = t_create = None
    if args.mode != "inproc" and not args.no_host_region and world == 1 and n_dev == 1 and hasattr(la, "ReadBatch"):
        t_c = time.perf_counter()
        pre_batch = la.ReadBatch(names, seqs, None, min_read_len=params.min_read_len, seq_lens=seq_lens)
        t_create = time.perf_counter() - t_c

        def step_pre(j=0):
            t_call = time.perf_counter()
            ho = host_outs[j]
            ln, st = lf.map_batch_from(pre_batch, ho.data_ptr(), ho.numel(), params=params)
            st["ms_python_call"] = (time.perf_counter() - t_call) * 1e3
            return _Sam(ho, ln), st
        run_steps(step_pre, max(1, args.warmup) * D)
        elapsed_pre, cpu_s_pre, _agg_pre, sam_pre = timed(step_pre, args.steps)
        if rank == 0 and not args.no_exclusive:
            snap_pre = digest(sam_pre)
        pre_batch.close()
    primary_is_host = world * n_dev == 1 and elapsed_host is not None or args.mode == "inproc"
    agg, cpu_s, sam = (agg_host, cpu_s_host, sam_host) if primary_is_host else (agg_hbm, cpu_s_hbm, sam_hbm)

    elapsed, xinfo = (elapsed_host if primary_is_host else elapsed_nx), None
    bases_total = bases_local
    if dist:
        bsum = torch.tensor([bases_local], dtype=torch.int64, device=rdev)
        dist.all_reduce(bsum, op=dist.ReduceOp.SUM)
        bases_total = int(bsum.item())

    # ---- weak-scaling rate next to a strong-scaling line: every rank maps --reads reads of its own (seed 2024 + rank), HBM-resident
    elapsed_weak = None
    if strong and dist and not os.environ.get("LF_BENCH_NO_WEAK"):
        wn, ws = make_reads(args, None, fa, rank)
        wshard = lfd.make_shards(torch, wn, ws, 1, bulk)[0][0]
        wcap = int(3.0 * sum(len(x) for x in ws) * 1.1) + len(ws) * 2048 + (1 << 20)
        wout = dev_out if wcap <= dev_out.numel() else torch.empty(wcap, dtype=torch.uint8, device=dev)
        del wn, ws
        def step_weak(j=0):
            ln, st = map_shard(wshard, wout)
            return _Sam(wout, ln), st
        step_weak()
        elapsed_weak, _, _, _ = timed(step_weak, args.steps)
        del wshard, wout

    # ---- exclusive pass (rank 0, outside the timed region): ONE step with one chunk at a time and the alignment size classes
    # on one stream, so that every HIP-event bracket is the kernel (group) ALONE on the GPU.  The timed steps above keep
    # eight chunks in flight: their brackets overlap and are only reported as `overlapped_bracket_ms`.
    excl = None
    snap_excl = None
    if rank == 0 and args.no_exclusive:
        excl = {k: (v / args.steps if isinstance(v, (int, float)) else v) for k, v in agg_hbm.items()}      # overlapped brackets instead
    elif rank == 0:
        saved = {k: os.environ.get(k) for k in ("LF_LANES", "LF_SERIAL_CLASSES", "LF_CHUNK_READS", "LF_CHUNK_BASES")}
        os.environ["LF_LANES"] = "1"; os.environ["LF_SERIAL_CLASSES"] = "1"
        if args.dup_frac > 0 and not os.environ.get("LF_CHUNK_READS"):
            # -n 30 on duplicated reads: a 100 k-read chunk's working set (every candidate's records with the full SEQ) does not fit
            # beside the 16-mer table; the exclusive pass keeps 25 k-read chunks, still one at a time on one lane
            os.environ["LF_CHUNK_READS"] = "25000"
        elif not os.environ.get("LF_BENCH_EXCL_CHUNKED"):
            # the whole batch as ONE chunk: every kernel is launched once per step over all of the step's work, so "alone on the
            # GPU" also means "with enough wavefronts to fill it" (a 25 k-read chunk left the small size classes with < 2 waves per SIMD)
            os.environ["LF_CHUNK_READS"] = str(1 << 30); os.environ["LF_CHUNK_BASES"] = str(1 << 40)
        try:
            hbm_step()                      # the one-chunk-at-a-time mode uses larger chunks: let the grow-only buffers settle
            sam_x, excl = hbm_step()
            if world == 1 and args.mode != "inproc":
                snap_excl = digest(sam_x)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    if dist:
        dist.barrier()

    try:
        free_b, total_b = torch.cuda.mem_get_info(local)
        hbm_used_gb = (total_b - free_b) / 1e9
    except Exception:                                                    # noqa: BLE001
        hbm_used_gb = None

    index_desc = lf.describe()
    try:
        bases_genome = (os.path.getsize(fa + ".pac") - 1) * 4
    except OSError:
        bases_genome = 0

    def report(elapsed, sam, xinfo):
        nonlocal order
        K = args.steps
        bases = bases_total
        value = n_total * K / elapsed
        per_rank = n_total * K / world
        # ---- roofline: algorithmic bytes (SURVEY 8d counters emitted by the kernels) / EXCLUSIVE kernel time ----
        # Durations come from the exclusive pass (one step = one launch sequence per chunk; HIP events on the launch
        # streams; nothing else on the GPU).  Their sum is below ms_per_step; profiles/ holds the rocprofv3
        # --kernel-trace --stats summary of the same serialized command.  The first block are single kernels (the event bracket
        # holds that kernel alone); the second block are small groups (a stage's helper kernels around its main kernel).
        def kernel_table(a):
            n_hits = a["n_seeds"]
            return {
                "lf_seed_search_kernel": (a["ms_k_search"], 16 * a["n_cache"] + 64 * a["n_occblk"] + a["n_readbytes"], a["search_launches"], True),
                "lf_edlib_rsweep_kernel": (a["ms_k_rsweep"], a["ext_bytes"] - a["ops_bytes"], max(1, 2 * a["edlib_launches"]), True),      # NW + SHW instantiation per round: q + t/4 read
                "lf_edlib_tb_kernel": (a["ms_k_tb"], a["ops_bytes"], max(1, a["edlib_launches"]), True),                               # the paths written (q + t region per problem)
                "lf_seed_locate_kernel": (a["ms_k_locate"], 8 * a["n_sa"] + 9 * a["n_sa"], a["locate_launches"], True),
                "lf_ksw_r4_kernel": (a["ms_k_ksw"], a.get("ksw_bytes", 0), max(1, a["n_ksw_problems"] and 1), True),      # (bands above 241 columns: lf_ksw_mw_kernel / lf_ksw_kernel)
                # ---- groups ----
                # lf_vote_cell_kernel (+ request-count scan): 9 B per hit read; votes live in LDS
                "lf_vote_cell_kernel (+ scan)": (a["ms_k_vote"], 9 * n_hits, a["search_launches"], False),
                # request gather + sort by qPos + lf_chain_n2_kernel + chain gather: 16 B per request seed, 8 B per chain seed
                "lf_chain_* (gather, sort, dp-n2 | clasp)": (a["ms_k_chain"], 16 * a["n_req_seeds"], a["search_launches"], False),
                "lf_hirsch_* (levels incl. per-level readbacks)": (a["ms_k_hirsch"], a.get("hirsch_bytes", 0), max(1, a["edlib_launches"]), False),
                "lf_desc_* (alignment binning: keys, sort, segments, scan, build)": (a["ms_k_bin"], 100 * a["n_edlib_problems"], max(1, a["edlib_launches"]), False),
                "lf_render_kernel (+ caps, scan)": (a["ms_k_render"], a["ops_bytes"] + a["render_bytes"], max(1, a["render_launches"]), False),      # single pass: ops read once, text written once
            }
        kx = kernel_table(excl)
        pmc, pmc_src = load_pmc(args, world)
        sq, isa, sq_src = load_sq(args, world)
        by_kernel = {}
        for kname, (kms, kbytes, kl, single) in kx.items():
            gbs = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
            traffic = None
            if pmc is not None:
                prefix = family_prefixes(kname)
                fam = [v for k, v in pmc.items() if k.startswith(prefix) and isinstance(v, dict)]
                if fam:      # the counter passes profile exactly ONE step (--no-exclusive --steps 1 --warmup 0): bytes per step
                    traffic = sum(v.get("fetch_kb", 0.0) + v.get("write_kb", 0.0) for v in fam) * 1024.0
            by_kernel[kname] = dict(ms_per_step=kms, launches_per_step=kl, avg_launch_ms=kms / max(1, kl), algorithmic_GB_per_step=kbytes / 1e9,
                                    achieved_GBps=gbs, frac_of_8TBps=gbs / 8000.0, single_kernel=single,
                                    hbm_traffic_GB_per_step=(traffic / 1e9 if traffic is not None else None),
                                    alu=alu_of(kname, kms, sq, isa, sq_src, args.reads))
        other = max(0.0, excl["ms_k_edlib"] - excl["ms_k_rsweep"] - excl["ms_k_tb"] - excl["ms_k_hirsch"] - excl["ms_k_bin"])      # large-leaf sweeps, stitch
        excl_sum = sum(v[0] for v in kx.values()) + other
        dom = max(kx, key=lambda k: kx[k][0])      # the entry with the largest exclusive time (a group is named by its main kernel; `single_kernel` says which it is)
        dms, dbytes, dl, _ = kx[dom]
        achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
        traffic = by_kernel[dom]["hbm_traffic_GB_per_step"]
        # what bounds an entry: its VALU-issue fraction where a counter pass of this tree gives one and it exceeds the HBM fraction; the kernels of this path are integer /
        # latency bound ("integer ALU + LDS, not HBM-bound" for the alignment kernels), the HBM figure is reported for all of them as the contract asks
        for kname, e in by_kernel.items():
            af = (e.get("alu") or {}).get("frac")
            e["bound"] = "valu" if (af is not None and af > e["frac_of_8TBps"]) else "hbm"
        roofline = dict(bound="hbm", bound_by_issue=by_kernel[dom]["bound"] == "valu", kernel=dom.split(" ")[0], achieved=achieved, peak=8000.0, unit="GB/s", frac=achieved / 8000.0,
                        traffic=(traffic * 1e9 / max(1, dl) if traffic is not None else None), traffic_source=pmc_src,
                        launches_per_step=int(dl), avg_launch_ms=dms / max(1, dl), algorithmic_bytes_per_launch=dbytes / max(1, dl),
                        exclusive_ms_per_step=dms, exclusive_ms_sum_all_kernels=excl_sum,
                        alignment_group_exclusive_ms=excl["ms_k_edlib"], by_kernel=by_kernel,
                        measured="exclusive pass inside bench.py: the whole batch as ONE chunk on one lane (LF_LANES=1, LF_CHUNK_READS / _BASES unlimited), the "
                                 "alignment classes on one stream (LF_SERIAL_CLASSES=1), HIP events on the launch streams, one step after the timed region; "
                                 "the rocprofv3 summary of the same mode is under profiles/",
                        overlapped_bracket_ms_per_step={k: round(v[0] / K, 2) for k, v in kernel_table(agg_hbm).items()},
                        lanes_in_flight_timed_region=(int(os.environ["LF_LANES"]) if os.environ.get("LF_LANES") else (8 if primary_is_host else 4)) if params.threads >= 4 else params.threads)
        # the issue-bound kernels' own roofline: VALU wave-instructions (SQ counters of this tree) x issue cycles per form (static mix of the
        # kernel's loops, profiles/tools/isa_mix.py; rates of profiles/tools/ubench/valu_rate.hip) / 1024 SIMDs / sustained clock vs exclusive time
        roofline["alu"] = by_kernel[dom]["alu"]
        shard_desc = (f"the SAME {n_total}-read set cut by bases over {world} ranks" if strong else f"{args.reads} reads per GPU")
        io_host = "reads in host memory -> SAM records in host memory (lf_map_batch_into_lens): H2D and D2H inside the step"
        io_hbm = "read bases resident in HBM when the timed region starts, SAM records left in HBM (lf_map_batch_dev)"
        with_x = bool(exchange and xinfo and xinfo.get("status") == "ok")
        out = {
            "metric": "aligned reads/s", "value": value, "unit": "reads/s", "n_gpus": world * n_dev, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64", "data": ("synthetic-error reads drawn from a real reference FASTA" if args.ref_fasta else "synthetic"),
            "config": {"workload": f"{args.config}: {shard_desc}, synthetic {'ONT' if args.config == 'c5' else 'PacBio'} reads (~{args.read_len} bp, {args.err:.0%} err"
                                   + (f", {args.dup_frac:.0%} of them from 2-4-copy segmental duplications at 1-3 % divergence" if args.dup_frac > 0 else "") + f") vs "
                                   + (f"the reference {os.path.basename(args.ref_fasta)} ({bases_genome / 1e6:.0f} Mbp)" if args.ref_fasta else f"{args.genome_mbp:g} Mbp synthetic genome ({args.repeat_profile} repeats)")
                                   + f", -k {kk} -c {cc} --chainAlg {args.chain_alg}" + (f" -n {args.max_map}" if args.max_map != 10 else "")
                                   + (f"; every step maps the SAME {n_total} reads ({K} steps = {n_total * K} mapped reads, not {n_total * K} different ones)" if K > 1 else ""),
                       "reads_total": n_total, "reads_mapped_in_timed_region": n_total * K, "distinct_reads": n_total, "mean_read_len": bases / max(1, n_total), "genome_mbp": (bases_genome / 1e6 if args.ref_fasta else args.genome_mbp),
                       "io": ("host buffers (lf_map_batch_multi): one process, every device copies through its own PCIe link" if args.mode == "inproc" else
                              "rank 0 owns the job's reads in its HBM and ends with the job's SAM records there; scatter / gather over RCCL inside the step" if with_x else
                              io_host if primary_is_host else io_hbm),
                       "parallelism": (f"one process drives {n_dev} devices (lf_map_batch_multi), index replicated" if args.mode == "inproc" else
                                       f"rank 0 owns the {n_total}-read job in HBM: RCCL point-to-point scatter of the packed reads + gather of the SAM "
                                       f"records, pipelined with the mapping, inside the timed region; {world} GPUs, index replicated" if with_x else
                                       f"reads sharded over {world} GPU(s), index replicated, no data-path collective"),
                       "index": index_desc},
            "gbp_per_s": bases * K / elapsed / 1e9, "host_cpu_seconds_per_step": cpu_s / K, "hbm_used_gb": hbm_used_gb,
            "host_ms_per_step": {k: agg[k] / K for k in ("ms_total", "ms_python_call", "ms_seed", "ms_vote", "ms_chain", "ms_extend", "ms_render", "ms_sam")},
            "per_read": {"seeds": agg["n_seeds"] / per_rank, "edlib_problems": agg["n_edlib_problems"] / per_rank,
                         "seed_bytes": (16 * agg["n_cache"] + 64 * agg["n_occblk"] + 8 * agg["n_sa"] + agg["n_readbytes"]) / per_rank,
                         "ext_bytes": agg["ext_bytes"] / per_rank, "cigar_md_text_bytes": agg["render_bytes"] / per_rank,
                         "dp_block_steps": agg["dp_block_steps"] / per_rank,
                         "chain_requests": agg["n_chain_problems"] / per_rank, "tie_requests": agg["n_tie_requests"] / per_rank,
                         "ksw_problems": agg["n_ksw_problems"] / per_rank,
                         "stale_first_windows_per_step": agg.get("n_stale_first_windows", 0) / K},
            "roofline": roofline,
            "host_waits_per_chunk": agg.get("n_host_waits", 0) / max(1, agg.get("n_chunks", 0)), "chunks_per_step": agg.get("n_chunks", 0) / K,
            "steps_in_flight": D,
            "source_tree": tree_hash(),      # digest of lordfast_amd/csrc: which kernels produced this line
        }
        # every region that was timed, under its own name
        out["value_hbm_resident"] = n_total * K / elapsed_nx               # no exchange, shards resident in HBM, SAM left in HBM
        out["ms_per_step_hbm_resident"] = elapsed_nx / K * 1e3
        out["host_cpu_seconds_per_step_hbm_resident"] = cpu_s_hbm / K
        if elapsed_host is not None:
            out["value_host_boundary"] = n_total * K / elapsed_host       # every rank: its reads in host memory -> its SAM records in host memory
            out["ms_per_step_host_boundary"] = elapsed_host / K * 1e3
            out["host_cpu_seconds_per_step_host_boundary"] = cpu_s_host / K
            out["host_boundary_bytes_per_step"] = {"h2d_bases": bases, "d2h_sam_text": int(len(sam_host)) if sam_host is not None else None,
                                                   "sam_egress": "SEQ / QUAL columns filled on the host (lf_sam.hip HOLES mode)" if not os.environ.get("LF_SAM_FULL") else "whole lines from the device"}
        if elapsed_pre is not None:
            out["value_prepacked_batch"] = n_total * K / elapsed_pre       # host boundary with a mapper-ready batch (lf_batch_create once, lf_map_batch_from per step)
            out["ms_per_step_prepacked_batch"] = elapsed_pre / K * 1e3
            out["host_cpu_seconds_per_step_prepacked_batch"] = cpu_s_pre / K
            out["prepacked_batch"] = {"io": "lf_batch_create (lengths + bit planes of the batch in pinned host memory: OUTSIDE the step, made once -- the reference's readChunk is "
                                            "outside its mapping timer the same way) -> lf_map_batch_from per step: the planes' bit ranges cross the link, SAM records come back "
                                            "into the caller's pinned buffer, SEQ / QUAL filled from the caller's strings",
                                      "create_seconds": t_create}
        if elapsed_weak is not None:
            out["value_weak_hbm_resident"] = args.reads * world * K / elapsed_weak
            out["ms_per_step_weak_hbm_resident"] = elapsed_weak / K * 1e3
        if xinfo:
            out["exchange"] = xinfo
        if args.dup_frac > 0:
            ndup = sum(1 for nm in names if nm.endswith(b"_dup"))
            out["per_read"]["dup_reads_fraction"] = ndup / max(1, len(names))
            if dup_probe is not None:
                out["per_read"]["chain_requests_on_dup_reads"] = dup_probe
        # ---- parity: the records of the TIMED steps ----
        if snap_host is not None or snap_hbm is not None or snap_excl is not None:
            snaps = {k: v for k, v in (("host_boundary_timed_steps", snap_host), ("prepacked_batch_timed_steps", snap_pre), ("hbm_resident_timed_steps", snap_hbm), ("exclusive_pass", snap_excl)) if v is not None}
            out["sam_digests"] = {k: {"xxh3_128": v[0], "bytes": v[1]} for k, v in snaps.items()}
            out["timed_output_equals_exclusive_pass_output"] = len({v for v in snaps.values()}) == 1 if len(snaps) > 1 else None
        if not args.no_cpu_baseline and world == 1 and (head_host is not None or sam is not None):
            try:
                if order is None:
                    order = sample_order(len(seqs), int(1.5 * args.cpu_seconds * 1200 * 15000 / max(1000, args.read_len)) + 256)
                cb = cpu_baseline(args, fa, [names[i] for i in order], [seqs[i] for i in order])
            except Exception as e:                                          # noqa: BLE001
                log("cpu baseline failed:", e)
                cb = None
            if cb:
                base, ref_sam, n = cb
                out["cpu_baseline"] = base
                # same boundary on both sides: host memory -> host memory
                out["speedup_vs_cpu_baseline"] = value / base["value"]          # vs `cpus_granted` host CPUs (not an optimisation target)
                out["speedup_vs_cpu_baseline_hbm_resident"] = out["value_hbm_resident"] / base["value"]
                # bit-match against the reference on the sampled reads (a stride over the whole batch).  `head_host` holds their records
                # as the TIMED host-boundary steps wrote them (taken before any other pass ran).  Primary records (the BASELINE metric)
                # and ALL records of a read (secondaries, supplementaries: same lines, same order) are compared.
                def by_read(txt):
                    d = {}
                    for l in txt.split(b"\n"):
                        if l:
                            d.setdefault(l.split(b"\t", 1)[0], []).append(l)
                    return d
                if head_host is not None:
                    mine = head_host
                    out["records_compared_come_from"] = "the timed host-boundary steps (taken out of their output before the exclusive pass)"
                else:
                    view = sam.t.numpy() if sam.t.device.type == "cpu" else np.frombuffer(sam.head(len(sam)), dtype=np.uint8)
                    mine = records_of(view, len(sam), [names[i] for i in order[:n]])
                    out["records_compared_come_from"] = "the last pass"
                want = by_read(ref_sam)
                hit = tot = hit_all = 0
                for nm, lines in want.items():
                    got = mine.get(nm)
                    if got is None:
                        continue
                    tot += 1
                    hit_all += (got == lines)
                    pw = [l for l in lines if not (int(l.split(b"\t")[1]) & (256 | 2048))]
                    pg = [l for l in got if not (int(l.split(b"\t")[1]) & (256 | 2048))]
                    hit += (pw == pg)
                out["parity_sample"] = f"golden-ratio stride over all {len(seqs)} reads of the step: read indices {min(order[:n])} .. {max(order[:n])}"
                out["primary_record_match_rate"] = hit / max(1, tot)
                out["all_records_match_rate"] = hit_all / max(1, tot)
                out["reads_compared"] = tot
                out["reads_in_reference_sample"] = len(want)
                out["records_compared"] = sum(len(v) for k, v in want.items() if k in mine)
        if sam is not None and os.environ.get("LF_BENCH_SAM_DIGEST"):
            # test hook: a digest of the job's gathered records (compared with the 1-rank run of the same read set)
            out["sam_md5"] = hashlib.md5(sam.head(len(sam))).hexdigest()
            out["sam_bytes"] = len(sam)
        print(json.dumps(out), flush=True)

    # chain requests per read on the duplicated fraction alone (one untimed call on a sample of them): fine mode must really happen
    dup_probe = None
    if args.dup_frac > 0 and rank == 0:
        di = [i for i, nm in enumerate(names) if nm.endswith(b"_dup")][:2000]
        if di:
            _, dst = lf.map_batch([names[i] for i in di], [seqs[i] for i in di], params=params, copy=False)
            dup_probe = dst["n_chain_problems"] / len(di)

    # ---- 3. N > 1: the pipelined exchange, inside the timed region ----
    if exchange:
        # D steps of every rank in flight (ring of D + 1 buffers): lordfast_amd/dist.py: run_pipeline.  Rank 0's gather buffers hold a
        # whole step's records: sized from the job's bases (a record is ~1.7 x its read; cap_one's generous per-rank bound x world x
        # ring would not fit beside the index under weak scaling)
        job_bases = sum(sh.bases() for sh in job_shards) if rank == 0 else 0
        gcap = int(2.2 * (1.0 + 4.0 * args.dup_frac) * job_bases) + n_total * 1024 + (64 << 20) if rank == 0 else None
        px = lfd.PipelinedExchange(dist, torch, bulk, ctl, read_cap, cap_one, ring=D + 1, gather_cap=gcap)
        state = {"done": False}
        from concurrent.futures import ThreadPoolExecutor
        xpool = ThreadPoolExecutor(D)
        xagg = {}
        xlock = threading.Lock()

        def map_step(k, shard, out, j):
            ln, st = map_shard(shard, out, j)
            with xlock:
                xagg["st"] = add(xagg.get("st"), st)
            return ln

        def run_exchange(n_steps):
            """EXACTLY n_steps steps, D of them in flight on every rank; reads of step t + 1 go out and records of step t - D come back
            as one grouped RCCL call per tick; the first scatter and the last gather are exposed"""
            xagg.pop("st", None)
            lfd.run_pipeline(px, n_steps, D, (lambda k: job_shards), map_step, pool=xpool)
            return xagg.get("st")

        def watchdog():
            if not state["done"]:
                if rank == 0:
                    log(f"exchange phase still running after {args.exchange_timeout:.0f}s: reporting the no-exchange rate")
                    state["timeout"] = True
                    report(elapsed_nx, None, dict(status="timeout"))       # the line says status "timeout" and carries the no-exchange rate ...
                os._exit(3)                                                   # ... and the process fails: a hung exchange is not a successful N-GPU run
        timer = threading.Timer(args.exchange_timeout, watchdog)
        timer.daemon = True
        timer.start()
        x_warm = max(1, args.warmup) * D
        run_exchange(x_warm)                                              # warm-up: buffers grow (every slot in flight), RCCL connects its peers
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_exchange(args.steps)
        dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        tmax = torch.tensor([el], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        state["done"] = True
        timer.cancel()
        if rank == 0:
            gt, glen = px.gathered(args.steps - 1)
            sam = _Sam(gt, glen)                                          # the whole job's records, input order
            xinfo = dict(status="ok", transport=backend, bulk_memory=str(bulk), GB_out_per_step=px.bytes_out / (args.steps + x_warm) / 1e9,
                         GB_in_per_step=px.bytes_in / (args.steps + x_warm) / 1e9, gathered_bytes_last_step=glen,
                         steps_in_flight_per_rank=D,
                         pipeline=f"{D} step(s) of every rank in flight; reads of step t+1 and SAM of step t-{D} travel (one grouped call per tick) while they map; first scatter and last gather exposed")

    if rank == 0:
        report(elapsed, sam, xinfo)
    for h in replicas:
        h.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def load_pmc(args, world):
    """the committed FETCH_SIZE / WRITE_SIZE counter passes (per kernel, one step) -- only if they were taken on THIS source tree
    (profiles/<dir>/pmc_fetch_write_summary.json carries the digest of the tree it profiled; collect.sh writes it) and on this
    configuration; otherwise None: a stale file says nothing about the kernels that just ran."""
    import subprocess
    if not (args.genome_mbp == 3100 and args.reads == 100000 and world == 1 and args.config == "c2" and args.repeat_profile == "default" and args.dup_frac == 0 and not args.ref_fasta):
        return None, "not the profiled configuration"
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:                                                    # noqa: BLE001
        head = ""
    import glob
    src_hash = tree_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c2", "pmc_fetch_write_summary.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except Exception:                                                # noqa: BLE001
            continue
        commit = pmc.get("_meta", {}).get("source_tree") or pmc.get("_meta", {}).get("git_commit")
        if commit and commit in (head, src_hash):
            return pmc, os.path.relpath(path, ROOT)
    return None, "no counter pass of this source tree under profiles/ (profiles/tools/collect.sh writes one)"


def load_sq(args, world):
    """the committed SQ counter pass (per kernel: VALU / SALU wave-instructions, GRBM cycles; profiles/tools/collect.sh) and the static
    instruction mix of the kernels' loops (profiles/tools/isa_mix.py) -- only if both describe THIS source tree"""
    import glob
    # (the counters describe C2's kernels on C2's reads: any other configuration gets none -- round 5's C4 / C5 lines carried C2's counters scaled by read count)
    if not (args.genome_mbp == 3100 and world == 1 and args.config == "c2" and args.repeat_profile == "default" and args.dup_frac == 0 and not args.ref_fasta):
        return None, None, "not the profiled configuration (the SQ counter passes under profiles/ are C2's)"
    src_hash = tree_hash()
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c2")), reverse=True):
        try:
            sq = json.load(open(os.path.join(d, "sq_counters_50k_reads.json")))
            isa = json.load(open(os.path.join(d, "isa_mix.json")))
        except Exception:                                                # noqa: BLE001
            continue
        if sq.get("_meta", {}).get("source_tree") == src_hash and isa.get("_meta", {}).get("source_tree") == src_hash:
            # one clock for the device: GRBM_GUI_ACTIVE (summed over the 8 XCDs) over the dispatch durations of the kernels that run for milliseconds
            # (a short dispatch's counter window is longer than its start .. end), never above the 2.4 GHz the part is specified at
            g = [(v["grbm_gui_active"], v["dur_ns_in_the_grbm_pass"]) for v in sq.values() if isinstance(v, dict) and v.get("dur_ns_in_the_grbm_pass", 0) >= 2e6 and v.get("launches", 0) <= 64]
            sq["_meta"]["clock_ghz"] = min(2.4, sum(a for a, _ in g) / 8.0 / sum(b for _, b in g)) if g else 2.4
            return sq, isa, os.path.relpath(d, ROOT)
    return None, None, "no SQ counter pass + instruction mix of this source tree under profiles/ (profiles/tools/collect.sh, isa_mix.py)"


def family_prefixes(kname):
    """rocprof kernel-name prefixes of a by_kernel entry ("lf_x_* (...)" -> ("lf_x_",)); the Hirschberg levels are two families of kernels"""
    p = kname.split(" ")[0].rstrip("*")
    return ("lf_hirsch_", "lf_hband_") if p == "lf_hirsch_" else (p,)


def alu_of(kname, kms, sq, isa, src, reads_now):
    """VALU-issue roofline of one kernel (family): sum over its instantiations of wave-level VALU instructions x cycles per instruction
    of the instantiation's loop mix, over 1024 SIMDs at the clock the counter pass saw"""
    if sq is None:
        return dict(frac=None, source=src)
    prefix = family_prefixes(kname)
    reads_sq = sq.get("_meta", {}).get("reads_per_step", 50000)
    scale = reads_now / float(reads_sq)
    n_valu = n_salu = cyc = 0.0
    for k, v in sq.items():
        if not isinstance(v, dict) or not k.startswith(prefix) or "sq_insts_valu" not in v:
            continue
        mix = isa.get(k) or next((m for kk, m in isa.items() if kk.startswith(k.split("<")[0]) and isinstance(m, dict) and m.get("cycles_per_valu")), None)
        cpi = (mix or {}).get("cycles_per_valu") or 2.7
        n_valu += v["sq_insts_valu"] * scale; n_salu += v.get("sq_insts_salu", 0.0) * scale
        cyc += v["sq_insts_valu"] * scale * cpi
    if n_valu == 0:
        return dict(frac=None, source=src)
    ghz = sq.get("_meta", {}).get("clock_ghz", 2.4)
    alu_ms = cyc / 1024.0 / (ghz * 1e9) * 1e3
    return dict(bound="VALU issue (integer)", valu_wave_instructions_per_step=n_valu, salu_wave_instructions_per_step=n_salu, cycles_per_valu_instruction=cyc / n_valu,
                clock_ghz=ghz, simds=1024, alu_ms_per_step=alu_ms, exclusive_ms_per_step=kms, frac=(alu_ms / kms if kms > 0 else None), source=src)


def tree_hash():
    """digest of the kernel sources: what a profile must have been taken on to describe the kernels that ran"""
    import glob
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "lordfast_amd", "csrc", "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--tree-hash":
        print(tree_hash())
        sys.exit(0)
    main()
