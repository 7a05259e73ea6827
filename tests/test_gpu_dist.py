"""N > 1 through the REAL mapper: two ranks (both on GPU 0 of the test box, gloo for the rendezvous) push one read set
through scatter_reads_p2p -> lf_map_batch -> gather_sam_p2p and rank 0 must hold the 1-rank SAM, byte for byte
(BASELINE config C3 in miniature: the same read set sharded over N ranks)."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import gzip, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import lordfast_amd as la
from lordfast_amd import dist as lfd
from conftest import read_fasta, golden_sam, GOLDEN_CONFIGS
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cpu")
cfg = sys.argv[3]
names, seqs = read_fasta(os.path.join(sys.argv[1], "tests", "golden", "reads.fa.gz")) if rank == 0 else (None, None)
my_names, my_seqs, bounds = lfd.scatter_reads_p2p(dist, torch, names, seqs, dev)
lf = la.LordFast(os.path.join(sys.argv[2], "genome.fa"), device=0)
sam, st = lf.map_batch(my_names, my_seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
lf.close()
buf = torch.zeros(8 << 20, dtype=torch.uint8)
buf[:len(sam)] = torch.frombuffer(bytearray(sam), dtype=torch.uint8)
tot = lfd.gather_sam_p2p(dist, torch, buf, len(sam), dev)
if rank == 0:
    got = bytes(buf[:tot].numpy().tobytes())
    exp = golden_sam(cfg)
    assert got == exp, "N-rank SAM differs from the 1-rank SAM"
    assert all(hi > lo for lo, hi in bounds), bounds
    print("SHARDED_OK", bounds, len(got))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,cfg", [(2, "default"), (3, "n30")])
def test_sharded_read_set_gives_the_one_rank_sam(golden_dir, tmp_path, world, cfg):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29650 + world), str(script), ROOT, golden_dir, cfg],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SHARDED_OK" in r.stdout


# ---- bench.py's own N > 1 path: started from a bare shell (it spawns its ranks), pipelined device-resident exchange ----
import json


def _bench(workdir, *args, backend=None, timeout=1500):
    env = dict(os.environ, LF_BENCH_DIR=str(workdir), LF_BENCH_SAM_DIGEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if backend:
        env["LF_BENCH_BACKEND"] = backend
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--genome-mbp", "5", "--reads", "600", "--read-len", "6000", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + list(args)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def bench_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("bench")


@pytest.fixture(scope="module")
def one_rank_line(bench_dir):
    return _bench(bench_dir, "--gpus", "1")


def test_bench_one_rank_reports_both_boundaries(one_rank_line):
    j = one_rank_line
    assert j["n_gpus"] == 1 and j["scaling"] == "weak" and j["value"] > 0
    # N = 1: `value` is the host boundary; the HBM-resident rate is reported next to it
    assert j["value_host_boundary"] == j["value"] and j["value_hbm_resident"] > 0 and "host memory" in j["config"]["io"]
    assert j["roofline"]["kernel"] in j["roofline"]["by_kernel"] and j["roofline"]["by_kernel"][j["roofline"]["kernel"]]["single_kernel"]
    assert j["roofline"]["frac"] > 0 and j["roofline"]["by_kernel"]["lf_ksw_r4_kernel"]["algorithmic_GB_per_step"] >= 0


def test_bench_strong_scaling_two_ranks_same_records(bench_dir, one_rank_line):
    """BASELINE config C3 in miniature: the SAME read set cut by bases over 2 ranks, scattered from and gathered to rank 0
    through PipelinedExchange (gloo hook: both ranks on GPU 0), must leave rank 0 with the 1-rank records, byte for byte"""
    j = _bench(bench_dir, "--gpus", "2", backend="gloo")                 # N > 1 defaults to strong scaling (config C3)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value_weak_hbm_resident"] > 0
    assert j["exchange"]["status"] == "ok" and j["exchange"]["GB_in_per_step"] > 0
    assert j["sam_md5"] == one_rank_line["sam_md5"] and j["sam_bytes"] == one_rank_line["sam_bytes"]
    assert j["value"] > 0 and j["value_hbm_resident"] > 0


def test_bench_exchange_keeps_several_steps_of_a_rank_in_flight(bench_dir, one_rank_line):
    """the strong-scaling default: every rank maps several of its steps concurrently inside the exchange loop (lordfast_amd/dist.py:
    run_pipeline, a ring of depth + 1 buffers) -- four for shards of at most 32 k reads like this one, two for larger ones; records identical to the
    1-rank run; explicit --inflight 2 and 3 too"""
    j = _bench(bench_dir, "--gpus", "2", "--steps", "4", backend="gloo")
    assert j["steps_in_flight"] == 4 and j["exchange"]["steps_in_flight_per_rank"] == 4 and j["exchange"]["status"] == "ok"
    assert j["sam_md5"] == one_rank_line["sam_md5"] and j["sam_bytes"] == one_rank_line["sam_bytes"]
    j2 = _bench(bench_dir, "--gpus", "2", "--steps", "4", "--inflight", "2", backend="gloo")
    assert j2["steps_in_flight"] == 2 and j2["exchange"]["steps_in_flight_per_rank"] == 2 and j2["sam_md5"] == one_rank_line["sam_md5"]
    j3 = _bench(bench_dir, "--gpus", "2", "--steps", "3", "--inflight", "3", backend="gloo")
    assert j3["exchange"]["steps_in_flight_per_rank"] == 3 and j3["sam_md5"] == one_rank_line["sam_md5"]


def test_bench_weak_scaling_three_ranks(bench_dir, one_rank_line):
    j = _bench(bench_dir, "--gpus", "3", "--scaling", "weak", backend="gloo")
    assert j["n_gpus"] == 3 and j["scaling"] == "weak" and j["config"]["reads_total"] == 1800
    assert j["exchange"]["status"] == "ok"
    assert j["sam_bytes"] > 2.5 * one_rank_line["sam_bytes"]          # three different shards, rank 0's first


def test_bench_rccl_when_two_gpus(bench_dir, one_rank_line):
    """the real transport: backend nccl (RCCL), one rank per GPU, device tensors end to end.  Needs two visible GPUs."""
    import lordfast_amd as la
    if la.device_count() < 2:
        pytest.skip("one GPU on this box: the RCCL path needs two (the gloo tests above cover the same code with host staging)")
    j = _bench(bench_dir, "--gpus", "2", "--scaling", "strong", "--steps", "4")
    assert j["exchange"]["status"] == "ok" and j["exchange"]["transport"] == "nccl" and "cuda" in j["exchange"]["bulk_memory"]
    assert j["exchange"]["steps_in_flight_per_rank"] == 2          # device tensors through PipelinedExchange / run_pipeline over RCCL
    assert j["sam_md5"] == one_rank_line["sam_md5"]


NCCL_RING_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from lordfast_amd import dist as lfd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl")
ctl = dist.new_group(backend="gloo")
dev = torch.device("cuda", rank)
STEPS, DEPTH = 6, 2
def job(step):
    rng = np.random.default_rng(900 + step)
    n = 40 + 7 * step
    names = [f"s{step}_r{i}".encode() for i in range(n)]
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(1, 20000)))) for _ in range(n)]
    return names, seqs
def fake_map(k, shard, out, slot):
    b = shard.blob[:shard.nbytes].cpu().numpy().tobytes()
    txt = b"".join(b[int(shard.name_off[i]):int(shard.name_off[i + 1]) - 1] + b"\t" + hashlib.md5(b[int(shard.seq_off[i]):int(shard.seq_off[i + 1]) - 1]).hexdigest().encode() + b"\n" for i in range(len(shard)))
    out[:len(txt)].copy_(torch.frombuffer(bytearray(txt), dtype=torch.uint8)) if txt else None
    torch.cuda.current_stream().synchronize()
    return len(txt)
px = lfd.PipelinedExchange(dist, torch, dev, ctl, read_cap=4 << 20, sam_cap=1 << 20, ring=DEPTH + 1)
cache = {}
def shards_of(k):
    if k not in cache:
        nm, sq = job(k)
        cache[k] = lfd.make_shards(torch, nm, sq, world, dev)[0]
    return cache[k]
got = {}
lfd.run_pipeline(px, STEPS, DEPTH, shards_of, fake_map, on_gathered=lambda k, t, ln: got.__setitem__(k, bytes(t[:ln].cpu().numpy().tobytes())))
if rank == 0:
    for k in range(STEPS):
        nm, sq = job(k)
        assert got[k] == b"".join(a + b"\t" + hashlib.md5(b).hexdigest().encode() + b"\n" for a, b in zip(nm, sq)), ("step", k)
    print("NCCL_RING_OK", px.bytes_out, px.bytes_in)
dist.barrier()
dist.destroy_process_group()
'''


def test_pipelined_exchange_device_tensors_over_rccl_when_two_gpus(tmp_path):
    """PipelinedExchange / run_pipeline with device tensors through a 2-rank nccl (RCCL) group, two steps in flight: the transport the
    N-GPU bench uses, without the mapper.  Needs two visible GPUs (tests/test_dist.py runs the same loop over gloo on the CPU)."""
    import lordfast_amd as la
    if la.device_count() < 2:
        pytest.skip("one GPU on this box: the RCCL transport needs two")
    script = tmp_path / "n.py"
    script.write_text(NCCL_RING_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29671", str(script), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "NCCL_RING_OK" in r.stdout


def test_bench_inproc_two_replicas(bench_dir, one_rank_line):
    """--mode inproc: one process, lf_map_batch_multi over N index replicas (two replicas on GPU 0 here)"""
    env_key = "LF_BENCH_SHARE_DEVICES"
    os.environ[env_key] = "1"
    try:
        j = _bench(bench_dir, "--gpus", "2", "--mode", "inproc", "--scaling", "strong")
    finally:
        os.environ.pop(env_key, None)
    assert j["n_gpus"] == 2 and j["sam_md5"] == one_rank_line["sam_md5"]
