"""N > 1 path on CPU: read sharding + scatter/gather over torch.distributed (gloo, world_size 2)."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from lordfast_amd import dist as lfd

WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from lordfast_amd import dist as lfd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cpu")
rng = np.random.default_rng(7)
names = [f"r{i}".encode() for i in range(37)]
seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(1, 4000)))) for _ in range(37)]
my_names, my_seqs, bounds = lfd.scatter_reads(dist, torch, names if rank == 0 else None, seqs if rank == 0 else None, dev)
lo, hi = bounds[rank]
assert my_names == names[lo:hi] and my_seqs == seqs[lo:hi]
fake_sam = b"".join(n + b"\t" + hashlib.md5(s).hexdigest().encode() + b"\n" for n, s in zip(my_names, my_seqs))
out = lfd.gather_sam(dist, torch, fake_sam, dev)
# point-to-point variants (what bench.py uses)
n2, s2, b2 = lfd.scatter_reads_p2p(dist, torch, names if rank == 0 else None, seqs if rank == 0 else None, dev)
assert b2 == bounds and n2 == my_names and s2 == my_seqs
buf = torch.zeros(1 << 16, dtype=torch.uint8)
buf[:len(fake_sam)] = torch.frombuffer(bytearray(fake_sam), dtype=torch.uint8)
tot = lfd.gather_sam_p2p(dist, torch, buf, len(fake_sam), dev)
# packed variant: byte ranges + offsets only, pointer arrays straight into the received blob
pk = lfd.pack_reads(names, seqs) if rank == 0 else None
shard, b3 = lfd.scatter_packed_p2p(dist, torch, pk, dev)
assert b3 == bounds and len(shard) == hi - lo
assert shard.tolists() == (my_names, my_seqs)
import ctypes as C
na, sa, sl = shard.arrays()
assert [na[i] for i in range(len(shard))] == my_names and [sa[i] for i in range(len(shard))] == my_seqs
assert list(sl) == [len(x) for x in my_seqs]
if rank == 0:
    exp = b"".join(n + b"\t" + hashlib.md5(s).hexdigest().encode() + b"\n" for n, s in zip(names, seqs))
    assert out == exp
    assert bytes(buf[:tot].numpy().tobytes()) == exp
    print("DIST_OK", bounds)
dist.destroy_process_group()
'''


def test_shard_bounds_balanced_by_bases():
    rng = np.random.default_rng(1)
    lens = rng.integers(1000, 60000, size=1000)
    for world in (1, 2, 4, 8):
        b = lfd.shard_bounds(lens, world)
        assert b[0][0] == 0 and b[-1][1] == len(lens)
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        loads = [int(lens[lo:hi].sum()) for lo, hi in b]
        assert max(loads) - min(loads) <= 2 * lens.max()
    assert lfd.shard_bounds([5], 4) == [(0, 0), (0, 0), (0, 1), (1, 1)] or sum(hi - lo for lo, hi in lfd.shard_bounds([5], 4)) == 1
    assert lfd.shard_bounds([], 2) == [(0, 0), (0, 0)]


def test_scatter_gather_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script), ROOT],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_OK" in r.stdout


PIPE_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from lordfast_amd import dist as lfd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cpu")
STEPS = 4
def job(step):
    rng = np.random.default_rng(100 + step)
    n = 23 + 5 * step
    names = [f"s{step}_r{i}".encode() for i in range(n)]
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(1, 3000)))) for _ in range(n)]
    return names, seqs
def fake_map(shard, out):
    """stands in for lf_map_batch_dev: one line per read, from the packed blob only"""
    b = shard.blob[:shard.nbytes].numpy().tobytes()
    lines = []
    for i in range(len(shard)):
        nm = b[int(shard.name_off[i]):int(shard.name_off[i + 1]) - 1]
        sq = b[int(shard.seq_off[i]):int(shard.seq_off[i + 1]) - 1]
        assert len(sq) == int(shard.seq_lens[i])
        lines.append(nm + b"\t" + hashlib.md5(sq).hexdigest().encode() + b"\n")
    txt = b"".join(lines)
    out[:len(txt)] = torch.frombuffer(bytearray(txt), dtype=torch.uint8) if txt else out[:0]
    return len(txt)
px = lfd.PipelinedExchange(dist, torch, dev, dist.group.WORLD, read_cap=1 << 20, sam_cap=1 << 20, pin=False)
shards = {}
if rank == 0:
    for k in range(STEPS):
        nm, sq = job(k)
        shards[k] = lfd.make_shards(torch, nm, sq, world, dev)[0]
# step -1: scatter of step 0 (exposed); then: post(k) -> map(k) -> complete()
px.post(-1, next_shards=shards[0] if rank == 0 else True); px.complete()
prev = None
got = {}
for k in range(STEPS):
    more = k + 1 < STEPS
    px.post(k, next_shards=(shards[k + 1] if rank == 0 else True) if more else None, prev_own_len=prev)
    sh = shards[k][0] if rank == 0 else px.rx_shard[k & 1]
    prev = fake_map(sh, px.sam[k & 1])
    px.complete()
    if rank == 0 and k >= 1:
        t, ln = px.gathered(k - 1)
        got[k - 1] = bytes(t[:ln].numpy().tobytes())
px.post(STEPS, next_shards=None, prev_own_len=prev); px.complete()
if rank == 0:
    t, ln = px.gathered(STEPS - 1)
    got[STEPS - 1] = bytes(t[:ln].numpy().tobytes())
    for k in range(STEPS):
        nm, sq = job(k)
        exp = b"".join(a + b"\t" + hashlib.md5(b).hexdigest().encode() + b"\n" for a, b in zip(nm, sq))
        assert got[k] == exp, ("step", k)
    assert px.bytes_out > 0 and px.bytes_in > 0
    print("PIPE_OK", STEPS, px.bytes_out, px.bytes_in)
dist.barrier()
dist.destroy_process_group()
'''


def test_pipelined_exchange_gloo(tmp_path):
    """rank 0 owns every step's reads and ends with every step's records, in input order, while the transfers of step
    k + 1 / k - 1 are in flight around the mapping of step k (world sizes 2 and 3)"""
    script = tmp_path / "p.py"
    script.write_text(PIPE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for world, port in ((2, 29613), (3, 29614)):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert "PIPE_OK" in r.stdout


RING_WORKER = r'''
import os, sys, hashlib, time, threading
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from lordfast_amd import dist as lfd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cpu")
STEPS, DEPTH = int(sys.argv[2]), int(sys.argv[3])
def job(step):
    rng = np.random.default_rng(500 + step)
    n = 19 + 3 * step
    names = [f"s{step}_r{i}".encode() for i in range(n)]
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(1, 2500)))) for _ in range(n)]
    return names, seqs
running, peak, lock = [0], [0], threading.Lock()
def fake_map(k, shard, out, slot):
    with lock:
        running[0] += 1; peak[0] = max(peak[0], running[0])
    assert slot == k % DEPTH
    time.sleep(0.02 * (1 + (k + rank) % 3))           # steps finish out of order across ranks
    b = shard.blob[:shard.nbytes].numpy().tobytes()
    lines = []
    for i in range(len(shard)):
        nm = b[int(shard.name_off[i]):int(shard.name_off[i + 1]) - 1]
        sq = b[int(shard.seq_off[i]):int(shard.seq_off[i + 1]) - 1]
        lines.append(nm + b"\t" + hashlib.md5(sq).hexdigest().encode() + b"\n")
    txt = b"".join(lines)
    out[:len(txt)] = torch.frombuffer(bytearray(txt), dtype=torch.uint8) if txt else out[:0]
    with lock:
        running[0] -= 1
    return len(txt)
px = lfd.PipelinedExchange(dist, torch, dev, dist.group.WORLD, read_cap=1 << 20, sam_cap=1 << 20, pin=False, ring=DEPTH + 1)
cache = {}
def shards_of(k):
    if k not in cache:
        nm, sq = job(k)
        cache[k] = lfd.make_shards(torch, nm, sq, world, dev)[0]
    return cache[k]
got = {}
def on_gathered(k, t, ln):
    got[k] = bytes(t[:ln].numpy().tobytes())
own = lfd.run_pipeline(px, STEPS, DEPTH, shards_of, fake_map, on_gathered=on_gathered)
assert sorted(own) == list(range(STEPS))
if rank == 0:
    for k in range(STEPS):
        nm, sq = job(k)
        exp = b"".join(a + b"\t" + hashlib.md5(b).hexdigest().encode() + b"\n" for a, b in zip(nm, sq))
        assert got[k] == exp, ("step", k)
    assert STEPS < 2 or DEPTH < 2 or peak[0] >= 2, peak
    print("RING_OK", STEPS, DEPTH, peak[0], px.bytes_out, px.bytes_in)
dist.barrier()
dist.destroy_process_group()
'''


def test_pipelined_exchange_with_several_steps_in_flight_gloo(tmp_path):
    """lfd.run_pipeline: `depth` steps of every rank are mapped concurrently (ring = depth + 1 buffers) while the reads of the next
    step and the records of the step that just left the ring travel; rank 0 sees every step's records complete and in input
    order (world 2 / depth 2, world 3 / depth 3, and more depth than steps)"""
    script = tmp_path / "r.py"
    script.write_text(RING_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for world, steps, depth, port in ((2, 7, 2, 29615), (3, 5, 3, 29616), (2, 1, 2, 29617)):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(steps), str(depth)],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert "RING_OK" in r.stdout


def test_bench_spawns_its_own_ranks_without_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` from a bare shell must start torch.distributed.run children itself (here: checked up to the
    point where a rank finds no GPU and says so -- the parent must not have imported torch or initialised HIP)"""
    env = dict(os.environ, LF_BENCH_BACKEND="gloo", LF_BENCH_DIR=str(tmp_path))
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--genome-mbp", "1",
                        "--reads", "10", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    assert "starting 2 ranks" in r.stderr, r.stderr[-2000:]
    assert "must be launched with torch.distributed.run" not in (r.stdout + r.stderr)
    # on this CPU box the ranks stop at the device check; on a GPU box test_gpu_dist.py runs the same command to the end
    assert r.returncode != 0 or '"metric"' in r.stdout


def test_bench_steps_in_flight_defaults(monkeypatch):
    """bench.py's --inflight default: one step after the other at one GPU and under weak scaling; under strong scaling two steps of a rank in flight,
    four when the rank's shard is at most 32 k reads (N = 4 and 8 of the 100 k-read set); an explicit value wins"""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")

    def inflight(*argv):
        monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
        return bench.parse().inflight

    assert inflight() == 1
    assert inflight("--gpus", "2") == 2            # 50 k reads per rank
    assert inflight("--gpus", "4") == 4            # 25 k
    assert inflight("--gpus", "8") == 4            # 12.5 k
    assert inflight("--gpus", "8", "--reads", "400000") == 2
    assert inflight("--gpus", "8", "--scaling", "weak") == 1
    assert inflight("--gpus", "8", "--inflight", "3") == 3
