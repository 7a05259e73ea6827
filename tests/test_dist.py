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
