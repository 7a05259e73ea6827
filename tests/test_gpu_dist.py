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
