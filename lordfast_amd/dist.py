"""Multi-GPU sharding of a read batch: one process per GPU, reads are independent units.

No data-path collective exists in the algorithm; torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in the CPU tests) is used only to scatter the packed read batch from rank 0 and to gather
the SAM records back, so that the 1-GPU and N-GPU outputs are byte-identical and in input order.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(lengths, world: int):
    """Contiguous slices balanced by cumulative bases (not read count). -> list of (lo, hi)."""
    n = len(lengths)
    if world <= 1:
        return [(0, n)]
    cum = np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64))])
    total = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        k = max(cuts[-1], min(n, k))
        cuts.append(k)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def _bcast_bytes(dist, torch, data: bytes | None, device, src: int = 0) -> bytes:
    """broadcast a byte string from src (length first)."""
    rank = dist.get_rank()
    ln = torch.tensor([len(data) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(ln, src)
    n = int(ln.item())
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device) if n else torch.empty(0, dtype=torch.uint8, device=device)
    else:
        buf = torch.empty(n, dtype=torch.uint8, device=device)
    if n:
        dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def scatter_reads(dist, torch, names, seqs, device, src: int = 0):
    """rank `src` holds (names, seqs); every rank returns ITS shard (names, seqs) and the global bounds.
    The packed batch travels as one broadcast over xGMI and each rank slices its part (volume is tiny next
    to the link rate: SURVEY section 5)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == src:
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        nlens = np.array([len(x) for x in names], dtype=np.int64)
        meta = np.concatenate([[len(seqs)], lens, nlens]).astype(np.int64).tobytes()
        payload = b"".join(seqs) + b"".join(names)
    else:
        meta = payload = None
    meta = np.frombuffer(_bcast_bytes(dist, torch, meta, device, src), dtype=np.int64)
    payload = _bcast_bytes(dist, torch, payload, device, src)
    n = int(meta[0])
    lens, nlens = meta[1:1 + n], meta[1 + n:1 + 2 * n]
    bounds = shard_bounds(lens, world)
    lo, hi = bounds[rank]
    so = np.concatenate([[0], np.cumsum(lens)])
    no = np.concatenate([[0], np.cumsum(nlens)]) + int(so[-1])
    my_seqs = [payload[int(so[i]):int(so[i + 1])] for i in range(lo, hi)]
    my_names = [payload[int(no[i]):int(no[i + 1])] for i in range(lo, hi)]
    return my_names, my_seqs, bounds


def gather_sam(dist, torch, sam, device, dst: int = 0) -> bytes | None:
    """concatenate the per-rank SAM blobs on rank dst in rank order (= input order)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    ln = torch.tensor([len(sam)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, ln)
    sizes = [int(x.item()) for x in sizes]
    mx = max(sizes + [1])
    mine = torch.zeros(mx, dtype=torch.uint8, device=device)
    if len(sam):
        src = sam if isinstance(sam, (bytearray, memoryview)) else bytearray(sam)
        mine[:len(sam)] = torch.frombuffer(src, dtype=torch.uint8).to(device)
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(bufs, mine)
    if rank != dst:
        return None
    return b"".join(bytes(b[:s].cpu().numpy().tobytes()) for b, s in zip(bufs, sizes))
