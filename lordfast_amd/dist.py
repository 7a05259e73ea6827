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


# ---- point-to-point variants: rank `src` talks to every other rank directly (one xGMI link per peer); nothing is
# ---- broadcast to ranks that do not need it and nothing is padded.  Work with "nccl" (device tensors) and "gloo".

def _send_bytes(dist, torch, data, device, dst):
    n = len(data)
    dist.send(torch.tensor([n], dtype=torch.int64, device=device), dst)
    if n:
        t = data if isinstance(data, torch.Tensor) else torch.frombuffer(data if isinstance(data, (bytearray, memoryview)) else bytearray(data), dtype=torch.uint8)
        dist.send(t.to(device), dst)


def _recv_tensor(dist, torch, device, src):
    ln = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(ln, src)
    n = int(ln.item())
    buf = torch.empty(n, dtype=torch.uint8, device=device)
    if n:
        dist.recv(buf, src)
    return buf


def scatter_reads_p2p(dist, torch, names, seqs, device, src: int = 0):
    """rank `src` holds the batch; every rank gets ITS shard only. -> (names, seqs, bounds)"""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == src:
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        bounds = shard_bounds(lens, world)
        for r in range(world):
            if r == src:
                continue
            lo, hi = bounds[r]
            sl = lens[lo:hi]
            nl = np.array([len(x) for x in names[lo:hi]], dtype=np.int64)
            meta = np.concatenate([[hi - lo, lo], [b for ab in bounds for b in ab], sl, nl]).astype(np.int64).tobytes()
            _send_bytes(dist, torch, meta, device, r)
            _send_bytes(dist, torch, b"".join(seqs[lo:hi]) + b"".join(names[lo:hi]), device, r)
        lo, hi = bounds[src]
        return names[lo:hi], seqs[lo:hi], bounds
    meta = np.frombuffer(_recv_tensor(dist, torch, device, src).cpu().numpy().tobytes(), dtype=np.int64)
    payload = _recv_tensor(dist, torch, device, src).cpu().numpy().tobytes()
    n = int(meta[0])
    bounds = [(int(meta[2 + 2 * r]), int(meta[3 + 2 * r])) for r in range(world)]
    sl = meta[2 + 2 * world:2 + 2 * world + n]
    nl = meta[2 + 2 * world + n:2 + 2 * world + 2 * n]
    so = np.concatenate([[0], np.cumsum(sl)])
    no = np.concatenate([[0], np.cumsum(nl)]) + int(so[-1])
    my_seqs = [payload[int(so[i]):int(so[i + 1])] for i in range(n)]
    my_names = [payload[int(no[i]):int(no[i + 1])] for i in range(n)]
    return my_names, my_seqs, bounds


def gather_sam_p2p(dist, torch, my_buf, my_len: int, device, dst: int = 0):
    """Concatenate the per-rank SAM texts on rank `dst` in rank order.  `my_buf` is a uint8 CPU tensor (ideally pinned);
    on `dst` it must be large enough for ALL ranks and already hold dst's own text at offset 0 (dst must be rank 0 so
    that its shard comes first).  Returns the total length on dst, None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank != dst:
        _send_bytes(dist, torch, my_buf[:my_len], device, dst)
        return None
    assert dst == 0
    off = my_len
    for r in range(world):
        if r == dst:
            continue
        t = _recv_tensor(dist, torch, device, r)
        n = t.numel()
        if off + n > my_buf.numel():
            raise RuntimeError("gather_sam_p2p: destination buffer too small")
        my_buf[off:off + n].copy_(t, non_blocking=True)
        off += n
    if device.type != "cpu":
        torch.cuda.synchronize()
    return off


# ---- packed batches: what travels is two contiguous byte ranges + two offset arrays per shard; no per-read Python object
# ---- is created on either side (a 100 k-read batch is 1.5 GB: joining / slicing `bytes` costs more than the transfer)

class PackedReads:
    """A read batch as ONE byte blob: the sequences, each followed by NUL, then the names, each followed by NUL -- the
    library takes arrays of C strings, so pointers into the blob are all it needs (the reference keeps a chunk the same
    way: one block per read, `[len][seq\0][qual\0][name\0]`, src/Reads.cpp:84-90)."""

    def __init__(self, blob, seq_off, name_off):
        self.blob = blob                      # uint8 numpy array (any memory: pinned, shared, ...)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)       # n + 1: start of sequence i (NUL before the next)
        self.name_off = np.ascontiguousarray(name_off, dtype=np.int64)     # n + 1, absolute offsets into blob

    def __len__(self):
        return len(self.seq_off) - 1

    @property
    def seq_lens(self):
        return (self.seq_off[1:] - self.seq_off[:-1] - 1).astype(np.uint32)

    def arrays(self):
        """-> (names char**, seqs char**, uint32 lengths) for lf_map_batch_into_lens; valid while self.blob lives"""
        import ctypes as C
        base = self.blob.ctypes.data
        self._np = (self.name_off[:-1] + base).astype(np.uint64)
        self._sp = (self.seq_off[:-1] + base).astype(np.uint64)
        return (C.cast(self._np.ctypes.data, C.POINTER(C.c_char_p)), C.cast(self._sp.ctypes.data, C.POINTER(C.c_char_p)), self.seq_lens)

    def tolists(self):
        b = self.blob.tobytes()
        n = len(self)
        return ([b[int(self.name_off[i]):int(self.name_off[i + 1]) - 1] for i in range(n)],
                [b[int(self.seq_off[i]):int(self.seq_off[i + 1]) - 1] for i in range(n)])


def pack_reads(names, seqs) -> PackedReads:
    n = len(seqs)
    sl = np.array([len(x) + 1 for x in seqs], dtype=np.int64)
    nl = np.array([len(x) + 1 for x in names], dtype=np.int64)
    seq_off = np.concatenate([[0], np.cumsum(sl)]).astype(np.int64)
    name_off = (np.concatenate([[0], np.cumsum(nl)]) + int(seq_off[-1])).astype(np.int64)
    blob = np.frombuffer(b"\0".join(seqs) + b"\0" + b"\0".join(names) + b"\0", dtype=np.uint8) if n else np.zeros(0, np.uint8)
    return PackedReads(blob, seq_off, name_off)


def scatter_packed_p2p(dist, torch, packed, device, src: int = 0):
    """rank `src` holds the packed batch; every rank gets ITS shard (balanced by bases) as a PackedReads.
    -> (PackedReads, bounds).  Per peer: one small meta message + the two byte ranges of its shard."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == src:
        lens = packed.seq_lens.astype(np.int64)
        bounds = shard_bounds(lens, world)
        blob_t = torch.from_numpy(packed.blob)
        mine = None
        for r in range(world):
            lo, hi = bounds[r]
            s0, s1 = int(packed.seq_off[lo]), int(packed.seq_off[hi])
            n0, n1 = int(packed.name_off[lo]), int(packed.name_off[hi])
            if r == src:
                mine = PackedReads(packed.blob, packed.seq_off[lo:hi + 1], packed.name_off[lo:hi + 1])
                continue
            meta = np.concatenate([[hi - lo], [b for ab in bounds for b in ab], packed.seq_off[lo:hi + 1] - s0,
                                   packed.name_off[lo:hi + 1] - n0 + (s1 - s0)]).astype(np.int64)
            _send_bytes(dist, torch, torch.from_numpy(meta.view(np.uint8)), device, r)
            _send_bytes(dist, torch, blob_t[s0:s1], device, r)
            _send_bytes(dist, torch, blob_t[n0:n1], device, r)
        return mine, bounds
    meta = _recv_tensor(dist, torch, device, src).cpu().numpy().view(np.int64)
    n = int(meta[0])
    bounds = [(int(meta[1 + 2 * r]), int(meta[2 + 2 * r])) for r in range(world)]
    seq_off = meta[1 + 2 * world:2 + 2 * world + n]
    name_off = meta[2 + 2 * world + n:3 + 2 * world + 2 * n]
    a = _recv_tensor(dist, torch, device, src)
    b = _recv_tensor(dist, torch, device, src)
    blob = torch.empty(a.numel() + b.numel(), dtype=torch.uint8)
    blob[:a.numel()].copy_(a)
    blob[a.numel():].copy_(b)
    return PackedReads(blob.numpy(), seq_off, name_off), bounds


# ---- device-resident, pipelined exchange (what bench.py --gpus N uses) ---------------------------------------------------
# Rank 0 owns the read batch of every step and ends up owning its SAM records; both live in HBM.  A step's bulk data moves
# with ONE grouped point-to-point call per rank (every peer's link carries its own shard at the same time, both directions)
# that is posted BEFORE the step is mapped:  reads of step k + 1 travel out and SAM text of step k - 1 travels back while
# step k is on the GPUs.  Only the first scatter and the last gather are exposed.  Metadata (offsets, lengths: a few hundred
# kB) goes over a host-side control group (gloo), so no device synchronisation sits between the steps.

import threading as _threading
_NA_LOCK = _threading.Lock()


class Shard:
    """one rank's share of a step: ONE byte blob `[seq\\0]* [name\\0]*` on the bulk device + host offset arrays (n + 1 each,
    absolute in the blob; sequence i is seq_off[i+1] - seq_off[i] - 1 long)."""

    def __init__(self, blob, seq_off, name_off, nbytes=None):
        self.blob = blob                                   # uint8 tensor (cuda under nccl; cpu under gloo)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        self.name_off = np.ascontiguousarray(name_off, dtype=np.int64)
        self.nbytes = int(nbytes if nbytes is not None else (self.name_off[-1] if len(self.name_off) else 0))

    def __len__(self):
        return len(self.seq_off) - 1

    @property
    def seq_lens(self):
        return (self.seq_off[1:] - self.seq_off[:-1] - 1).astype(np.uint32)

    def bases(self):
        return int(self.seq_lens.astype(np.int64).sum())

    def name_array(self, torch):
        """ctypes char*[n] into a HOST copy of the names part of the blob (kept alive by self)"""
        import ctypes as C
        with _NA_LOCK:                                           # a shard may be mapped by several threads at once (steps in flight): built once per shard content
            if getattr(self, "_na_for", None) is not self.blob:
                n0, n1 = int(self.name_off[0]), int(self.name_off[-1])
                names_host = self.blob[n0:n1].cpu().numpy() if n1 > n0 else np.zeros(1, np.uint8)
                ptrs = (self.name_off[:-1] - n0 + names_host.ctypes.data).astype(np.uint64)
                self._names_host, self._np, self._na_for = names_host, ptrs, self.blob
            return C.cast(self._np.ctypes.data, C.POINTER(C.c_char_p))


def make_shards(torch, names, seqs, world, device, by_bases=True):
    """cut one read set into `world` contiguous shards (balanced by bases) and pack each as a Shard resident on `device`.
    -> (shards, bounds)"""
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    bounds = shard_bounds(lens, world)
    out = []
    for lo, hi in bounds:
        pk = pack_reads(names[lo:hi], seqs[lo:hi])
        t = torch.from_numpy(pk.blob.copy()) if len(pk.blob) else torch.zeros(0, dtype=torch.uint8)
        out.append(Shard(t.to(device), pk.seq_off, pk.name_off))
    return out, bounds


class PipelinedExchange:
    """ring = number of steps whose buffers exist at once.  2: the one-step-ahead / one-step-behind pipeline (post / complete around
    every mapping).  D + 1: D steps of a rank are mapped concurrently (bench.py --inflight D; a rank's shard under strong scaling
    is small, and two or three small steps in flight keep the GPU as busy as one large one): tick t waits for the mapping of step
    t - D, completes tick t - 1's transfers, posts {records of step t - D back, reads of step t + 1 out} as ONE grouped
    point-to-point call and hands step t to a mapper thread -- every rank issues the same groups in the same order."""

    def __init__(self, dist, torch, bulk_device, ctl_group, read_cap: int, sam_cap: int, pin=True, ring: int = 2, gather_cap: int | None = None):
        self.dist, self.torch, self.dev, self.ctl = dist, torch, bulk_device, ctl_group
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.ring = max(2, int(ring))
        kw = dict(dtype=torch.uint8, device=bulk_device)
        if bulk_device.type == "cpu" and pin and torch.cuda.is_available():
            kw["pin_memory"] = True
        # rank 0: sam[b] holds the whole step's records (own first, then the peers' in rank order); peers: their own text
        cap0 = int(gather_cap) if gather_cap else sam_cap * self.world
        self.sam = [torch.empty(cap0 if self.rank == 0 else sam_cap, **kw) for _ in range(self.ring)]
        self.rx = [torch.empty(read_cap, **kw) if self.rank else None for _ in range(self.ring)]
        self.rx_shard = [None] * self.ring
        self.works = []
        self.gather_lens = [None] * self.ring     # rank 0: per ring slot, [own, peer1, ...] lengths
        self.bytes_out = self.bytes_in = 0

    # -- control plane (host tensors over gloo)
    def _ctl_send(self, arr, dst):
        self.dist.send(self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)), dst, group=self.ctl)

    def _ctl_recv(self, n, src):
        t = self.torch.empty(n, dtype=self.torch.int64)
        self.dist.recv(t, src, group=self.ctl)
        return t.numpy()

    def post_steps(self, scatter_step=None, shards=None, gather_step=None, own_len=None):
        """ONE grouped point-to-point call: the reads of step `scatter_step` travel out (rank 0: `shards` = that step's N shards)
        and the records of step `gather_step` come back (`own_len` = length of this rank's text of that step, which lies in
        sam[gather_step % ring]).  Either may be None.  Completes in complete()."""
        dist = self.dist
        ops = []
        if self.rank == 0:
            if scatter_step is not None:
                for r in range(1, self.world):
                    sh = shards[r]
                    self._ctl_send([len(sh), sh.nbytes], r)
                    self._ctl_send(np.concatenate([sh.seq_off, sh.name_off]), r)
                    if sh.nbytes:
                        ops.append(dist.P2POp(dist.isend, sh.blob[:sh.nbytes], r))
                        self.bytes_out += sh.nbytes
            if gather_step is not None:
                pb = gather_step % self.ring
                lens = [int(own_len)] + [int(self._ctl_recv(1, r)[0]) for r in range(1, self.world)]
                self.gather_lens[pb] = lens
                off = lens[0]
                need = sum(lens)
                if need > self.sam[pb].numel():
                    # grow instead of raising: the peers have already posted their sends -- a rank that leaves the loop here would leave them
                    # blocked.  Rank 0's own records (complete: own_len is known) move to the new buffer, like the receive buffers grow.
                    bigger = self.torch.empty(int(need * 1.25) + (1 << 20), dtype=self.sam[pb].dtype, device=self.sam[pb].device)
                    bigger[:lens[0]].copy_(self.sam[pb][:lens[0]])
                    self.sam[pb] = bigger
                for r in range(1, self.world):
                    if lens[r]:
                        ops.append(dist.P2POp(dist.irecv, self.sam[pb][off:off + lens[r]], r))
                        self.bytes_in += lens[r]
                    off += lens[r]
        else:
            if scatter_step is not None:
                nb = scatter_step % self.ring
                n, nbytes = (int(x) for x in self._ctl_recv(2, 0))
                meta = self._ctl_recv(2 * (n + 1), 0)
                if nbytes > self.rx[nb].numel():
                    raise RuntimeError("PipelinedExchange: receive buffer too small")
                self.rx_shard[nb] = Shard(self.rx[nb], meta[:n + 1], meta[n + 1:], nbytes)
                if nbytes:
                    ops.append(dist.P2POp(dist.irecv, self.rx[nb][:nbytes], 0))
            if gather_step is not None:
                pb = gather_step % self.ring
                self._ctl_send([int(own_len)], 0)
                if own_len:
                    ops.append(dist.P2POp(dist.isend, self.sam[pb][:int(own_len)], 0))
        self.works = dist.batch_isend_irecv(ops) if ops else []

    def post(self, k: int, next_shards=None, prev_own_len=None):
        """Call before mapping step k (ring 2).  rank 0: next_shards = the N shards of step k + 1 (or None after the last step),
        prev_own_len = length of its own SAM text of step k - 1 (None for k == 0).  Peers: next_shards is ignored (pass True
        when a step k + 1 exists), prev_own_len = length of their SAM text of step k - 1 (None for k == 0).
        Everything posted here completes in complete()."""
        have_next = next_shards is not None and next_shards is not False
        self.post_steps(k + 1 if have_next else None, next_shards if have_next else None,
                        k - 1 if prev_own_len is not None else None, prev_own_len)

    def complete(self):
        for w in self.works:
            w.wait()
        self.works = []
        if self.dev.type == "cuda":
            self.torch.cuda.current_stream().synchronize()      # the waits above only order the current stream behind RCCL's

    def gathered(self, k: int):
        """rank 0, after the transfers of step k's records completed: (tensor, total length) of step k's records, input order"""
        lens = self.gather_lens[k % self.ring]
        return self.sam[k % self.ring], sum(lens)


def run_pipeline(px, n_steps: int, depth: int, shards_of, map_step, pool=None, on_gathered=None):
    """The exchange loop with `depth` steps of every rank in flight (ring = depth + 1).  shards_of(k) -> rank 0: the N shards of
    step k (peers: ignored); map_step(k, shard, out_tensor, slot) -> length of the text written to out_tensor (called on a pool
    thread; slot = k % depth: per-thread staging buffers).  Returns the lengths (per step) of this rank's own texts.
    on_gathered(k, tensor, length): rank 0, called once step k's records are complete in input order (the buffer is reused by step
    k + depth + 1 right afterwards).
    Order of events per tick t: wait map(t - depth) | complete tick t - 1's transfers (reads of step t have arrived, the buffer
    of step t - depth - 1's records is free) | post {records of t - depth, reads of t + 1} | start map(t)."""
    from concurrent.futures import ThreadPoolExecutor
    D = max(1, int(depth))
    if px.ring < D + 1:
        raise ValueError("run_pipeline: ring %d < depth %d + 1" % (px.ring, D))
    own = {}
    ex = pool or ThreadPoolExecutor(D)
    try:
        px.post_steps(0 if n_steps > 0 else None, shards_of(0) if (px.rank == 0 and n_steps > 0) else None)      # reads of step 0: exposed
        futs = {}
        for t in range(n_steps + D):
            g = t - D
            if g >= 0:
                own[g] = futs.pop(g).result()
            px.complete()
            if on_gathered is not None and px.rank == 0 and g - 1 >= 0:
                on_gathered(g - 1, *px.gathered(g - 1))
            s = t + 1 if t + 1 < n_steps else None
            px.post_steps(s, shards_of(s) if (s is not None and px.rank == 0) else None, g if g >= 0 else None, own.get(g) if g >= 0 else None)
            if t < n_steps:
                shard = shards_of(t)[0] if px.rank == 0 else px.rx_shard[t % px.ring]
                futs[t] = ex.submit(map_step, t, shard, px.sam[t % px.ring], t % D)
        px.complete()
        if on_gathered is not None and px.rank == 0 and n_steps > 0:
            on_gathered(n_steps - 1, *px.gathered(n_steps - 1))
    finally:
        if pool is None:
            ex.shutdown(wait=True)
    return own
