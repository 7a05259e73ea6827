"""Seeded synthetic genomes and PacBio/ONT-profile reads.

Only numpy; deterministic for a given seed.  Used by bench.py, the tests and the golden-vector
generator.  Reads are uppercase ACGT (edlib compares raw bytes: SURVEY App. B #11), drawn fully
inside one contig, 50/50 strand, with independent per-base errors split sub:ins:del.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def make_genome(total_bp: int, n_contigs: int = 3, seed: int = 11, repeat_frac: float = 0.10,
                n_families: int = 0) -> list[tuple[str, np.ndarray]]:
    """Uniform ACGT contigs; optionally `repeat_frac` of the sequence is overwritten with copies of
    `n_families` repeat families (300 bp - 6 kbp, 5-20 % divergence) to emulate repeat content."""
    rng = np.random.default_rng(seed)
    # contig lengths: geometric-ish split so contigs differ in size
    w = rng.uniform(0.5, 1.5, size=n_contigs)
    lens = np.maximum((w / w.sum() * total_bp).astype(np.int64), 2000)
    contigs = []
    for i, ln in enumerate(lens):
        contigs.append((f"chr{i + 1}", _ACGT[rng.integers(0, 4, size=int(ln), dtype=np.uint8)]))
    if n_families > 0 and repeat_frac > 0:
        budget = int(total_bp * repeat_frac)
        fams = []
        for _ in range(n_families):
            fl = int(rng.integers(300, 6001))
            fams.append(_ACGT[rng.integers(0, 4, size=fl, dtype=np.uint8)])
        used = 0
        while used < budget:
            f = fams[int(rng.integers(0, n_families))]
            ci = int(rng.integers(0, n_contigs))
            name, s = contigs[ci]
            if len(s) <= len(f) + 1:
                continue
            div = rng.uniform(0.05, 0.20)
            copy = f.copy()
            mut = rng.random(len(copy)) < div
            copy[mut] = _ACGT[rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)]
            if rng.random() < 0.5:
                copy = revcomp(copy)
            p = int(rng.integers(0, len(s) - len(copy)))
            s[p:p + len(copy)] = copy
            used += len(copy)
    return contigs


def write_fasta(path: str, contigs, width: int = 80) -> None:
    with open(path, "wb") as fh:
        for name, s in contigs:
            fh.write(b">" + name.encode() + b"\n")
            n = len(s)
            full = (n // width) * width
            if full:
                body = np.empty((n // width, width + 1), dtype=np.uint8)
                body[:, :width] = s[:full].reshape(-1, width)
                body[:, width] = ord("\n")
                fh.write(body.tobytes())
            if full < n:
                fh.write(s[full:].tobytes() + b"\n")


def mutate(seq: np.ndarray, err: float, rng, mix=(0.15, 0.50, 0.35)) -> np.ndarray:
    """Independent per-base errors; mix = (substitution, insertion, deletion) shares."""
    n = len(seq)
    u = rng.random(n)
    p_sub, p_ins, p_del = (err * m for m in mix)
    is_sub = u < p_sub
    is_ins = (u >= p_sub) & (u < p_sub + p_ins)
    is_del = (u >= p_sub + p_ins) & (u < p_sub + p_ins + p_del)
    out = seq.copy()
    # substitution: a different base
    k = int(is_sub.sum())
    if k:
        cur = np.searchsorted(_ACGT, out[is_sub])
        out[is_sub] = _ACGT[(cur + rng.integers(1, 4, size=k)) % 4]
    keep = ~is_del
    counts = keep.astype(np.int64) + is_ins.astype(np.int64)  # inserted base goes before the kept base
    total = int(counts.sum())
    res = np.empty(total, dtype=np.uint8)
    ends = np.cumsum(counts)
    # place kept bases at the last slot of their group
    res[ends[keep] - 1] = out[keep]
    ins_slots = ends[is_ins] - counts[is_ins]
    res[ins_slots] = _ACGT[rng.integers(0, 4, size=int(is_ins.sum()), dtype=np.uint8)]
    return res


def make_reads(contigs, n_reads: int, mean_len: int, err: float, seed: int = 2024,
               mix=(0.15, 0.50, 0.35), min_len: int = 1000, sigma: float = 0.35):
    """Returns list of (name, seq_bytes). Name carries the true origin: r<i>_<contig>_<pos>_<strand>."""
    rng = np.random.default_rng(seed)
    clens = np.array([len(s) for _, s in contigs], dtype=np.float64)
    pc = clens / clens.sum()
    mu = np.log(mean_len) - 0.5 * sigma * sigma
    reads = []
    for i in range(n_reads):
        while True:
            ci = int(rng.choice(len(contigs), p=pc))
            ln = int(max(min_len, rng.lognormal(mu, sigma)))
            name, s = contigs[ci]
            if ln < len(s):
                break
        p = int(rng.integers(0, len(s) - ln))
        frag = s[p:p + ln]
        strand = "+"
        if rng.random() < 0.5:
            frag = revcomp(frag)
            strand = "-"
        r = mutate(frag, err, rng, mix)
        reads.append((f"r{i}_{name}_{p}_{strand}", r.tobytes()))
    return reads


def write_reads_fasta(path: str, reads) -> None:
    with open(path, "wb") as fh:
        for name, s in reads:
            fh.write(b">" + name.encode() + b"\n" + s + b"\n")
