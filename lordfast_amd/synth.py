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
                n_families: int = 0, profile: str = "default") -> list[tuple[str, np.ndarray]]:
    """Uniform ACGT contigs; optionally `repeat_frac` of the sequence is overwritten with copies of
    `n_families` repeat families (300 bp - 6 kbp, 5-20 % divergence) to emulate repeat content.
    profile "grch38like": about half of the sequence is repeats with the copy-number structure of a human genome --
    one 300 bp family with ~10^6 copies per 3 Gbp (Alu-like, 10 % of the sequence, 5-15 % divergence), twenty 1-6 kbp
    families (LINE-like, 17 %, truncated copies), and `n_families` low-copy families for the rest.
    profile "t2tlike": grch38like plus what a telomere-to-telomere assembly adds -- one centromeric satellite array per contig (a
    171 bp monomer family in higher-order repeats of 4 - 16 monomers, monomers 20 - 30 % apart, HOR copies 0.1 - 2 % apart, arrays
    summing to ~6 % of the sequence) and simple-sequence arrays (a 5 bp unit at ~5 % divergence, ~1 %)."""
    rng = np.random.default_rng(seed)
    if profile == "grch38like":
        return _make_genome_grch38like(total_bp, n_contigs, rng, max(n_families, 100))
    if profile == "t2tlike":
        contigs = _make_genome_grch38like(total_bp, n_contigs, rng, max(n_families, 100))
        _add_satellites(contigs, total_bp, rng)
        return contigs
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


def _make_genome_grch38like(total_bp, n_contigs, rng, n_families):
    w = rng.uniform(0.5, 1.5, size=n_contigs)
    lens = np.maximum((w / w.sum() * total_bp).astype(np.int64), 2000)
    contigs = [(f"chr{i + 1}", _ACGT[rng.integers(0, 4, size=int(ln), dtype=np.uint8)]) for i, ln in enumerate(lens)]
    starts = np.concatenate([[0], np.cumsum(lens)])

    def paste(fams, budget, lo_div, hi_div, truncate):
        used = 0
        while used < budget:
            f = fams[int(rng.integers(0, len(fams)))]
            if truncate and len(f) > 600:                      # 5'-truncated copies, like LINE insertions
                f = f[int(rng.integers(0, len(f) - 300)):]
            g = int(rng.integers(0, total_bp))
            ci = int(np.searchsorted(starts, g, side="right") - 1)
            ci = min(ci, n_contigs - 1)
            s = contigs[ci][1]
            if len(s) <= len(f) + 1:
                continue
            copy = f.copy()
            mut = rng.random(len(copy)) < rng.uniform(lo_div, hi_div)
            copy[mut] = _ACGT[rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)]
            if rng.random() < 0.5:
                copy = revcomp(copy)
            p = int(rng.integers(0, len(s) - len(copy)))
            s[p:p + len(copy)] = copy
            used += len(copy)

    rnd = lambda n: _ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]                 # noqa: E731
    paste([rnd(int(rng.integers(300, 6001))) for _ in range(n_families)], int(total_bp * 0.23), 0.05, 0.20, False)
    paste([rnd(int(rng.integers(1000, 6001))) for _ in range(20)], int(total_bp * 0.17), 0.03, 0.15, True)
    paste([rnd(300)], int(total_bp * 0.10), 0.05, 0.15, False)                       # pasted last: the youngest family
    return contigs


def _add_satellites(contigs, total_bp, rng):
    """centromeric alpha-satellite-like arrays and simple-sequence arrays, pasted over the middle of every contig (in place)"""
    cons = _ACGT[rng.integers(0, 4, size=171, dtype=np.uint8)]
    n_contigs = len(contigs)
    for ci, (_, s) in enumerate(contigs):
        arr_len = int(min(len(s) * 0.5, max(20000, total_bp * 0.06 / n_contigs)))
        if len(s) < 4 * 171 or arr_len < 4 * 171:
            continue
        k = int(rng.integers(4, 17))
        monos = []
        for _ in range(k):                                      # the HOR's monomers: 20 - 30 % from the consensus
            mo = cons.copy()
            mut = rng.random(171) < rng.uniform(0.20, 0.30)
            mo[mut] = _ACGT[rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)]
            monos.append(mo)
        hor = np.concatenate(monos)
        reps = arr_len // len(hor) + 1
        arr = np.tile(hor, reps)[:arr_len].copy()
        # HOR copies 0.1 - 2 % apart: divergence drawn per copy
        for a in range(0, arr_len, len(hor)):
            b = min(arr_len, a + len(hor))
            mut = rng.random(b - a) < rng.uniform(0.001, 0.02)
            if mut.any():
                arr[a:b][mut] = _ACGT[rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)]
        p = (len(s) - arr_len) // 2
        s[p:p + arr_len] = arr
        # a simple-sequence array next to it
        ss_len = int(min(len(s) * 0.1, max(2000, total_bp * 0.01 / n_contigs)))
        unit = _ACGT[rng.integers(0, 4, size=5, dtype=np.uint8)]
        ss = np.tile(unit, ss_len // 5 + 1)[:ss_len].copy()
        mut = rng.random(ss_len) < 0.05
        ss[mut] = _ACGT[rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)]
        q = p + arr_len
        if q + ss_len < len(s):
            s[q:q + ss_len] = ss


def read_fasta(path: str, min_contig: int = 2000) -> list[tuple[str, np.ndarray]]:
    """A reference FASTA (plain or gzip) as (name, uint8 array) contigs, upper-cased; contigs shorter than min_contig are dropped."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    contigs, name, parts = [], None, []

    def flush():
        if name is not None and parts:
            a = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
            a[(a >= 97) & (a <= 122)] -= 32
            if len(a) >= min_contig:
                contigs.append((name, a))
    with op(path, "rb") as fh:
        for line in fh:
            if line.startswith(b">"):
                flush()
                name, parts = line[1:].split()[0].decode(), []
            else:
                parts.append(line.strip())
    flush()
    return contigs


def add_duplications(contigs, seg_len: int = 12000, copies: int = 3, div: float = 0.025, seed: int = 5):
    """Copy one `seg_len` segment to `copies-1` other places at `div` divergence (substitutions):
    reads drawn from it have several near-equal candidate windows -> lordFAST's fine mode."""
    rng = np.random.default_rng(seed)
    name, s0 = contigs[0]
    src = int(rng.integers(0, len(s0) - seg_len))
    seg = s0[src:src + seg_len].copy()
    places = [(0, src)]
    for k in range(copies - 1):
        ci = (k + 1) % len(contigs)
        _, s = contigs[ci]
        p = int(rng.integers(0, len(s) - seg_len))
        c = seg.copy()
        mut = rng.random(seg_len) < div
        cur = np.searchsorted(_ACGT, c[mut])
        c[mut] = _ACGT[(cur + rng.integers(1, 4, size=int(mut.sum()))) % 4]
        s[p:p + seg_len] = c
        places.append((ci, p))
    return places


def add_segdups(contigs, n_segments: int, seg_len=(40000, 80000), copies=(2, 4), div=(0.01, 0.03), seed: int = 7):
    """Segmental duplications: `n_segments` source segments of seg_len[0] .. seg_len[1] bp, each present in 2 .. 4 copies
    (the source + pasted copies on random contigs, either strand) at 1 - 3 % divergence (substitutions).  Reads drawn from
    them have several near-equal candidate windows: lordFAST's fine mode (src/LordFAST.cpp:542-562) with -n > 1 aligns every
    one of them.  Copies do not overlap each other (10 kbp occupancy grid).  Returns one list of (contig, pos, len) per
    family; contigs are modified in place."""
    rng = np.random.default_rng(seed)
    G = 10000
    occ = [np.zeros(len(s) // G + 2, dtype=bool) for _, s in contigs]
    clens = np.array([len(s) for _, s in contigs], dtype=np.float64)
    pc = clens / clens.sum()

    def place(ln):
        for _ in range(200):
            ci = int(rng.choice(len(contigs), p=pc))
            s = contigs[ci][1]
            if len(s) <= ln + 2 * G:
                continue
            p = int(rng.integers(0, len(s) - ln))
            a, b = p // G, (p + ln) // G + 1
            if not occ[ci][a:b].any():
                occ[ci][a:b] = True
                return ci, p
        return None

    fams = []
    for _ in range(n_segments):
        ln = int(rng.integers(seg_len[0], seg_len[1] + 1))
        src = place(ln)
        if src is None:
            continue
        seg = contigs[src[0]][1][src[1]:src[1] + ln].copy()
        fam = [(src[0], src[1], ln)]
        for _k in range(int(rng.integers(copies[0], copies[1] + 1)) - 1):
            dst = place(ln)
            if dst is None:
                break
            c = seg.copy()
            mut = rng.random(ln) < rng.uniform(div[0], div[1])
            cur = np.searchsorted(_ACGT, c[mut])
            c[mut] = _ACGT[(cur + rng.integers(1, 4, size=int(mut.sum()))) % 4]
            if rng.random() < 0.5:
                c = revcomp(c)
            contigs[dst[0]][1][dst[1]:dst[1] + ln] = c
            fam.append((dst[0], dst[1], ln))
        fams.append(fam)
    return fams


def special_reads(contigs, dup_places=None, seed: int = 77, err: float = 0.10):
    """Reads engineered to reach the rare branches of the extension code:
    big deletion / insertion (split + supplementary), inversion with indel (inverted middle segment),
    junk head / tail (clip test), chimera, duplicated-segment read (fine mode), too short, N-rich,
    lower case, unmappable."""
    rng = np.random.default_rng(seed)
    out = []

    def frag(ci, p, ln):
        return contigs[ci][1][p:p + ln].copy()

    def rnd(n):
        return _ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]

    def emit(tag, arr, rc=False, e=err):
        if rc:
            arr = revcomp(arr)
        out.append((f"s{len(out)}_{tag}", mutate(arr, e, rng).tobytes()))

    big = max(range(len(contigs)), key=lambda i: len(contigs[i][1]))
    n_big = len(contigs[big][1])
    for rc in (False, True):
        p = int(rng.integers(1000, n_big - 12000))
        f = frag(big, p, 9000)
        emit("del600", np.concatenate([f[:4000], f[4600:]]), rc)                       # 600 bp deletion in read
        emit("ins600", np.concatenate([f[:4000], rnd(600), f[4000:]]), rc)             # 600 bp insertion in read
        emit("ins2500", np.concatenate([f[:3000], rnd(2500), f[3000:]]), rc)           # long junk insertion
        # inversion + deletion sized so that the gap fails the 0.40 identity test forward but its reverse
        # complement passes the 0.60 test -> inverted middle segment (src/LordFAST.cpp:2033-2077)
        emit("inv", np.concatenate([f[:3000], revcomp(f[3000:4500]), f[4850:]]), rc, 0.06)
        emit("inv2", np.concatenate([f[:3000], revcomp(f[3000:5000]), rnd(900), f[5000:]]), rc, 0.05)
        emit("inv3", np.concatenate([f[:3000], revcomp(f[3000:4500]), f[4700:]]), rc)  # aligned through
        emit("junkhead", np.concatenate([rnd(1500), f[:6000]]), rc)
        emit("junktail", np.concatenate([f[:6000], rnd(1500)]), rc)
        emit("junkboth", np.concatenate([rnd(800), f[:5000], rnd(900)]), rc, 0.15)
        q = int(rng.integers(1000, len(contigs[0][1]) - 6000))
        emit("chimera", np.concatenate([f[:4000], frag(0, q, 4000)]), rc)
    if dup_places:
        for (ci, p) in dup_places:
            for rc in (False, True):
                emit("dup", frag(ci, p + 1500, 8000), rc, 0.12)
    emit("short", frag(big, 500, 700))
    emit("unmappable", rnd(4000), e=0.0)
    nread = frag(big, 2000, 5000)
    nread[rng.integers(0, 5000, size=150)] = ord("N")
    emit("withN", nread)
    low = mutate(frag(big, 8000, 4000), err, rng)
    out.append((f"s{len(out)}_lower", low.tobytes().lower()))
    emit("edge_start", frag(big, 0, 3000))
    emit("edge_end", frag(big, n_big - 3000, 3000))
    emit("edge_end_rc", frag(big, n_big - 3500, 3500), True)
    last = len(contigs) - 1
    emit("genome_end", frag(last, len(contigs[last][1]) - 3000, 3000))
    return out


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
               mix=(0.15, 0.50, 0.35), min_len: int = 1000, sigma: float = 0.35, segdups=None, dup_frac: float = 0.0,
               acgt_only: bool = False):
    """Returns list of (name, seq_bytes). Name carries the true origin: r<i>_<contig>_<pos>_<strand>.
    segdups (add_segdups) + dup_frac: that fraction of the reads lies inside a copy of a duplicated segment (name ends in
    `_dup`); with dup_frac == 0 the random stream, hence the read set, is what it always was.  acgt_only: reads off a real reference
    (bench.py --ref-fasta) avoid its N runs."""
    rng = np.random.default_rng(seed)
    rng_dup = np.random.default_rng(seed + 100003) if (segdups and dup_frac > 0) else None
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
        tag = ""
        if rng_dup is not None and rng_dup.random() < dup_frac:
            fam = segdups[int(rng_dup.integers(0, len(segdups)))]
            ci, p0, sl = fam[int(rng_dup.integers(0, len(fam)))]
            name, s = contigs[ci]
            ln = min(ln, sl)
            p = p0 + int(rng_dup.integers(0, sl - ln + 1))
            tag = "_dup"
        frag = s[p:p + ln]
        if acgt_only:            # a real reference holds N runs: a read is drawn again until its origin is (almost) free of them
            for _ in range(50):
                if np.isin(frag, _ACGT).mean() > 0.99:
                    break
                p = int(rng.integers(0, len(s) - ln)); frag = s[p:p + ln]
            frag = frag.copy(); bad = ~np.isin(frag, _ACGT)
            frag[bad] = _ACGT[rng.integers(0, 4, size=int(bad.sum()), dtype=np.uint8)]
        strand = "+"
        if rng.random() < 0.5:
            frag = revcomp(frag)
            strand = "-"
        r = mutate(frag, err, rng, mix)
        reads.append((f"r{i}_{name}_{p}_{strand}{tag}", r.tobytes()))
    return reads


def write_reads_fasta(path: str, reads) -> None:
    with open(path, "wb") as fh:
        for name, s in reads:
            fh.write(b">" + name.encode() + b"\n" + s + b"\n")
