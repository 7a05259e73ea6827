#!/usr/bin/env python3
"""dbg_render.py -- golden reads through the product path; for the first records whose CIGAR / MD differ from the golden SAM,
the common prefix length and the text around the first difference"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tempfile, gzip, shutil
import conftest
import lordfast_amd as la

G = conftest.GOLDEN
d = tempfile.mkdtemp()
for f in os.listdir(G):
    if f.startswith("genome.fa."):
        if f.endswith(".gz"):
            with gzip.open(os.path.join(G, f), "rb") as fi, open(os.path.join(d, "genome.fa"), "wb") as fo:
                shutil.copyfileobj(fi, fo)
        else:
            shutil.copy(os.path.join(G, f), os.path.join(d, f))
names, seqs = conftest.read_fasta(os.path.join(G, "reads.fa.gz"))
h = la.LordFast(os.path.join(d, "genome.fa"), device=0, full_sa=True)
sam, st = h.map_batch(names, seqs, params=la.default_params(**conftest.GOLDEN_CONFIGS["default"]))
exp = conftest.golden_sam("default")
shown = 0
for i, (x, y) in enumerate(zip(sam.split(b"\n"), exp.split(b"\n"))):
    if x == y:
        continue
    fx, fy = x.split(b"\t"), y.split(b"\t")
    for j, (u, v) in enumerate(zip(fx, fy)):
        if u != v:
            k = 0
            while k < min(len(u), len(v)) and u[k] == v[k]:
                k += 1
            print(f"line {i} {fx[0].decode()} field {j}: lengths {len(u)} vs {len(v)}, common prefix {k}")
            print("   got", u[max(0, k - 30):k + 40])
            print("   exp", v[max(0, k - 30):k + 40])
    shown += 1
    if shown >= 4:
        break
print("records differing:", sum(1 for x, y in zip(sam.split(b"\n"), exp.split(b"\n")) if x != y))
h.close()
