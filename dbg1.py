import os, sys, gzip, re, subprocess
sys.path.insert(0, "tests")
import numpy as np
from conftest import GOLDEN, GOLDEN_CONFIGS, golden_sam, read_fasta
import lordfast_amd as la, tempfile, shutil
tmp = tempfile.mkdtemp()
for f in os.listdir(GOLDEN):
    if f.startswith("genome.fa.") and not f.endswith(".gz"): shutil.copy(os.path.join(GOLDEN, f), tmp)
names, seqs = read_fasta(os.path.join(GOLDEN, "reads.fa.gz"))
cfg = sys.argv[1] if len(sys.argv) > 1 else "k12c300m20"
exp = golden_sam(cfg).split(b"\n")
def cig_ops(c):
    return [(int(n), o) for n, o in re.findall(rb"(\d+)([MIDS])", c)]
def run(tag):
    lf = la.LordFast(os.path.join(tmp, "genome.fa"), device=0)
    sam, st = lf.map_batch(names, seqs, params=la.default_params(**GOLDEN_CONFIGS[cfg]))
    lf.close()
    got = sam.split(b"\n")
    bad = 0
    for i, (a, b) in enumerate(zip(got, exp)):
        if a != b:
            bad += 1
            fa, fb = a.split(b"\t"), b.split(b"\t")
            print(tag, "line", i, fa[0], "flag", fa[1], fb[1], "pos", fa[3], fb[3])
            ca, cb = cig_ops(fa[5]), cig_ops(fb[5])
            qa = 0
            for k, (x, y) in enumerate(zip(ca, cb)):
                if x != y:
                    print("   first diff at cigar op", k, x, y, "query offset", qa, "of", len(ca), len(cb)); break
                if x[1] in b"MIS": qa += x[0]
            for j, (u, v) in enumerate(zip(fa, fb)):
                if u != v and j != 5: print("   field", j, u[:60], v[:60])
    print(tag, "mismatching lines:", bad, "of", len(exp))
run("default")
os.environ["LF_NO_LAZY"] = "1"; run("nolazy"); del os.environ["LF_NO_LAZY"]
os.environ["LF_HOST_CIGAR"] = "1"; run("hostcigar"); del os.environ["LF_HOST_CIGAR"]
os.environ["LF_HIST_STATS"] = "1"; os.environ["LF_LANES"]="1"; run("stats")
