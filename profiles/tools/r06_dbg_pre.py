import os, sys, gzip, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
import lordfast_amd as la
gold = "/root/repo/tests/golden"
import shutil, tempfile
tmp = tempfile.mkdtemp()
for f in os.listdir(gold):
    if f.startswith("genome.fa.") and not f.endswith(".gz"): shutil.copy(os.path.join(gold, f), tmp)
names, seqs = [], []
with gzip.open(os.path.join(gold, "reads.fa.gz"), "rb") as fh:
    for line in fh:
        line = line.strip()
        if line.startswith(b">"): names.append(line[1:])
        elif line: seqs.append(line)
lf = la.LordFast(os.path.join(tmp, "genome.fa"), device=0)
L = lf.L
L.lfg_host_alloc.restype = C.c_void_p; L.lfg_host_alloc.argtypes = [C.c_size_t]
cap = 8 << 20
buf = L.lfg_host_alloc(cap)
plain = np.zeros(cap, dtype=np.uint8)
os.environ["LF_LANES"] = sys.argv[2] if len(sys.argv) > 2 else "3"
os.environ["LF_CHUNK_READS"] = sys.argv[1] if len(sys.argv) > 1 else "7"
ref, _ = lf.map_batch(names, seqs)
b = la.ReadBatch(names, seqs, None, min_read_len=1000)
for it in range(6):
    for label, ptr in (("pinned", buf), ("plain", plain.ctypes.data)):
        C.memset(ptr, 0x23, cap)
        ln, st = lf.map_batch_from(b, ptr, cap)
        got = C.string_at(ptr, ln)
        if got != ref:
            gl, rl = got.split(b"\n"), ref.split(b"\n")
            bad = [i for i, (x, y) in enumerate(zip(gl, rl)) if x != y]
            f = bad[0]
            gf, rf = gl[f].split(b"\t"), rl[f].split(b"\t")
            fields = [k for k, (x, y) in enumerate(zip(gf, rf)) if x != y]
            print(it, label, "DIFF lines", len(bad), "first", f, gf[0], "fields", fields, "got9 head", gf[9][:24] if len(gf) > 9 else None, flush=True)
        else:
            print(it, label, "ok", flush=True)
