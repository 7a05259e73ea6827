import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import lordfast_amd as la
from conftest import split_ragged
st = np.load('/root/repo/tests/golden/stages.npz')
qs = split_ragged(st["ed_q"].tobytes(), st["ed_qn"]); ts = split_ragged(st["ed_t"].tobytes(), st["ed_tn"]); ops = split_ragged(st["ed_ops"], st["ed_opsn"])
modes = st["ed_mode"]
i = 136
def run(Q, T, M, tag):
    r, _ = la.api.edlib_batch(Q, T, M)
    for k, x in enumerate(r):
        if len(Q[k]) == len(qs[i]) and Q[k] == qs[i] and T[k] == ts[i]:
            ok = np.array_equal(x[2], ops[i])
            d = next((z for z in range(min(len(x[2]), len(ops[i]))) if x[2][z] != ops[i][z]), -1)
            print(tag, "pos", k, "ok", ok, "ed", x[0], "end", x[1], "len", len(x[2]), "firstdiff", d)
print("expected ed", st["ed_dist"][i], "end", st["ed_end"][i], "n", len(qs[i]), "m", len(ts[i]))
run([qs[i]], [ts[i]], [1], "single")
run([qs[i]], [ts[i]], [1], "single again")
run([qs[i], qs[i]], [ts[i], ts[i]], [1, 1], "twice")
run([qs[i]], [ts[i]], [0], "single NW (different problem, just to see)")
run([qs[i], qs[5]], [ts[i], ts[5]], [1, modes[5]], "with other")
# all SHW problems single
bad = []
for k in range(len(qs)):
    if modes[k] == 1:
        r, _ = la.api.edlib_batch([qs[k]], [ts[k]], [1])
        if not np.array_equal(r[0][2], ops[k]): bad.append((k, len(qs[k]), len(ts[k]), int(st["ed_end"][k])))
print("SHW singles bad:", bad)
