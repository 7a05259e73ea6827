"""CPU check of the banded Hirschberg sweep's ALGORITHM (band geometry, lane schedule, score bookkeeping): the lane-by-lane
host model tests/models/hband_model.cpp of lordfast_amd/csrc/lf_hband.hip against a plain full-matrix DP, on low-complexity
strings (many co-optimal paths: the split rule `first row with left + right == best`, lib/edlib/edlib.cpp:1263-1289, must see
every candidate row).  The device kernel itself is compared with the oracle in tests/test_gpu_stages.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "models", "hband_model.cpp")


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hbm") / "libhbm.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", out, SRC], check=True)
    L = C.CDLL(out)
    i8p = np.ctypeslib.ndpointer(np.int8, flags="C")
    lp = np.ctypeslib.ndpointer(np.int64, flags="C")
    L.hbm_node.argtypes = [i8p, C.c_int, i8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, lp]
    L.hbm_ref_node.argtypes = [i8p, C.c_int, i8p, C.c_int, lp]
    L.hbm_shw.argtypes = [i8p, C.c_int, i8p, C.c_int, C.c_int, C.c_int, C.c_int, lp]
    L.hbm_ref_shw.argtypes = [i8p, C.c_int, i8p, C.c_int, lp]
    return L


def mutate(rng, s, rate, alpha):
    out = []
    for c in s:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append(rng.integers(0, alpha))
        if r < rate:
            out.append(rng.integers(0, alpha))
            continue
        out.append(c)
    return np.array(out if out else [0], dtype=np.int8)


def make_pair(rng, n, kind):
    alpha = [2, 4, 4, 3][kind % 4]
    if kind % 5 == 4:                       # tandem repeats of a short unit: ties everywhere
        unit = rng.integers(0, alpha, rng.integers(1, 7)).astype(np.int8)
        q = np.tile(unit, n // len(unit) + 1)[:n].copy()
    else:
        q = rng.integers(0, alpha, n).astype(np.int8)
    rate = [0.02, 0.1, 0.17, 0.3, 0.0][kind % 5]
    if kind % 7 == 6:
        t = rng.integers(0, alpha, max(1, int(n * rng.uniform(0.6, 1.4)))).astype(np.int8)      # unrelated
    else:
        t = mutate(rng, q, rate, alpha)
        if kind % 3 == 0 and len(t) > 40:   # a long gap on one side
            cut = rng.integers(0, len(t) - 30)
            t = np.concatenate([t[:cut], t[cut + rng.integers(1, 30):]]).astype(np.int8)
    if kind % 11 == 10:                     # non-ACGT query bytes never match
        q = q.copy(); q[rng.integers(0, n, max(1, n // 50))] = -1
    return np.ascontiguousarray(q), np.ascontiguousarray(t)


def run_node(model, q, t, k, trial, L, Wmax):
    out = np.zeros(8, dtype=np.int64)
    model.hbm_node(q, len(q), t, len(t), int(k), trial, L, Wmax, out)
    return out


def ref_node(model, q, t):
    out = np.zeros(8, dtype=np.int64)
    model.hbm_ref_node(q, len(q), t, len(t), out)
    return out


@pytest.mark.parametrize("L,Wmax,nmax,ncase", [(64, 1, 6500, 24), (8, 1, 1500, 150), (8, 4, 3000, 150), (4, 2, 900, 150), (4, 16, 3000, 100), (64, 2, 9000, 6), (16, 1, 7000, 60), (32, 1, 7000, 30)])   # (16, 1), (32, 1): lf_hband_group_kernel's groups of sixteen / thirty-two lanes
def test_banded_node_equals_full_matrix_node(model, L, Wmax, nmax, ncase):
    rng = np.random.default_rng(1000 * L + Wmax)
    done = fits = 0
    ws = set()
    for it in range(ncase):
        n = int(rng.integers(70, nmax))
        q, t = make_pair(rng, n, 6 if (L == 64 and Wmax == 2 and it % 2 == 0) else it)      # (unrelated strings: a band for two wavefronts)
        ref = ref_node(model, q, t)
        assert ref[0] == 1
        best = int(ref[4])
        got = run_node(model, q, t, best, 0, L, Wmax)
        done += 1
        if got[0] == -1:
            continue                        # band wider than the lanes hold: the device takes the unbanded kernel
        fits += 1
        ws.add(int(got[5]))
        assert got[0] == 1 and tuple(got[1:5]) == tuple(ref[1:5]), (it, n, len(t), best, got, ref)
        # a trial bound at or above the distance: the distance falls out of the sweep; below it: the sweep says so
        for k0 in (best, best + int(rng.integers(1, 40)), n + len(t)):           # (n + m: the whole matrix -- what a root without a trial bound is swept with)
            g2 = run_node(model, q, t, k0, 1, L, Wmax)
            if g2[0] != -1:
                assert g2[0] == 1 and tuple(g2[1:5]) == tuple(ref[1:5]), (it, "trial", k0, g2, ref)
        if best > abs(len(t) - n) + 2:
            g3 = run_node(model, q, t, best - 1 - int(rng.integers(0, min(20, best - abs(len(t) - n) - 1))), 1, L, Wmax)
            if g3[0] != -1:
                assert g3[0] == 0, (it, "trial below", g3, ref)
    assert fits >= ncase // 3, (fits, done)
    if Wmax > 1:
        assert len(ws) > 1, ws              # more than one wavefront class was exercised


@pytest.mark.parametrize("L,Wmax,nmax,ncase", [(64, 2, 5000, 16), (8, 4, 1200, 150), (4, 4, 600, 150), (4, 16, 2500, 100), (16, 1, 5000, 40), (32, 1, 5000, 20)])
def test_banded_shw_equals_full_matrix_shw(model, L, Wmax, nmax, ncase):
    rng = np.random.default_rng(77 * L + Wmax)
    fits = 0
    for it in range(ncase):
        n = int(rng.integers(70, nmax))
        if it % 4 == 0:
            n = (n // 64 + 1) * 64          # n % 64 == 0: the empty prefix is no candidate (lib/edlib/edlib.cpp:595,615)
        q, t0 = make_pair(rng, n, it)
        # the target of an extension is longer than what the query will use
        t = np.ascontiguousarray(np.concatenate([t0, rng.integers(0, 4, int(rng.integers(1, n // 2 + 2))).astype(np.int8)]))
        ref = np.zeros(8, dtype=np.int64); model.hbm_ref_shw(q, n, t, len(t), ref)
        ed = int(ref[1])
        for k0 in (ed, ed + int(rng.integers(1, 50)), max(0, ed - 1 - int(rng.integers(0, 10))), n + len(t)):
            out = np.zeros(8, dtype=np.int64)
            model.hbm_shw(q, n, t, len(t), k0, L, Wmax, out)
            if out[0] == -1:
                continue
            fits += 1
            if k0 >= ed:
                assert out[0] == 1 and out[1] == ed and out[2] == ref[2], (it, n, len(t), k0, out, ref)
            else:
                assert out[0] == 0, (it, n, len(t), k0, out, ref)
    assert fits >= ncase


@pytest.mark.parametrize("L,Wmax", [(64, 4), (16, 1), (8, 4)])
def test_failed_trial_returns_a_bound_whose_band_succeeds(model, L, Wmax):
    """lf_hrequeue_bound (lf_hirsch.hip): a root whose trial bound was too small goes back to the queue with what its sweep FOUND inside the band -- the cost of a
    real path, so an upper bound of the distance, so the band of that bound holds an optimal path: the second sweep must succeed and give the full matrix's node
    (NW: min F + R; SHW: the last row's minimum)."""
    rng = np.random.default_rng(4242 + L)
    failed_nw = failed_shw = 0
    for it in range(120):
        n = int(rng.integers(70, 2500 if L >= 16 else 900))
        q, t = make_pair(rng, n, it)
        ref = ref_node(model, q, t)
        best = int(ref[4])
        if best >= 3:
            k0 = int(rng.integers(max(1, abs(len(t) - n)), best)) if best > abs(len(t) - n) + 1 else best - 1      # a bound below the distance
            got = run_node(model, q, t, k0, 1, L, Wmax)
            if got[0] == 0:
                failed_nw += 1
                found = int(got[4])
                assert found >= best, ("what a banded sweep finds is the cost of a real path", n, len(t), k0, found, best)
                if found < n + len(t):
                    again = run_node(model, q, t, found, 1, L, Wmax)
                    if again[0] != -1:      # (-1: wider than the lanes hold -- the device takes the next class up)
                        assert again[0] == 1 and list(again[1:5]) == list(ref[1:5]), (n, len(t), k0, found, list(again[:5]), list(ref[:5]))
            elif got[0] == 1:
                assert list(got[1:5]) == list(ref[1:5])
        # SHW: the target goes on behind the alignment
        t2 = np.ascontiguousarray(np.concatenate([t, rng.integers(0, 4, int(rng.integers(1, n // 2 + 2))).astype(np.int8)]))
        r2 = np.zeros(8, dtype=np.int64); model.hbm_ref_shw(q, n, t2, len(t2), r2)
        ed = int(r2[1])
        if ed >= 3:
            k0 = int(rng.integers(1, ed))
            out = np.zeros(8, dtype=np.int64); model.hbm_shw(q, n, t2, len(t2), k0, L, Wmax, out)
            if out[0] == 0:
                failed_shw += 1
                found = int(out[1])
                assert found >= ed, (n, len(t2), k0, found, ed)
                if found < n + len(t2):
                    o2 = np.zeros(8, dtype=np.int64); model.hbm_shw(q, n, t2, len(t2), found, L, Wmax, o2)
                    if o2[0] != -1:
                        assert o2[0] == 1 and (int(o2[1]), int(o2[2])) == (ed, int(r2[2])), (n, len(t2), k0, found, list(o2[:3]), list(r2[:3]))
    assert failed_nw > 20 and failed_shw > 20, (failed_nw, failed_shw)


def test_queue_choice_holds_the_band(tmp_path):
    """lf_hqueue_of / lf_hqueue_of_bound (lordfast_amd/csrc/lf_hirsch.h, __host__ __device__): over 400 000 random nodes -- known distance, trial bound, a failed
    trial's second bound, NW / SHW, lane groups on and off -- the queue chosen is one whose kernel holds the node's band, and the narrowest such one; every queue
    is chosen at least once.  Built with hipcc for the host side only (no kernel is launched)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "hqueue_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "lordfast_amd", "csrc"),
                    "-Wno-unused-result", "-o", exe, os.path.join(HERE, "models", "hqueue_check.hip")], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
