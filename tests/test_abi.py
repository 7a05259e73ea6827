"""CPU-only checks of the boundary: liblfgpu.so loads, exports every function include/lordfast_amd.h declares,
refuses to run without a device (no CPU fallback), and its host-side order-exact sort agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, split_ragged

HEADER = os.path.join(ROOT, "include", "lordfast_amd.h")


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", txt, flags=re.S)
    txt = re.sub(r"typedef\s+enum\s*\{.*?\}\s*\w+\s*;", "", txt, flags=re.S)
    names = re.findall(r"\b([A-Za-z_]\w*)\s*\([^;{}]*\)\s*;", txt)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_library_exports_every_declared_symbol():
    import lordfast_amd as la
    L = la.lib()
    names = declared_functions()
    assert len(names) >= 30, names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    for g in ("lf_global_params", "lf_global_output", "lf_global_no_header", "lf_global_cmdline"):
        C.c_int.in_dll(L, g)


def test_no_device_means_loud_failure(golden_dir):
    """on a box without a gfx950 the product must refuse, never silently compute on the CPU"""
    import lordfast_amd as la
    if la.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(la.LfError, match="no gfx950 device"):
        la.LordFast(os.path.join(golden_dir, "genome.fa"))
    with pytest.raises(la.LfError, match="no gfx950 device"):
        la.edlib_batch([b"ACGT"], [b"ACGT"], [0])
    with pytest.raises(la.LfError, match="no gfx950 device"):
        la.index_build(os.path.join(golden_dir, "genome.fa"))


def test_params_default_matches_reference_defaults():
    import lordfast_amd as la
    L = la.lib()
    p = la.Params()
    L.lf_params_default(C.byref(p))
    assert (p.min_anchor_len, p.sampling_count, p.max_map, p.min_read_len, p.max_ref_hits) == (14, 1000, 10, 1000, 1000)
    assert (p.chain_reward, p.chain_penalty, p.gap_penalty) == (9.3, 11.4, 0.15)


def test_host_introsort_order_matches_std_sort(stages, oracle_lib):
    """lf_stdsort.h (host glue) vs the reference's std::sort order stored in the golden vectors + oracle fuzz"""
    import lordfast_amd as la
    from lordfast_amd.api import _triples_to_seeds, _seeds_to_triples
    L = la.lib()
    L.lf_sort_seeds_by_qpos.argtypes = [C.c_void_p, C.c_long]
    ins = split_ragged(stages["chain_in"], stages["chain_n"])
    srt = split_ragged(stages["chain_sorted"], stages["chain_n"])
    for a, b in zip(ins, srt):
        s = _triples_to_seeds(a)
        L.lf_sort_seeds_by_qpos(s.ctypes.data, len(s))
        assert np.array_equal(_seeds_to_triples(s.reshape(-1)), b)
    orc = oracle_lib.Oracle()
    rng = np.random.default_rng(0)
    for it in range(200):
        n = int(rng.integers(0, 3000))
        q = rng.integers(0, max(2, n // int(rng.integers(1, 40))), size=n).astype(np.uint32)
        if it % 3 == 0:
            q = np.sort(q)[::-1].copy()
        tr = np.stack([rng.integers(0, 1 << 30, size=n).astype(np.uint32), q, rng.integers(12, 30, size=n).astype(np.uint32)], axis=1)
        s = _triples_to_seeds(tr)
        L.lf_sort_seeds_by_qpos(s.ctypes.data, len(s))
        assert np.array_equal(_seeds_to_triples(s.reshape(-1)), orc.sort_seeds(tr)), it


def test_read_group_line_like_set_read_group():
    """lf_params_set_read_group follows set_read_group (src/CommandLineParser.cpp:85-124): escapes, ID field, errors"""
    import lordfast_amd as la
    L = la.lib()
    L.lf_params_set_read_group.argtypes = [C.POINTER(la.Params), C.c_char_p]
    p = la.default_params()
    assert L.lf_params_set_read_group(C.byref(p), b"@RG\\tID:grp1\\tSM:x\\\\y") == 0
    assert p.read_group == b"@RG\tID:grp1\tSM:x\\y" and p.read_group_id == b"grp1"
    for bad, msg in ((b"RG\\tID:a", b"does not start with @RG"), (b"@RG\tID:a", b"literal <tab>"), (b"@RG\\tSM:a", b"no ID within")):
        q = la.default_params()
        assert L.lf_params_set_read_group(C.byref(q), bad) != 0
        assert msg in L.lf_last_error()
    assert C.sizeof(la.Params) == C.sizeof(C.c_int) * 6 + 8 * 3 + 4 + 256 + 1000 + 4    # the C struct incl. tail padding


def test_clasp_entry_points_need_a_device():
    import lordfast_amd as la
    if la.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(la.LfError, match="no gfx950 device"):
        la.chain_clasp_batch([np.array([[100, 0, 20], [130, 30, 20]], dtype=np.uint32)])
