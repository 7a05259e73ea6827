"""Stage-API calls, batch-API calls and index release / reload mixed in one process.

With the ~10^2 HIP streams of eight lanes in use, ROCm 7.2's own device-wide wait (inside hipFree / hipMalloc) could fail to
arm its signal handlers and never return; `lf_mem.hip` now waits for every lane stream itself before such calls
(`lfg_quiesce`).  The sequence below is the shape that used to hang (stage calls first, then batches on all lanes with the
clasp chainer's extra streams, then the release of the index); `LF_WATCHDOG` (conftest) turns a regression into an abort."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_CONFIGS, golden_sam

pytestmark = pytest.mark.gpu


def _ragged(buf, lens):
    out, o = [], 0
    for n in lens:
        out.append(buf[o:o + int(n)]); o += int(n)
    return out


def test_stage_calls_then_batches_then_release(golden_dir, golden_reads, stages):
    import lordfast_amd as la
    names, seqs = golden_reads
    qs = _ragged(stages["ed_q"].tobytes(), stages["ed_qn"])
    ts = _ragged(stages["ed_t"].tobytes(), stages["ed_tn"])
    for rep in range(2):
        res, _ = la.edlib_batch(qs, ts, stages["ed_mode"])                      # byte-string stage API: its own uploads, class streams
        assert [r[0] for r in res] == [int(x) for x in stages["ed_dist"]]
        h = la.LordFast(os.path.join(golden_dir, "genome.fa"), device=0, full_sa=True)
        for cfg in GOLDEN_CONFIGS:                                               # incl. both clasp option sets
            sam, _ = h.map_batch(names, seqs, params=la.default_params(threads=8, **GOLDEN_CONFIGS[cfg]))
            assert sam == golden_sam(cfg), cfg
        h.close()                                                                # hipFree of the index: used to hang here
