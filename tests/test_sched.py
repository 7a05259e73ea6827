"""CPU: the by-bases chunk cutter of the batch scheduler (lf_sched.c: lf_cut_chunks_by_bases) -- one chunk per lane, bounded by a lane's working
set AND by the reads a chunk may hold (reads x sampling positions is a 31-bit index in the seed stage and sizes its buffers): a batch of millions of
short reads must be cut into more chunks, not handed to the seed stage as four huge ones."""
import ctypes as C

import numpy as np

import lordfast_amd as la

READS_MAX = 65536


def cut(lens, n_lanes, ramp=0.0, sampling=1000):
    L = la.lib()
    L.lf_cut_chunks_by_bases.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int]
    L.lf_cut_chunks_by_bases.restype = C.c_int
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    ends = np.zeros(len(lens) + 1, dtype=np.int32)
    k = L.lf_cut_chunks_by_bases(lens.ctypes.data, len(lens), n_lanes, ramp, sampling, ends.ctypes.data, len(ends))
    assert k > 0
    e = ends[:k]
    assert e[-1] == len(lens) and np.all(np.diff(np.concatenate([[0], e])) > 0)      # a partition, no empty chunk
    return e


def test_one_chunk_per_lane_by_bases():
    rng = np.random.default_rng(1)
    lens = rng.integers(5000, 30000, 100000)
    e = cut(lens, 4)
    assert len(e) == 4
    b = np.add.reduceat(lens, np.concatenate([[0], e[:-1]]))
    assert b.max() / b.min() < 1.01                                                   # equal shares of the bases
    e8 = cut(lens, 8, ramp=0.25)
    assert len(e8) == 8
    b8 = np.add.reduceat(lens, np.concatenate([[0], e8[:-1]]))
    assert np.all(np.diff(b8) > 0) and 1.5 < b8[-1] / b8[0] < 1.8                     # the ramp: chunks grow along the batch


def test_small_batches_get_fewer_chunks_not_smaller_ones():
    lens = np.full(13000, 15000)
    assert len(cut(lens, 8)) == 2                                                     # no chunk below 6250 reads
    assert len(cut(np.full(3000, 15000), 8)) == 1


def test_many_short_reads_respect_the_read_cap():
    n = 4_400_000                                                                     # at -c 1000 the old cutter made four chunks of 1.1 M reads: 2^30 samples each
    lens = np.full(n, 1200)
    for sampling, cap in ((1000, READS_MAX), (2000, READS_MAX), (100000, (1 << 30) // 100000)):
        for ramp in (0.0, 0.25):
            e = cut(lens, 4, ramp=ramp, sampling=sampling)
            sizes = np.diff(np.concatenate([[0], e]))
            assert sizes.max() <= cap, (sampling, ramp, sizes.max())
            assert sizes.max() * sampling < (1 << 31)


def test_working_set_bound():
    lens = np.full(40000, 100000)                                                     # 4 Gbases: more than 4 x 768 MB
    e = cut(lens, 4)
    b = np.add.reduceat(lens.astype(np.uint64), np.concatenate([[0], e[:-1]]))
    assert len(e) >= 5 and b.max() <= (768 << 20) + 100000
