"""lordfast_amd -- MI355X-native implementation of lordFAST's seed -> chain -> extend hot path.

The product is lordfast_amd/liblfgpu.so (C ABI in include/lordfast_amd.h, HIP kernels for gfx950 under
lordfast_amd/csrc).  This package is only the thin ctypes binding the tests and bench.py use.
"""
from .api import (LordFast, Params, default_params, lib, lib_path, build_library, device_count,  # noqa: F401
                  LfError, index_build, edlib_batch, chain_n2_batch, chain_clasp_batch, ksw_extend2_batch, read_file, map_batch_multi, ReadBatch)
