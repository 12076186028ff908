"""Profiling helpers: roctx ranges and rocprofv3 recipes.

The reference's only instrumentation is wall-clock per iteration
(/root/reference/part1/main.py:36,53) plus network/CPU time series sampled outside
the program.  On a GPU the useful views are per-kernel time and hardware counters, so
this module offers

* ``range(name)`` - a roctx push/pop pair (visible with ``rocprofv3 --marker-trace``)
  around the phases of a step; a no-op unless enabled;
* ``ROCPROF_RECIPES`` - the command lines used to collect the summaries under
  ``profiles/`` (kernel time, MFMA utilisation / LDS conflicts, HBM bytes).  Counter
  collection is always a run of its own with ``--kernel-trace`` only.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Optional

_roctx = None
_roctx_tried = False


def _load_roctx():
    global _roctx, _roctx_tried
    if _roctx_tried:
        return _roctx
    _roctx_tried = True
    for name in ("libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            _roctx = lib
            break
        except OSError:
            continue
    return _roctx


@contextlib.contextmanager
def range(name: str, enabled: bool = True):
    lib = _load_roctx() if enabled else None
    if lib is not None:
        lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        if lib is not None:
            lib.roctxRangePop()


ROCPROF_RECIPES = {
    # per-kernel time table (copy the *_kernel_stats.csv into profiles/)
    "kernel_stats": "rocprofv3 --kernel-trace --stats --output-format csv "
                    "-d gpurun_out/prof_stats -- python3 bench.py --steps 10 --warmup 3",
    # matrix-core / LDS behaviour, one pass (8 SQ counters at most)
    "mfma_lds": "rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES "
                "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 "
                "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS "
                "-d gpurun_out/prof_mfma -- python3 bench.py --steps 3 --warmup 1",
    # HBM read traffic (FETCH_SIZE takes 3 of the 4 TCC slots) and, separately, writes
    "hbm_read": "rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE "
                "-d gpurun_out/prof_fetch -- python3 bench.py --steps 3 --warmup 1",
    "hbm_write": "rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE "
                 "-d gpurun_out/prof_write -- python3 bench.py --steps 3 --warmup 1",
}


def serialize_kernels_env() -> dict:
    """Environment for triaging stream races: every kernel launch becomes blocking."""
    env = dict(os.environ)
    env["AMD_SERIALIZE_KERNEL"] = "3"
    env["AMD_SERIALIZE_COPY"] = "3"
    return env
