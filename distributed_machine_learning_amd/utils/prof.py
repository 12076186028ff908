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
    # per-kernel time table: the result database is summarised by tools/rocpd_summary.py
    # (profiles/part3_*_kernels.txt) and tools/rocpd_gaps.py (idle gaps between kernels)
    "kernel_stats": "rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bn -o bn -- "
                    "python3 bench.py --steps 50 --warmup 10",
    # matrix-core / LDS behaviour of every conv kernel, one pass (8 SQ counters at most);
    # summarised by tools/rocpd_pmc.py (profiles/conv_kernels_pmc.txt)
    "mfma_lds": "rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT "
                "SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE "
                "-d gpurun_out/pmc1 -o pmc -- python3 tools/pmc_conv.py",
    # L2 behaviour and MFMA op counts, second pass
    "l2_mfma_ops": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum "
                   "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_LDS_DATA_FIFO_FULL "
                   "SQ_LDS_CMD_FIFO_FULL -d gpurun_out/pmc2 -o pmc -- python3 tools/pmc_conv.py",
    # one kernel, one plan (LAYER / KIND / PLAN in the environment): tools/pmc_one.py
    "one_kernel": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES "
                  "GRBM_GUI_ACTIVE -d gpurun_out/pmcA -o a -- python3 tools/pmc_one.py",
}


def serialize_kernels_env() -> dict:
    """Environment for triaging stream races: every kernel launch becomes blocking."""
    env = dict(os.environ)
    env["AMD_SERIALIZE_KERNEL"] = "3"
    env["AMD_SERIALIZE_COPY"] = "3"
    return env
