"""Communication / compute split of a training step.

The reference's authors estimated communication time as (N-node iteration time minus
1-node compute time) from two separate runs (/root/reference/group25.pdf p10).  Here the
same quantity is measured inside one job, three timings over the same batches:

* ``step_ms``          the full step (compute + gradient collectives, overlapped or not)
* ``compute_only_ms``  the same step with every gradient collective detached
* ``comm_only_ms``     exactly the step's gradient collectives, nothing else

``exposed_comm_ms = step - compute_only`` is what communication adds to the iteration and
``overlap_fraction = 1 - exposed / comm_only`` is the share of it hidden behind backward
(0 for the two manual modes, which synchronise after backward has finished).

A backend takes part by providing ``set_comm_enabled(bool)`` and ``comm_only_step()``.
Ranks drift apart while collectives are detached, so run this after the training that
matters.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch


def _sync(device: torch.device, comm) -> None:
    if comm is not None:
        comm.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _timed(fn: Callable[[int], None], n: int, device: torch.device, comm) -> float:
    _sync(device, comm)
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    _sync(device, comm)
    return (time.perf_counter() - t) / n * 1e3


def split_from_times(step_ms: float, compute_ms: float, comm_ms: float) -> Dict[str, Optional[float]]:
    """The derived quantities from the three raw timings."""
    exposed = max(step_ms - compute_ms, 0.0)
    frac = round(1.0 - min(exposed / comm_ms, 1.0), 3) if comm_ms > 0 else None
    return {"step_ms": round(step_ms, 4), "compute_only_ms": round(compute_ms, 4),
            "comm_only_ms": round(comm_ms, 4), "exposed_comm_ms": round(exposed, 4),
            "overlap_fraction": frac}


def measure_comm_split(backend, comm, batches: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                       n: int = 10, warmup: int = 1) -> Optional[Dict[str, Optional[float]]]:
    """Time ``n`` steps in each of the three configurations; None when the run has no
    collectives to account for."""
    if comm is None or not hasattr(backend, "set_comm_enabled") or not batches:
        return None
    device = backend.device

    def step(i: int) -> None:
        backend.train_step(*batches[i % len(batches)])

    for i in range(warmup):
        step(i)
    full = _timed(step, n, device, comm)
    backend.set_comm_enabled(False)
    try:
        for i in range(warmup):
            step(i)
        compute = _timed(step, n, device, comm)
    finally:
        backend.set_comm_enabled(True)
    for _ in range(warmup):
        backend.comm_only_step()
    comm_ms = _timed(lambda i: backend.comm_only_step(), n, device, comm)
    return split_from_times(full, compute, comm_ms)


def format_comm_split(mode: str, s: Dict[str, Optional[float]]) -> str:
    frac = s["overlap_fraction"]
    return (f"Communication ({mode}): step {s['step_ms']:.3f} ms, compute only "
            f"{s['compute_only_ms']:.3f} ms, collectives alone {s['comm_only_ms']:.3f} ms, "
            f"exposed {s['exposed_comm_ms']:.3f} ms, overlap "
            + (f"{100.0 * frac:.0f}%" if frac is not None else "n/a"))
