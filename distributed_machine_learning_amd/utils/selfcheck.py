"""Numerical self-checks of a multi-rank run, made inside the run on the live communicator.

The reference's authors validated their three synchronisation modes by hand
(/root/reference/group25.pdf p3, "Sanity Checks"): the same seed gives the same model on every
node, and the loss of the distributed runs tracks the single-node run at the same iteration
(/root/reference/part2/2a/main.py:89-116, part2/2b/main.py:101-106, part3/main.py:137 against
part1/main.py).  That all ranks END with identical parameters proves little: every all-reduce
leaves identical results on all ranks even when a peer read stale gradients.  These checks
compare against an independent computation instead:

* ``handoff_forms_agree``   the same steps from one snapshot with flag-word hand-offs and with
                            event pairs must produce the same parameter BITS (same schedule, same
                            arithmetic; a rank whose peers read its gradients before they were
                            visible breaks this);
* ``dp_matches_single``     K mean-reduced data-parallel steps on W ranks against the same K steps
                            of the W x B global batch on rank 0 alone (no BatchNorm: per-rank
                            statistics would differ by construction);
* ``manual_modes_match``    the gather / scatter and per-tensor all-reduce modes against the
                            bucketed all-reduce over the same data (the first two are built from
                            grouped ncclSend / ncclRecv and one ncclAllReduce per tensor).

Everything here runs outside any timed region and leaves the caller's backend as it found it.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..config import SEED, SYNC_ALL_REDUCE, SYNC_DDP, SYNC_GATHER_SCATTER, SYNC_NONE, TrainConfig


def _all_agree(comm, ok: bool, device) -> bool:
    """True only if every rank passes True (bounded wait)."""
    if comm is None:
        return bool(ok)
    t = torch.tensor([0.0 if ok else 1.0], device=device)
    comm.all_reduce(t)
    comm.guard_sync(comm.timeout_s)
    return float(t.item()) == 0.0


def handoff_forms_agree(backend, comm, batches: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                        steps: int = 3) -> Dict[str, object]:
    """Run ``steps`` steps with flag-word hand-offs and again with event pairs, both from the
    backend's current state (restored afterwards, as are the forms it came with).  Every rank
    compares its own two parameter vectors bit for bit; the verdict is the AND over ranks."""
    was = backend.handoff_forms()
    snap = backend.snapshot()
    finals = []
    # (the event form twice: a transport that does not reproduce its OWN results bit for bit - a
    # collective whose reduction order depends on timing - makes the comparison meaningless)
    for by_value in (True, False, False):
        backend.set_handoff_forms(by_value)
        for i in range(steps):
            backend.train_step(*batches[i % len(batches)])
        torch.cuda.synchronize(backend.device)
        finals.append((backend.params.clone(), backend.bn_buffers.clone(), backend.handoff_forms()))
        backend.restore(snap)

    def same_bits(a, b):
        return bool(torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and
                    torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)))

    same = same_bits(finals[0], finals[1])
    repeatable = same_bits(finals[1], finals[2])
    moved = not torch.equal(finals[0][0], snap[0])          # (the steps did change something)
    torch.cuda.synchronize(backend.device)
    backend.engine.set_value_hand_offs(was["engine"] == "value")
    if backend.reducer is not None:
        backend.reducer.set_departure_by_value(was["buckets"] == "value")
    repeatable_everywhere = _all_agree(comm, repeatable, backend.device)
    agree_everywhere = _all_agree(comm, same and moved, backend.device)
    out = {"steps": steps, "forms_compared": [finals[0][2], finals[1][2]],
           "event_form_repeats_bitwise": repeatable_everywhere}
    if not repeatable_everywhere:
        out["ok"] = None
        out["inconclusive"] = ("two runs of the event form differ: the transport does not reproduce its "
                               "own results bit for bit, so the forms cannot be compared this way")
    else:
        out["ok"] = agree_everywhere
    return out


def _rel_err(a: torch.Tensor, b: torch.Tensor, slots) -> float:
    worst = 0.0
    for s in slots:
        if len(s.shape) < 2:
            continue
        x = a[s.offset:s.offset + s.numel].double()
        y = b[s.offset:s.offset + s.numel].double()
        worst = max(worst, float((x - y).norm() / y.norm().clamp_min(1e-12)))
    return worst


def _small_cfg(base: TrainConfig, sync: str, batch: int, world: int, rank: int) -> TrainConfig:
    return base.replace(sync=sync, grad_reduce="mean", batch_size=batch, batch_norm=False, lr=0.01,
                        num_nodes=world, rank=rank, use_graph=False, fused_optimizer=False,
                        update_overlap=0, summary=False, eval=False)


def _run_small(cfg: TrainConfig, device, comm, data, labels, steps: int):
    """A fresh backend under the reference's seed, ``steps`` steps over the given batches, its
    final parameters (a clone) and its slots."""
    from ..engine.native_backend import NativeBackend
    torch.manual_seed(SEED)
    be = NativeBackend(cfg, device, comm)
    be.train_mode(True)
    for i in range(steps):
        be.train_step(data[i % len(data)], labels[i % len(labels)])
    torch.cuda.synchronize(device)
    if comm is not None:
        comm.guard_sync(comm.timeout_s)
    out = be.params.clone()
    slots = list(be.slots)
    be.close()
    return out, slots


def dp_and_manual_modes(base_cfg: TrainConfig, comm, device, layout: str, per_rank: int = 32,
                        steps: int = 3, modes: Sequence[str] = (SYNC_DDP, SYNC_GATHER_SCATTER,
                                                                SYNC_ALL_REDUCE)) -> Dict[str, object]:
    """Mean-reduced data parallelism on the live communicator against rank 0 alone on the global
    batch, for each sync mode in ``modes``; all on freshly built small engines (``per_rank`` images
    per rank, no BatchNorm, lr 0.01, bf16 or fp32 as in ``base_cfg``).

    Returns per mode the worst relative error of a weight tensor against the single-process run
    (rank 0's number, broadcast) and whether all ranks hold the same bits."""
    from ..data import synthetic_cifar
    from ..data.transforms import BatchTransform
    world, rank = comm.world_size, comm.rank
    n_img = per_rank * world * steps
    ds = synthetic_cifar(n_img, SEED + 1).to(device)
    tf = BatchTransform(True, layout, SEED + 1)
    mine_x, mine_y, glob_x, glob_y = [], [], [], []
    for i in range(steps):
        base = i * per_rank * world
        idx = torch.arange(per_rank, device=device) * world + rank + base        # rank-strided
        imgs, labels = ds.take(idx)
        mine_x.append(tf(imgs, idx).contiguous())
        mine_y.append(labels.contiguous())
        if rank == 0:
            gidx = torch.arange(per_rank * world, device=device) + base
            gi, gl = ds.take(gidx)
            glob_x.append(tf(gi, gidx).contiguous())
            glob_y.append(gl.contiguous())
    single = None
    if rank == 0:
        single, _ = _run_small(_small_cfg(base_cfg, SYNC_NONE, per_rank * world, 1, 0), device, None,
                               glob_x, glob_y, steps)
    out: Dict[str, object] = {"per_rank_batch": per_rank, "steps": steps, "modes": {}}
    ok = True
    for mode in modes:
        cfg = _small_cfg(base_cfg, mode, per_rank, world, rank).replace(bucket_cap_mb=4.0,
                                                                         first_bucket_cap_mb=1.0,
                                                                         tail_bucket_cap_mb=0.0)
        params, slots = _run_small(cfg, device, comm, mine_x, mine_y, steps)
        err = torch.zeros(1, device=device)
        if rank == 0:
            err[0] = _rel_err(params, single, slots)
        comm.broadcast(err, 0)
        checksum = int(params.view(torch.int32).to(torch.int64).sum().item())
        sums = comm.all_gather_scalars(checksum, torch.int64)
        same = bool((sums == sums[0]).all())
        e = float(err.item())
        out["modes"][mode] = {"rel_err_vs_single": e, "ranks_identical": same}
        ok = ok and same and e < 1e-4
    out["ok"] = ok
    return out


def timed_window(base_cfg: TrainConfig, comm, device, batches_for, steps: int = 20,
                 warmup: int = 5, **overrides) -> Optional[float]:
    """ms per step of a freshly built backend with ``overrides`` applied to ``base_cfg`` (another
    dtype, another wire type), max over ranks; ``batches_for(layout)`` supplies its input."""
    import time
    from ..engine.native_backend import NativeBackend
    torch.manual_seed(SEED)
    be = NativeBackend(base_cfg.replace(**overrides), device, comm)
    be.train_mode(True)
    batches = batches_for(be.loader_layout)
    for i in range(warmup):
        be.train_step(*batches[i % len(batches)])
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize(device)
    t = time.perf_counter()
    for i in range(steps):
        be.train_step(*batches[i % len(batches)])
    if comm is not None:
        comm.barrier()
    torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t) / steps * 1e3
    be.close()
    if comm is not None:
        ms = float(comm.all_gather_scalars(ms, torch.float32).max())
    return round(ms, 4)


def transport_facts() -> Dict[str, object]:
    """RCCL's version and every NCCL_* / RCCL_* / HSA_* setting of this process's environment."""
    import os
    from ..ops import native
    v = int(native.lib().rccl_version())
    env = {k: os.environ[k] for k in sorted(os.environ)
           if k.startswith(("NCCL_", "RCCL_", "HSA_", "GPU_MAX_HW_QUEUES"))}
    return {"rccl_version": v, "env": env}
