"""The gradient-synchronisation strategies (the only thing the reference's four
scripts differ in).

* ``GatherScatterSync`` - every parameter's gradient is gathered on rank 0, summed
  there and scattered back (/root/reference/part2/2a/main.py:89-116).  Result: SUM.
* ``AllReduceSync``     - one all_reduce(SUM) per parameter tensor after backward
  (/root/reference/part2/2b/main.py:101-106).  Result: SUM.
* ``BucketedSync``      - what ``DistributedDataParallel(model, bucket_cap_mb=25)``
  does for part3 (/root/reference/part3/main.py:137): rank 0's parameters and buffers
  are broadcast once, gradients are grouped into buckets in reverse parameter order,
  each bucket's all-reduce starts as soon as its last gradient is produced and
  overlaps the rest of backward, the result is the MEAN, and BatchNorm buffers are
  re-broadcast from rank 0 before every forward.

All strategies operate on views of the flat gradient buffer, so nothing is copied
in or out.  ``reduce`` ("sum" | "mean") can override the per-mode default so the
manual modes can be compared against a single-process run of the global batch.

The classes here drive the stock-torch backend only (autograd hooks mark gradients
ready).  The native engine performs the same three protocols in C++ without passing
through this module: the bucket reducer (csrc/comm/bucket_reducer.cpp) for part3 and
``ManualSync`` (csrc/comm/manual_sync.cpp: grouped send / recv with an in-place root for
part2a, one all-reduce per tensor for part2b), both enqueued by the engine itself between
backward and the update.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .comm import Communicator
from .flat import FlatModule, plan_buckets


class GradSync:
    name = "none"

    def __init__(self, comm: Optional[Communicator], reduce: str = "sum"):
        self.comm = comm
        self.reduce = reduce

    def comm_only(self) -> None:            # the step's collectives and nothing else
        self.finish()

    def prepare_forward(self) -> None:      # before every forward pass
        pass

    def begin_backward(self) -> None:       # after zero_grad, before backward
        pass

    def finish(self) -> None:               # after backward, before optimizer.step
        pass

    def close(self) -> None:
        pass


class NoSync(GradSync):
    pass


class GatherScatterSync(GradSync):
    name = "gather_scatter"

    def __init__(self, comm: Communicator, grads: Sequence[torch.Tensor], reduce: str = "sum",
                 root: int = 0):
        super().__init__(comm, reduce)
        self.grads = list(grads)
        self.root = root
        w = comm.world_size
        self.is_root = comm.rank == root
        # root-side landing area: W slabs per parameter, allocated once (the reference
        # allocates fresh zero tensors every iteration, part2/2a/main.py:96-98,112)
        self.slabs = None
        if self.is_root:
            total = sum(g.numel() for g in self.grads)
            total = (total + 3) // 4 * 4        # the slab-sum kernel moves 16-byte vectors
            self._slab_store = torch.zeros(w, total, dtype=self.grads[0].dtype,
                                           device=self.grads[0].device)
            self.slabs, off = [], 0
            for g in self.grads:
                self.slabs.append([self._slab_store[r, off:off + g.numel()] for r in range(w)])
                off += g.numel()

    def _root_sum(self) -> None:
        """slab[0] += slab[1] + ... + slab[W-1] for every parameter in one pass."""
        store = self._slab_store
        if store.is_cuda:
            from ..ops import native
            native.sum_slabs_(store)
        else:
            store[0].add_(store[1:].sum(dim=0))

    def finish(self) -> None:
        comm, w = self.comm, self.comm.world_size
        comm.group_start()
        for i, g in enumerate(self.grads):
            comm.gather(g, self.slabs[i] if self.is_root else None, self.root)
        comm.group_end()
        if self.is_root:
            self._root_sum()
            if self.reduce == "mean":
                self._slab_store[0].div_(w)
        comm.group_start()
        for i, g in enumerate(self.grads):
            # every rank receives the same summed tensor (part2/2a/main.py:109-114)
            chunks = [self.slabs[i][0]] * w if self.is_root else None
            comm.scatter(g, chunks, self.root)
        comm.group_end()


class AllReduceSync(GradSync):
    name = "all_reduce"

    def __init__(self, comm: Communicator, grads: Sequence[torch.Tensor], reduce: str = "sum"):
        super().__init__(comm, reduce)
        self.grads = list(grads)

    def finish(self) -> None:
        self.comm.group_start()
        for g in self.grads:
            self.comm.all_reduce(g, self.reduce)
        self.comm.group_end()


class BucketedSync(GradSync):
    name = "ddp"

    def __init__(self, comm: Communicator, flat: FlatModule, reduce: str = "mean",
                 bucket_cap_mb: float = 25.0, broadcast_buffers: bool = True,
                 int_buffers: Optional[List[torch.Tensor]] = None,
                 first_bucket_cap_mb: float = 1.0):
        super().__init__(comm, reduce)
        self.flat = flat
        self.broadcast_buffers = broadcast_buffers and comm.world_size > 1
        self.int_buffers = int_buffers or []
        sizes = [s.numel * 4 for s in flat.slots]
        self.buckets = plan_buckets(sizes, int(bucket_cap_mb * (1 << 20)),
                                    int(first_bucket_cap_mb * (1 << 20)))
        self._bucket_of = {}
        self._slices = []
        for b, (lo, hi) in enumerate(self.buckets):
            for i in range(lo, hi):
                self._bucket_of[i] = b
            self._slices.append(flat.grads[flat.slots[lo].offset:flat.slots[hi - 1].end])
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        self.enabled = True                 # False: hooks and broadcasts do nothing (timing aid)
        # construction-time broadcast of rank 0's state (DDP constructor behaviour)
        if comm.world_size > 1:
            comm.broadcast(flat.params, 0)
            self._broadcast_buffers(force=True)
        for i, p in enumerate(flat._param_list):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _broadcast_buffers(self, force: bool = False) -> None:
        if not (self.broadcast_buffers or force):
            return
        if self.flat.buffers.numel():
            self.comm.broadcast(self.flat.buffers, 0)
        for t in self.int_buffers:
            self.comm.broadcast(t, 0)

    def _make_hook(self, index: int):
        def hook(_param):
            if not self.enabled:
                return
            b = self._bucket_of[index]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._handles.append(self.comm.all_reduce_async(self._slices[b], self.reduce))
        return hook

    def prepare_forward(self) -> None:
        if self.enabled and self.flat.module.training:
            self._broadcast_buffers()

    def begin_backward(self) -> None:
        self._pending = [hi - lo for lo, hi in self.buckets]
        self._handles = []

    def finish(self) -> None:
        if not self.enabled:
            return
        if any(self._pending):
            raise RuntimeError("backward finished with gradients missing from a bucket")
        for h in self._handles:
            h.wait()
        self._handles = []

    def comm_only(self) -> None:
        self._broadcast_buffers()
        handles = [self.comm.all_reduce_async(sl, self.reduce) for sl in self._slices]
        for h in handles:
            h.wait()

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
