"""Flat parameter / gradient storage.

All parameters of a model live in one contiguous fp32 buffer and all gradients in a
second one of the same layout.  Every strategy then works on views: the manual
modes address one view per parameter tensor (the reference iterates
``model.parameters()``, /root/reference/part2/2a/main.py:91, part2/2b/main.py:102),
the bucketed mode all-reduces contiguous slices with no copy-in/copy-out, and the
fused SGD kernel updates the whole model in one launch.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch
from torch import nn

ALIGN_ELEMS = 64  # every tensor starts on a 256-byte boundary of the flat buffer


@dataclass(frozen=True)
class Slot:
    name: str
    shape: Tuple[int, ...]
    offset: int      # in elements
    numel: int

    @property
    def end(self) -> int:
        return self.offset + self.numel


def _round_up(n: int, a: int) -> int:
    return (n + a - 1) // a * a


def plan_slots(named_shapes: Sequence[Tuple[str, Tuple[int, ...]]],
               align: int = ALIGN_ELEMS) -> Tuple[List[Slot], int]:
    slots, cursor = [], 0
    for name, shape in named_shapes:
        n = 1
        for d in shape:
            n *= int(d)
        slots.append(Slot(name, tuple(int(d) for d in shape), cursor, n))
        cursor = _round_up(cursor + n, align)
    return slots, cursor


class FlatModule:
    """Re-homes a module's parameters (and float buffers) into flat storage."""

    def __init__(self, module: nn.Module):
        self.module = module
        params = list(module.named_parameters())
        device = params[0][1].device
        self.slots, total = plan_slots([(n, tuple(p.shape)) for n, p in params])
        self.params = torch.zeros(total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(total, dtype=torch.float32, device=device)
        self._param_list = []
        with torch.no_grad():
            for slot, (_, p) in zip(self.slots, params):
                view = self.params[slot.offset:slot.end].view(slot.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grads[slot.offset:slot.end].view(slot.shape)
                self._param_list.append(p)
        # float buffers (BatchNorm running statistics) -> one flat tensor for one broadcast
        fbufs = [(n, b) for n, b in module.named_buffers() if b.dtype.is_floating_point]
        self.buffer_slots, btotal = plan_slots([(n, tuple(b.shape)) for n, b in fbufs], align=1)
        self.buffers = torch.zeros(max(btotal, 0), dtype=torch.float32, device=device)
        with torch.no_grad():
            for slot, (name, b) in zip(self.buffer_slots, fbufs):
                view = self.buffers[slot.offset:slot.end].view(slot.shape)
                view.copy_(b)
                owner, _, leaf = name.rpartition(".")
                mod = module.get_submodule(owner) if owner else module
                mod._buffers[leaf] = view

    def grad_views(self) -> List[torch.Tensor]:
        return [self.grads[s.offset:s.end] for s in self.slots]

    def zero_grad(self) -> None:
        self.grads.zero_()
        # autograd may have replaced a .grad view (it does not when one is present); re-pin
        for slot, p in zip(self.slots, self._param_list):
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * slot.offset:
                p.grad = self.grads[slot.offset:slot.end].view(slot.shape)


def plan_buckets(sizes_bytes: Sequence[int], cap_bytes: int,
                 first_cap_bytes: int = 1 << 20) -> List[Tuple[int, int]]:
    """Group tensors, walked in REVERSE order (gradients become ready last layer
    first), into buckets.  A bucket closes once its size reaches its cap after adding
    a tensor; the first bucket uses a small cap so communication starts early.
    Returns ``(lo, hi)`` index ranges into the forward-ordered list, in the order the
    buckets complete during backward; each range is contiguous in the flat buffer."""
    buckets: List[Tuple[int, int]] = []
    hi = len(sizes_bytes)
    acc = 0
    cap = first_cap_bytes if first_cap_bytes > 0 else cap_bytes
    for i in range(len(sizes_bytes) - 1, -1, -1):
        acc += int(sizes_bytes[i])
        if acc >= cap:
            buckets.append((i, hi))
            hi, acc, cap = i, 0, cap_bytes
    if hi > 0:
        buckets.append((0, hi))
    return buckets


def cap_tail_bucket(buckets: List[Tuple[int, int]], sizes_bytes: Sequence[int],
                    tail_cap_bytes: int) -> List[Tuple[int, int]]:
    """Split the last bucket (the earliest layers: it completes with the final wgrad, so its
    all-reduce cannot hide behind backward) so that what remains exposed is at most
    ``tail_cap_bytes``.  ``buckets`` are ``[lo, hi)`` index ranges in completion order."""
    if tail_cap_bytes <= 0 or not buckets:
        return list(buckets)
    lo, hi = buckets[-1]
    acc, k = 0, lo
    while k < hi and acc + sizes_bytes[k] <= tail_cap_bytes:
        acc += sizes_bytes[k]
        k += 1
    if k == lo or k == hi:          # nothing fits, or the bucket is small enough already
        return list(buckets)
    return list(buckets[:-1]) + [(k, hi), (lo, k)]
