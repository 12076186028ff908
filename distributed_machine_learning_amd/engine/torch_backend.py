"""Stock-torch execution backend.

Runs the network with ``torch.nn`` modules and autograd on whatever device it is
given.  It is (a) the CPU configuration of the benchmark matrix (the reference's
part1 runs on CPU with four threads, /root/reference/part1/main.py:15-16), (b) the
golden model the native engine is checked against, and (c) the way the three sync
strategies are exercised multi-process without a GPU.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from ..config import SYNC_ALL_REDUCE, SYNC_DDP, SYNC_GATHER_SCATTER, SYNC_NONE, TrainConfig
from ..models.vgg import VGG
from ..parallel.comm import Communicator
from ..parallel.flat import FlatModule
from ..parallel.strategies import (AllReduceSync, BucketedSync, GatherScatterSync, GradSync,
                                   NoSync)


def build_sync(cfg: TrainConfig, comm: Optional[Communicator], flat: FlatModule) -> GradSync:
    reduce = cfg.resolved_grad_reduce()
    if cfg.sync == SYNC_NONE or comm is None:
        return NoSync(comm, reduce)
    if cfg.sync == SYNC_GATHER_SCATTER:
        return GatherScatterSync(comm, flat.grad_views(), reduce)
    if cfg.sync == SYNC_ALL_REDUCE:
        return AllReduceSync(comm, flat.grad_views(), reduce)
    if cfg.sync == SYNC_DDP:
        ints = [b for _, b in flat.module.named_buffers() if not b.dtype.is_floating_point]
        return BucketedSync(comm, flat, reduce, cfg.bucket_cap_mb, cfg.broadcast_buffers, ints,
                            cfg.first_bucket_cap_mb)
    raise ValueError(cfg.sync)


class TorchBackend:
    name = "torch"
    loader_layout = "nchw_fp32"

    def __init__(self, cfg: TrainConfig, device: torch.device, comm: Optional[Communicator]):
        self.cfg = cfg
        self.device = device
        self.comm = comm
        self.model = VGG(cfg.arch, cfg.batch_norm, cfg.num_classes).to(device)
        self.flat = FlatModule(self.model)
        self.flat.params.grad = self.flat.grads
        self.optimizer = torch.optim.SGD([self.flat.params], lr=cfg.lr, momentum=cfg.momentum,
                                         weight_decay=cfg.weight_decay)
        self.sync = build_sync(cfg, comm, self.flat)
        self.bf16 = cfg.dtype == "bf16"

    def _autocast(self):
        if self.bf16:
            return torch.autocast(self.device.type, dtype=torch.bfloat16)
        return contextlib.nullcontext()

    def train_mode(self, on: bool = True) -> None:
        self.model.train(on)

    # ------------------------------------------- communication / compute accounting
    def set_comm_enabled(self, on: bool) -> None:
        """Detach (or re-attach) every collective of the step; timing aid only, ranks drift
        apart while detached."""
        if isinstance(self.sync, BucketedSync):
            self.sync.enabled = on
        elif on:
            self.sync = self._sync_saved
        else:
            self._sync_saved = self.sync
            self.sync = NoSync(self.comm, self.cfg.resolved_grad_reduce())

    def comm_only_step(self) -> None:
        self.sync.comm_only()

    def train_step(self, data: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        self.flat.zero_grad()
        self.sync.prepare_forward()
        self.sync.begin_backward()
        with self._autocast():
            logits = self.model(data)
        loss = F.cross_entropy(logits.float(), target)
        loss.backward()
        self.sync.finish()
        self.optimizer.step()
        return loss.detach()

    @torch.no_grad()
    def eval_batch(self, data: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        with self._autocast():
            logits = self.model(data)
        loss = F.cross_entropy(logits.float(), target)
        correct = (logits.argmax(dim=1) == target).sum()
        return loss, correct

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().clone().contiguous() for k, v in self.model.state_dict().items()}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        self.model.load_state_dict(sd)

    def parameters_flat(self) -> torch.Tensor:
        return self.flat.params

    def close(self) -> None:
        self.sync.close()
