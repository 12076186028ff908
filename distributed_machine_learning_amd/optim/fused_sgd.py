"""SGD with momentum and weight decay as ONE kernel launch over a flat parameter buffer.

Semantics of ``torch.optim.SGD(lr, momentum, weight_decay)`` with dampening 0 and no
Nesterov (/root/reference/part1/main.py:120-121): ``g += wd * p; buf = mu * buf + g
(first step: buf = g); p -= lr * buf``.  The module's parameters and gradients are
re-homed into contiguous fp32 buffers (:class:`..parallel.flat.FlatModule`), which is also
the layout the gradient-sync strategies communicate from.
"""
from __future__ import annotations

import torch

from ..ops import native
from ..parallel.flat import FlatModule


class FusedSGD:
    def __init__(self, module_or_flat, lr: float = 0.1, momentum: float = 0.9,
                 weight_decay: float = 1e-4):
        self.flat = module_or_flat if isinstance(module_or_flat, FlatModule) \
            else FlatModule(module_or_flat)
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.buf = torch.zeros_like(self.flat.params)
        self.steps = 0

    def zero_grad(self) -> None:
        self.flat.grads.zero_()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> None:
        p, g = self.flat.params, self.flat.grads
        if p.is_cuda:
            # zero momentum makes "buf = mu*buf + g" equal "buf = g" on the first step too
            native.sgd_momentum_(p, g, self.buf, self.lr, self.momentum, self.weight_decay,
                                 grad_scale, None, False)
        else:
            g = g * grad_scale + self.weight_decay * p
            self.buf.mul_(self.momentum).add_(g)
            p.sub_(self.lr * self.buf)
        self.steps += 1
