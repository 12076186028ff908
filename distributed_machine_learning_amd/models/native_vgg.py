"""VGG as an ``nn.Module`` whose layers run the native HIP kernels under autograd.

Same module tree, parameter shapes, initialisation and ``state_dict`` as the stock model
(/root/reference/part1/model.py:11-46; BatchNorm per /root/reference/part3/model.py:24):
every layer subclasses its torch counterpart and only replaces ``forward``, so a
checkpoint moves freely between :class:`VGG`, :class:`NativeVGG` and the C++ engine.  ReLU
is fused into the producing layer (conv epilogue, or the BatchNorm kernel); its slot in the
``Sequential`` is kept as a pass-through so the indices in the keys do not shift.

Input: bf16 ``[B, 32, 32, 4]`` (NHWC, zero fourth channel) as produced by
``BatchTransform(layout="nhwc4_bf16")``; output: fp32 logits.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..ops import functional as NF
from .vgg import FC_IN, layer_plan


class Conv3x3(nn.Conv2d):
    def __init__(self, cin: int, cout: int, fuse_relu: bool):
        super().__init__(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
        self.fuse_relu = fuse_relu

    def forward(self, x):
        return NF.conv3x3(x, self.weight, self.bias, self.fuse_relu)


class BatchNormReLU(nn.BatchNorm2d):
    def forward(self, x):
        if self.training:
            self.num_batches_tracked += 1
        return NF.batch_norm_relu(x, self.weight, self.bias, self.running_mean, self.running_var,
                                  self.training, self.momentum, self.eps)


class FusedReLU(nn.Module):
    """Place-holder for the reference's nn.ReLU slot; the activation already happened."""

    def forward(self, x):
        return x


class MaxPool2x2(nn.Module):
    def forward(self, x):
        return NF.max_pool2x2(x)


class Classifier(nn.Linear):
    def forward(self, x):
        return NF.linear(x.reshape(x.shape[0], -1), self.weight, self.bias)


class NativeVGG(nn.Module):
    def __init__(self, arch: str = "VGG11", batch_norm: bool = False, num_classes: int = 10):
        super().__init__()
        self.arch, self.batch_norm = arch, batch_norm
        self.plan = layer_plan(arch, batch_norm, num_classes)
        mods = []
        for s in self.plan:
            if s.kind == "conv":
                mods.append(Conv3x3(s.cin, s.cout, fuse_relu=not batch_norm))
            elif s.kind == "bn":
                mods.append(BatchNormReLU(s.cout))
            elif s.kind == "relu":
                mods.append(FusedReLU())
            elif s.kind == "pool":
                mods.append(MaxPool2x2())
        self.layers = nn.Sequential(*mods)
        fc = self.plan[-1]
        if fc.cin != FC_IN:
            raise ValueError(f"{arch} flattens to {fc.cin}, expected {FC_IN}")
        self.fc1 = Classifier(fc.cin, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc1(self.layers(x))


def native_vgg11(batch_norm: bool = False) -> NativeVGG:
    return NativeVGG("VGG11", batch_norm)
