"""VGG family for 3x32x32 inputs.

Module tree and therefore ``state_dict`` keys follow the reference
(/root/reference/part1/model.py:3-50; BatchNorm variant /root/reference/part3/model.py:24):
``layers.<i>.{weight,bias}`` for the conv (and BN) entries of one ``nn.Sequential``
plus ``fc1.{weight,bias}``.  A layer *plan* (`LayerSpec` list) is derived from the
same table so the native engine, the summary printer and the checkpoint converter
all walk one description of the network.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import nn

# Width tables: an int is a 3x3 conv with that many output channels, "M" is a 2x2 max-pool.
VGG_WIDTHS = {
    "VGG11": (64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"),
    "VGG13": (64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"),
    "VGG16": (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M",
              512, 512, 512, "M"),
    "VGG19": (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
              512, 512, 512, 512, "M"),
}
IN_CHANNELS = 3
IMAGE_HW = 32
FC_IN = 512


@dataclass(frozen=True)
class LayerSpec:
    """One entry of ``layers`` (or the classifier) with its place in the Sequential."""

    kind: str                 # "conv" | "bn" | "relu" | "pool" | "fc"
    index: Optional[int]      # position inside ``layers`` (None for fc1)
    cin: int = 0
    cout: int = 0
    hw: int = 0               # input spatial size (square)

    @property
    def key(self) -> str:
        return "fc1" if self.kind == "fc" else f"layers.{self.index}"


def layer_plan(arch: str = "VGG11", batch_norm: bool = False,
               num_classes: int = 10) -> List[LayerSpec]:
    """Flatten a width table into the exact module order of the Sequential."""
    widths = VGG_WIDTHS[arch]
    plan: List[LayerSpec] = []
    cin, hw, idx = IN_CHANNELS, IMAGE_HW, 0
    for w in widths:
        if w == "M":
            plan.append(LayerSpec("pool", idx, cin, cin, hw))
            hw //= 2
            idx += 1
            continue
        plan.append(LayerSpec("conv", idx, cin, int(w), hw))
        idx += 1
        if batch_norm:
            plan.append(LayerSpec("bn", idx, int(w), int(w), hw))
            idx += 1
        plan.append(LayerSpec("relu", idx, int(w), int(w), hw))
        idx += 1
        cin = int(w)
    plan.append(LayerSpec("fc", None, cin * hw * hw, num_classes, hw))
    return plan


def _module_for(spec: LayerSpec) -> nn.Module:
    if spec.kind == "conv":
        return nn.Conv2d(spec.cin, spec.cout, kernel_size=3, stride=1, padding=1, bias=True)
    if spec.kind == "bn":
        return nn.BatchNorm2d(spec.cout)
    if spec.kind == "relu":
        return nn.ReLU(inplace=True)
    if spec.kind == "pool":
        return nn.MaxPool2d(kernel_size=2, stride=2)
    raise ValueError(spec.kind)


class VGG(nn.Module):
    """Stock-torch VGG: the golden model every native kernel is checked against,
    and the model the CPU configuration trains."""

    def __init__(self, arch: str = "VGG11", batch_norm: bool = False, num_classes: int = 10):
        super().__init__()
        self.arch = arch
        self.batch_norm = batch_norm
        self.plan = layer_plan(arch, batch_norm, num_classes)
        body = [s for s in self.plan if s.kind != "fc"]
        self.layers = nn.Sequential(*[_module_for(s) for s in body])
        fc = self.plan[-1]
        if fc.cin != FC_IN:
            raise ValueError(f"{arch} flattens to {fc.cin}, expected {FC_IN}")
        self.fc1 = nn.Linear(fc.cin, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        feat = self.layers(x)
        return self.fc1(feat.flatten(1))


def VGG11(batch_norm: bool = False) -> VGG:
    return VGG("VGG11", batch_norm=batch_norm)


def VGG13(batch_norm: bool = False) -> VGG:
    return VGG("VGG13", batch_norm=batch_norm)


def VGG16(batch_norm: bool = False) -> VGG:
    return VGG("VGG16", batch_norm=batch_norm)


def VGG19(batch_norm: bool = False) -> VGG:
    return VGG("VGG19", batch_norm=batch_norm)


def count_parameters(arch: str = "VGG11", batch_norm: bool = False) -> int:
    total = 0
    for s in layer_plan(arch, batch_norm):
        if s.kind == "conv":
            total += s.cout * s.cin * 9 + s.cout
        elif s.kind == "bn":
            total += 2 * s.cout
        elif s.kind == "fc":
            total += s.cin * s.cout + s.cout
    return total


def parameter_keys(arch: str = "VGG11", batch_norm: bool = False) -> List[str]:
    """``named_parameters`` order: conv/bn weight+bias per layer, then fc1."""
    keys: List[str] = []
    for s in layer_plan(arch, batch_norm):
        if s.kind in ("conv", "bn", "fc"):
            keys += [f"{s.key}.weight", f"{s.key}.bias"]
    return keys
