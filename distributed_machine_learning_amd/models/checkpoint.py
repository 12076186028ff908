"""VGG ``state_dict`` checkpoints.

The reference never saves a model; the contract is the ``nn.Module`` layout its
``model.py`` implies (/root/reference/part1/model.py:38-40): conv weights OIHW fp32
under ``layers.<i>.weight``, ``fc1.weight`` as ``[10, 512]``, BatchNorm entries with
running statistics for the part3 variant (/root/reference/part3/model.py:24), and a
``module.`` prefix on every key when the model was wrapped by DDP
(/root/reference/part3/main.py:137).  Checkpoints written here load into a plain
``torch.nn`` VGG and vice versa, whichever backend produced them; files are written
with ``torch.save`` of a flat ``{key: tensor}`` dict and read with
``weights_only=True``.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .vgg import layer_plan

DDP_PREFIX = "module."


def expected_keys(arch: str = "VGG11", batch_norm: bool = False) -> List[str]:
    keys: List[str] = []
    for s in layer_plan(arch, batch_norm):
        if s.kind == "conv" or s.kind == "fc":
            keys += [f"{s.key}.weight", f"{s.key}.bias"]
        elif s.kind == "bn":
            keys += [f"{s.key}.{leaf}" for leaf in
                     ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    return keys


def expected_shapes(arch: str = "VGG11", batch_norm: bool = False) -> Dict[str, tuple]:
    shapes: Dict[str, tuple] = {}
    for s in layer_plan(arch, batch_norm):
        if s.kind == "conv":
            shapes[f"{s.key}.weight"] = (s.cout, s.cin, 3, 3)
            shapes[f"{s.key}.bias"] = (s.cout,)
        elif s.kind == "fc":
            shapes[f"{s.key}.weight"] = (s.cout, s.cin)
            shapes[f"{s.key}.bias"] = (s.cout,)
        elif s.kind == "bn":
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"{s.key}.{leaf}"] = (s.cout,)
            shapes[f"{s.key}.num_batches_tracked"] = ()
    return shapes


def strip_ddp_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    if sd and all(k.startswith(DDP_PREFIX) for k in sd):
        return {k[len(DDP_PREFIX):]: v for k, v in sd.items()}
    return dict(sd)


def add_ddp_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {DDP_PREFIX + k: v for k, v in sd.items()}


def validate_state_dict(sd: Dict[str, torch.Tensor], arch: str = "VGG11",
                        batch_norm: bool = False) -> None:
    shapes = expected_shapes(arch, batch_norm)
    missing = [k for k in shapes if k not in sd]
    extra = [k for k in sd if k not in shapes]
    if missing or extra:
        raise KeyError(f"state_dict mismatch: missing {missing}, unexpected {extra}")
    for k, shape in shapes.items():
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shape}")


def save_checkpoint(path: str, sd: Dict[str, torch.Tensor], ddp_prefix: bool = False) -> None:
    out = {k: v.detach().to("cpu").contiguous() for k, v in sd.items()}
    for k, v in out.items():
        if v.dtype.is_floating_point and v.dtype != torch.float32:
            out[k] = v.float()
    if ddp_prefix:
        out = add_ddp_prefix(out)
    torch.save(out, path)


def load_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, dict):
        raise TypeError(f"{path}: expected a dict of tensors")
    return strip_ddp_prefix(sd)
