"""Batch-level CIFAR transforms.

The reference composes per-image torchvision transforms on the host
(/root/reference/part1/main.py:84-93): ``RandomCrop(32, padding=4)``,
``RandomHorizontalFlip()``, ``ToTensor()``, ``Normalize(mean, std)`` for training and
the last two for testing.  Here one call transforms a whole uint8 HWC batch on the
training device: crop offsets and flip bits are drawn per image, then a single
gather (torch path) or one fused HIP kernel (native path) produces the normalised
batch.  Zero padding is applied to the raw pixels before normalisation, exactly as
the per-image pipeline does, so a padded pixel becomes ``(0 - mean) / std``.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..config import CIFAR_MEAN, CIFAR_STD

PAD = 4
HW = 32


def _mix(x: torch.Tensor) -> torch.Tensor:
    """Counter-based integer hash (xorshift-multiply rounds on int64, kept to 31 bits)."""
    m = (1 << 31) - 1
    x = (x ^ (x >> 15)) * 0x2C1B3C6D & m
    x = (x ^ (x >> 12)) * 0x297A2D39 & m
    return (x ^ (x >> 15)) & m


def draw_augmentation(index: torch.Tensor, seed: int, epoch: int = 0) -> torch.Tensor:
    """``[n, 3]`` int32: crop row offset, crop column offset (0..8) and flip bit.

    The draw is a pure function of (seed, epoch, dataset index), not of the rank or
    the order batches are fetched in, so R ranks of batch B see exactly the augmented
    images a single process of batch R*B sees."""
    base = index.to(torch.int64) * 3 + (int(seed) * 1000003 + int(epoch) * 7919) % (1 << 30)
    oy = _mix(base) % (2 * PAD + 1)
    ox = _mix(base + 1) % (2 * PAD + 1)
    flip = _mix(base + 2) & 1
    return torch.stack([oy, ox, flip], dim=1).to(torch.int32)


def _augment_reference(images_u8: torch.Tensor, aug: Optional[torch.Tensor]) -> torch.Tensor:
    """uint8 ``[N,32,32,3]`` -> float32 ``[N,3,32,32]`` normalised (plain torch ops)."""
    n = images_u8.shape[0]
    dev = images_u8.device
    x = images_u8.to(torch.float32) / 255.0
    if aug is not None:
        ar = torch.arange(HW, device=dev)
        oy = aug[:, 0].long().view(n, 1)
        ox = aug[:, 1].long().view(n, 1)
        flip = aug[:, 2].bool().view(n, 1)
        src_y = ar.view(1, HW) + oy - PAD                                  # [N,32]
        col = torch.where(flip, (HW - 1) - ar.view(1, HW), ar.view(1, HW))
        src_x = col + ox - PAD                                             # [N,32]
        valid = ((src_y >= 0) & (src_y < HW)).view(n, HW, 1) & \
                ((src_x >= 0) & (src_x < HW)).view(n, 1, HW)
        sy = src_y.clamp(0, HW - 1).view(n, HW, 1).expand(n, HW, HW)
        sx = src_x.clamp(0, HW - 1).view(n, 1, HW).expand(n, HW, HW)
        bi = torch.arange(n, device=dev).view(n, 1, 1).expand(n, HW, HW)
        x = x[bi, sy, sx] * valid.unsqueeze(-1).to(x.dtype)
    mean = torch.tensor(CIFAR_MEAN, device=dev, dtype=torch.float32)
    std = torch.tensor(CIFAR_STD, device=dev, dtype=torch.float32)
    x = (x - mean) / std
    return x.permute(0, 3, 1, 2).contiguous()


class BatchTransform:
    """Callable handed to ``BatchLoader``.

    ``layout="nchw_fp32"`` feeds the stock-torch model; ``layout="nhwc4_bf16"`` /
    ``"nhwc4_fp32"`` produce the channel-padded NHWC batch the native first-layer kernels
    read (bf16 or fp32 engine) and run the fused HIP kernel when the batch lives on a GPU.
    """

    def __init__(self, train: bool, layout: str = "nchw_fp32", seed: int = 0):
        if layout not in ("nchw_fp32", "nhwc4_bf16", "nhwc4_fp32"):
            raise ValueError(layout)
        self.train = train
        self.layout = layout
        self.seed = seed
        self.epoch = 0

    def __call__(self, images_u8: torch.Tensor, index: Optional[torch.Tensor] = None) -> torch.Tensor:
        aug = None
        if self.train:
            if index is None:
                index = torch.arange(images_u8.shape[0])
            aug = draw_augmentation(index.to(images_u8.device), self.seed, self.epoch)
        if self.layout == "nchw_fp32":
            return _augment_reference(images_u8, aug)
        dtype = torch.bfloat16 if self.layout == "nhwc4_bf16" else torch.float32
        if images_u8.is_cuda:
            from ..ops import native
            return native.augment_normalize(images_u8, aug, dtype)
        x = _augment_reference(images_u8, aug).permute(0, 2, 3, 1)
        out = torch.zeros(x.shape[0], HW, HW, 4, dtype=dtype, device=x.device)
        out[..., :3] = x.to(dtype)
        return out
