"""CIFAR-10-shaped datasets: a deterministic synthetic generator and a reader for
the real CIFAR-10 *binary* distribution (``data_batch_N.bin`` / ``test_batch.bin``).

The reference loads ``torchvision.datasets.CIFAR10`` (50 000 train / 10 000 test,
/root/reference/part1/main.py:96-97,107-108).  torchvision and the dataset are not
available offline, so the default is synthetic uint8 images of the same shape and
label range; ``CifarBinary`` is the opt-in real-data path and needs no unpickling.

Images are held as one uint8 tensor ``[N, 32, 32, 3]`` (HWC, like the PIL images
torchvision hands to its transforms) plus an int64 label vector; batches are
assembled and augmented on the training device by ``transforms.BatchTransform``.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np
import torch

IMAGE_SHAPE = (32, 32, 3)
RECORD_BYTES = 1 + 3 * 32 * 32


class ArrayDataset:
    """uint8 HWC images + int64 labels resident in host or device memory."""

    def __init__(self, images: torch.Tensor, labels: torch.Tensor):
        assert images.dtype == torch.uint8 and images.shape[1:] == IMAGE_SHAPE
        assert labels.dtype == torch.int64 and labels.shape[0] == images.shape[0]
        self.images = images
        self.labels = labels

    def __len__(self) -> int:
        return int(self.images.shape[0])

    def to(self, device) -> "ArrayDataset":
        return ArrayDataset(self.images.to(device), self.labels.to(device))

    def take(self, index: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        index = index.to(self.images.device)
        return self.images.index_select(0, index), self.labels.index_select(0, index)


def synthetic_cifar(n: int, seed: int, num_classes: int = 10) -> ArrayDataset:
    """``n`` random images; identical for every rank given the same seed so the
    rank-strided sampler reproduces the reference's "same data for all the tasks"
    property (/root/reference/part2/2a/main.py:161)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    images = torch.randint(0, 256, (n,) + IMAGE_SHAPE, dtype=torch.uint8, generator=gen)
    labels = torch.randint(0, num_classes, (n,), dtype=torch.int64, generator=gen)
    return ArrayDataset(images, labels)


class CifarBinary:
    """Reader for ``cifar-10-batches-bin``: each record is 1 label byte followed by
    3072 bytes in CHW order."""

    TRAIN_FILES = tuple(f"data_batch_{i}.bin" for i in range(1, 6))
    TEST_FILES = ("test_batch.bin",)

    @staticmethod
    def available(root: str) -> bool:
        d = os.path.join(root, "cifar-10-batches-bin")
        return all(os.path.exists(os.path.join(d, f)) for f in CifarBinary.TRAIN_FILES)

    @staticmethod
    def load(root: str, train: bool) -> ArrayDataset:
        d = os.path.join(root, "cifar-10-batches-bin")
        names = CifarBinary.TRAIN_FILES if train else CifarBinary.TEST_FILES
        chunks = []
        for name in names:
            raw = np.fromfile(os.path.join(d, name), dtype=np.uint8)
            if raw.size % RECORD_BYTES:
                raise ValueError(f"{name}: size {raw.size} is not a multiple of {RECORD_BYTES}")
            chunks.append(raw.reshape(-1, RECORD_BYTES))
        rec = np.concatenate(chunks, axis=0)
        labels = torch.from_numpy(rec[:, 0].astype(np.int64))
        chw = rec[:, 1:].reshape(-1, 3, 32, 32)
        images = torch.from_numpy(np.ascontiguousarray(chw.transpose(0, 2, 3, 1)))
        return ArrayDataset(images, labels)


def build_datasets(synthetic: bool, data_root: str, train_size: int, test_size: int,
                   seed: int) -> Tuple[ArrayDataset, ArrayDataset]:
    if not synthetic:
        if not CifarBinary.available(data_root):
            raise FileNotFoundError(
                f"real CIFAR-10 requested but {data_root}/cifar-10-batches-bin is missing; "
                "there is no network here, use --synthetic")
        return CifarBinary.load(data_root, True), CifarBinary.load(data_root, False)
    return synthetic_cifar(train_size, seed), synthetic_cifar(test_size, seed + 1)
