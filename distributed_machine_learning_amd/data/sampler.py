"""Rank-strided sharding and batching.

Equivalent of ``DistributedSampler(rank, num_replicas, shuffle=False)`` followed by a
``DataLoader(batch_size, shuffle=False)`` as the reference wires them
(/root/reference/part2/2a/main.py:158-167): the index list is padded by wrapping
around until it divides evenly, rank r then owns indices r, r+R, r+2R, ...; with
R ranks of batch B the union of every rank's batch i is the single-process batch i
of size R*B.
"""
from __future__ import annotations

import math
from typing import Iterator, Optional, Tuple

import torch

from .datasets import ArrayDataset


class ShardedSampler:
    def __init__(self, length: int, rank: int = 0, num_replicas: int = 1,
                 shuffle: bool = False, seed: int = 0):
        if not 0 <= rank < num_replicas:
            raise ValueError(f"rank {rank} outside [0, {num_replicas})")
        self.length = length
        self.rank = rank
        self.num_replicas = num_replicas
        self.shuffle = shuffle
        self.seed = seed
        self.epoch = 0
        self.num_samples = math.ceil(length / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def indices(self) -> torch.Tensor:
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.length, generator=g)
        else:
            order = torch.arange(self.length)
        pad = self.total_size - self.length
        if pad > 0:
            reps = math.ceil(pad / max(self.length, 1))
            order = torch.cat([order, order.repeat(reps)[:pad]])
        return order[self.rank:self.total_size:self.num_replicas]

    def __len__(self) -> int:
        return self.num_samples


class BatchLoader:
    """Iterates ``(data, target)`` batches.  ``transform`` maps a uint8 HWC batch to
    whatever tensor the model consumes; it runs on the dataset's device."""

    def __init__(self, dataset: ArrayDataset, batch_size: int,
                 sampler: Optional[ShardedSampler] = None, transform=None):
        self.dataset = dataset
        self.batch_size = batch_size
        self.sampler = sampler or ShardedSampler(len(dataset))
        self.transform = transform

    def __len__(self) -> int:
        return math.ceil(len(self.sampler) / self.batch_size)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        idx = self.sampler.indices()
        for start in range(0, idx.numel(), self.batch_size):
            batch_idx = idx[start:start + self.batch_size]
            images, labels = self.dataset.take(batch_idx)
            if self.transform is not None:
                images = self.transform(images, batch_idx)
            yield images, labels
