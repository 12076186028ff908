"""Iteration timing.

The reference brackets ``zero_grad .. optimizer.step`` with
``datetime.now().timestamp()`` and skips iteration 0
(/root/reference/part1/main.py:36,53-55).  GPU work is asynchronous, so a wall-clock
read is only meaningful after the device has drained: ``IterationTimer`` synchronises
the device at both ends of the bracket and also records a HIP event pair so the
device-side duration of the same region is available.
"""
from __future__ import annotations

import time
from typing import List, Optional

import torch


class IterationTimer:
    def __init__(self, device: torch.device, skip_first: int = 1):
        self.cuda = device.type == "cuda"
        self.skip_first = skip_first
        self.wall: List[float] = []
        self.device_ms: List[float] = []
        self._t0 = 0.0
        self._e0 = self._e1 = None
        self._n = 0

    def start(self) -> None:
        if self.cuda:
            torch.cuda.synchronize()
            self._e0 = torch.cuda.Event(enable_timing=True)
            self._e1 = torch.cuda.Event(enable_timing=True)
            self._e0.record()
        self._t0 = time.perf_counter()

    def stop(self) -> float:
        if self.cuda:
            self._e1.record()
            torch.cuda.synchronize()
        dt = time.perf_counter() - self._t0
        if self._n >= self.skip_first:
            self.wall.append(dt)
            if self.cuda:
                self.device_ms.append(self._e0.elapsed_time(self._e1))
        self._n += 1
        return dt

    @property
    def total(self) -> float:
        return float(sum(self.wall))

    @property
    def count(self) -> int:
        return len(self.wall)

    def mean(self) -> float:
        return self.total / max(self.count, 1)

    def mean_device_ms(self) -> Optional[float]:
        if not self.device_ms:
            return None
        return float(sum(self.device_ms)) / len(self.device_ms)
