"""Experiment driver, training loop and evaluation loop.

Mirrors the control flow of the reference scripts
(driver ``main``: /root/reference/part1/main.py:80-127, part2/2a/main.py:139-189,
part3/main.py:95-143; ``train_model``: part1/main.py:19-60; ``test_model``:
part1/main.py:62-77) and keeps their console lines byte-for-byte, because scripts
parse them:

    Loss at {k}th batch is {loss}
    Total execution time is : {t} seconds
    Average execution time is  : {t/39} seconds      (one space before ':' in part3)
    Test set: Average loss: {:.4f}, Accuracy: {}/{} ({:.0f}%)

On top of that it prints an images/sec line and, with ``--json``, one JSON record.
"""
from __future__ import annotations

import json
from typing import Dict, Optional

import torch

from ..config import SYNC_NONE, TrainConfig
from ..data.datasets import build_datasets
from ..data.sampler import BatchLoader, ShardedSampler
from ..data.transforms import BatchTransform
from ..models.summary import print_summary
from ..parallel.comm import Communicator
from ..utils import prof
from ..utils.timing import IterationTimer


def resolve_device(cfg: TrainConfig, local_rank: int = 0) -> torch.device:
    want = cfg.device
    if want == "auto":
        want = "cuda" if torch.cuda.is_available() else "cpu"
    if want == "cuda":
        if not torch.cuda.is_available():
            raise RuntimeError("--device cuda requested but no GPU is visible")
        return torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    return torch.device("cpu")


def resolve_backend(cfg: TrainConfig, device: torch.device) -> str:
    if cfg.backend != "auto":
        return cfg.backend
    # a GPU always runs the native kernels (bf16 or fp32 engine, by --dtype); stock torch is the
    # CPU configuration and, on a GPU, an explicit --backend torch (the golden model)
    return "native" if device.type == "cuda" else "torch"


def make_backend(cfg: TrainConfig, device: torch.device, comm: Optional[Communicator]):
    kind = resolve_backend(cfg, device)
    dtype = cfg.dtype
    if dtype == "auto":
        dtype = "bf16" if kind == "native" else "fp32"
    cfg = cfg.replace(dtype=dtype, backend=kind)
    if kind == "native":
        if device.type != "cuda":
            raise RuntimeError("the native backend runs on MI355X GPUs only")
        from .native_backend import NativeBackend
        return NativeBackend(cfg, device, comm)
    from .torch_backend import TorchBackend
    return TorchBackend(cfg, device, comm)


class Trainer:
    def __init__(self, cfg: TrainConfig, device: torch.device, comm: Optional[Communicator]):
        self.cfg = cfg
        self.device = device
        self.comm = comm
        self.rank = comm.rank if comm is not None else 0
        self.world = comm.world_size if comm is not None else 1
        self.backend = make_backend(cfg, device, comm)
        self.cfg = self.backend.cfg
        if cfg.profile_ranges and self.backend.name == "native":
            # per-phase roctx ranges from inside the C++ step (forward / backward / wgrad side
            # stream / buckets / sync / optimizer) in addition to the per-iteration one below
            from ..ops import native
            native.lib().set_trace_ranges(True)
        self.last_timer: Optional[IterationTimer] = None
        self.losses: Dict[int, float] = {}

    # ------------------------------------------------------------------ data
    def build_loaders(self):
        cfg = self.cfg
        train_set, test_set = build_datasets(cfg.synthetic, cfg.data_root, cfg.train_size,
                                             cfg.test_size, cfg.seed)
        train_set, test_set = train_set.to(self.device), test_set.to(self.device)
        layout = self.backend.loader_layout
        sampler = ShardedSampler(len(train_set), rank=self.rank, num_replicas=self.world,
                                 shuffle=False, seed=cfg.seed)
        train_loader = BatchLoader(train_set, cfg.batch_size, sampler,
                                   BatchTransform(True, layout, cfg.seed))
        # every rank evaluates the whole test set (part2/2a/main.py:171-175)
        test_loader = BatchLoader(test_set, cfg.batch_size, None, BatchTransform(False, layout))
        return train_loader, test_loader

    # ----------------------------------------------------------------- train
    def train_epoch(self, loader, epoch: int) -> IterationTimer:
        cfg = self.cfg
        timer = IterationTimer(self.device, skip_first=1)
        if cfg.announce_epoch:
            print(f"Epoch {epoch} triggered")
        self.backend.train_mode(True)
        count_volume = cfg.report_comm_volume and hasattr(self.comm, "reset_volume")
        if count_volume:
            self.comm.reset_volume()
        done = 0
        for batch_idx, (data, target) in enumerate(loader):
            if batch_idx == cfg.iters:
                break
            done += 1
            timer.start()
            with prof.range("train_step", cfg.profile_ranges):
                loss = self.backend.train_step(data, target)
            if (batch_idx + 1) % 20 == 0:
                value = float(loss.item())
                self.losses[batch_idx + 1] = value
                print(f"Loss at {batch_idx+1}th batch is {value}")
            if self.comm is not None and self.world > 1:
                self.comm.guard_sync(cfg.comm_timeout_s)
            timer.stop()
        denom = max(cfg.iters - 1, 1)
        colon = " " * cfg.avg_colon_spaces + ":"
        print(f"Total execution time is : {timer.total} seconds")
        print(f"Average execution time is{colon} {timer.total/denom} seconds")
        if timer.count:
            per_iter = timer.mean()
            global_batch = cfg.batch_size * self.world
            print(f"Throughput: {global_batch / per_iter:.1f} images/sec "
                  f"({cfg.batch_size} img/rank x {self.world} ranks, {per_iter*1e3:.3f} ms/iter)")
        if count_volume:
            self.last_volume = vol = self.comm.volume()
            calls = ", ".join(f"{n} {k}" for k, n in vol["calls"].items() if n)
            print(f"[rank {self.rank}] communication volume over {done} iterations: "
                  f"TX {vol['tx_bytes'] / 1e9:.3f} GB, RX {vol['rx_bytes'] / 1e9:.3f} GB "
                  f"({calls or 'no collectives'})")
        self.last_timer = timer
        return timer

    # ------------------------------------------------------------------ test
    def test(self, loader) -> Dict[str, float]:
        self.backend.train_mode(False)
        loss_sum = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device, dtype=torch.int64)
        batches = 0
        seen = 0
        for batch_idx, (data, target) in enumerate(loader):
            if self.cfg.eval_batches is not None and batch_idx >= self.cfg.eval_batches:
                break
            loss, ok = self.backend.eval_batch(data, target)
            loss_sum += loss.to(loss_sum.dtype)
            correct += ok.to(correct.dtype)
            batches += 1
            seen += int(target.shape[0])
        total = len(loader.dataset) if self.cfg.eval_batches is None else seen
        # mean of per-batch means, divided by the number of batches (part1/main.py:74)
        test_loss = float(loss_sum.item()) / max(batches, 1)
        n_ok = int(correct.item())
        print('Test set: Average loss: {:.4f}, Accuracy: {}/{} ({:.0f}%)\n'.format(
            test_loss, n_ok, total, 100. * n_ok / max(total, 1)))
        return {"test_loss": test_loss, "correct": n_ok, "total": total}

    # ------------------------------------------------------- comm/compute split
    def comm_split(self, loader, batches: int = 4, steps: int = 10):
        """Communication / compute split and overlap fraction of this run's sync mode (the
        reference derived it from separate 1-node and N-node runs, group25.pdf p10).  Runs
        last: parameters keep training and ranks drift apart during the measurement."""
        from ..utils.overlap import format_comm_split, measure_comm_split
        if self.comm is None or self.world < 1 or self.cfg.sync == SYNC_NONE:
            return None
        held = []
        for i, batch in enumerate(loader):
            if i == batches:
                break
            held.append(batch)
        self.backend.train_mode(True)
        split = measure_comm_split(self.backend, self.comm, held, n=steps)
        if split is not None:
            print(format_comm_split(self.cfg.sync, split))
        return split

    # ------------------------------------------------------------------- run
    def run(self) -> Dict[str, object]:
        cfg = self.cfg
        train_loader, test_loader = self.build_loaders()
        if cfg.summary:
            print_summary(cfg.arch, cfg.batch_norm)
        if cfg.checkpoint_in:
            from ..models.checkpoint import load_checkpoint
            self.backend.load_state_dict(load_checkpoint(cfg.checkpoint_in))
        result: Dict[str, object] = {}
        for epoch in range(cfg.epochs):
            timer = self.train_epoch(train_loader, epoch)
            result["avg_s_per_iter"] = timer.total / max(cfg.iters - 1, 1)
            result["device_ms_per_iter"] = timer.mean_device_ms()
            if cfg.eval:
                result.update(self.test(test_loader))
        result["losses"] = dict(self.losses)
        if cfg.checkpoint_out and self.rank == 0:
            from ..models.checkpoint import save_checkpoint
            save_checkpoint(cfg.checkpoint_out, self.backend.state_dict(),
                            ddp_prefix=cfg.sync == "ddp")
        if getattr(self, "last_volume", None) is not None:
            result["comm_volume"] = self.last_volume
        if cfg.report_overlap:
            split = self.comm_split(train_loader)
            if split is not None:
                result["comm_split"] = split
        if cfg.json and self.rank == 0:
            rec = {"sync": cfg.sync, "world_size": self.world, "batch_size": cfg.batch_size,
                   "backend": self.backend.name, "dtype": cfg.dtype,
                   "batch_norm": cfg.batch_norm, **result}
            if result.get("avg_s_per_iter"):
                rec["images_per_sec"] = cfg.batch_size * self.world / result["avg_s_per_iter"]
            print(json.dumps(rec))
        return result

    def close(self) -> None:
        self.backend.close()
