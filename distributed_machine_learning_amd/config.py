"""Run configuration for the four training modes.

The reference keeps its knobs as module constants and three argparse flags
(/root/reference/part1/main.py:15-18, part2/2a/main.py:27-33,210-216,
part3/main.py:27-31,165-171).  Here they are one dataclass; each ``partN``
wrapper pins the reference's values and the unified CLI can override any of them.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Optional

SEED = 69143  # part1/main.py:17, part2/2a/main.py:199, sampler seed part2/2a/main.py:159

# sync strategy names
SYNC_NONE = "none"
SYNC_GATHER_SCATTER = "gather_scatter"
SYNC_ALL_REDUCE = "all_reduce"
SYNC_DDP = "ddp"
SYNC_MODES = (SYNC_NONE, SYNC_GATHER_SCATTER, SYNC_ALL_REDUCE, SYNC_DDP)

# CIFAR-10 channel statistics used by every variant (part1/main.py:82-83)
CIFAR_MEAN = tuple(v / 255.0 for v in (125.3, 123.0, 113.9))
CIFAR_STD = tuple(v / 255.0 for v in (63.0, 62.1, 66.7))


@dataclass
class TrainConfig:
    """Everything one training process needs to know."""

    # --- distribution (reference CLI: --master-ip --rank --num-nodes) ---
    master_ip: str = "127.0.1.1:8000"
    rank: int = 0
    num_nodes: int = 1
    sync: str = SYNC_NONE
    grad_reduce: Optional[str] = None      # "sum" | "mean"; None -> the mode's reference semantics
    bucket_cap_mb: float = 25.0            # part3/main.py:137
    first_bucket_cap_mb: float = 1.0       # DistributedDataParallel closes its first bucket early
    tail_bucket_cap_mb: float = 0.0        # > 0: cap the LAST bucket (its all-reduce is exposed)
    broadcast_buffers: bool = True         # DDP default relied on by part3
    comm_timeout_s: float = 300.0
    serialize_streams: bool = False        # debug: run comm on the compute stream
    grad_comm_dtype: str = "fp32"          # DDP bucket payload on the wire: "fp32" | "bf16"
    inline_last_bucket: Optional[bool] = None   # None: on the compute stream only with one rank

    # --- model ---
    arch: str = "VGG11"
    batch_norm: bool = False               # part3/model.py:24 enables it
    num_classes: int = 10

    # --- optimisation (part1/main.py:120-121) ---
    batch_size: int = 256
    lr: float = 0.1
    momentum: float = 0.9
    weight_decay: float = 1e-4
    iters: int = 40                        # part1/main.py:32
    epochs: int = 1
    seed: int = SEED

    # --- execution ---
    device: str = "auto"                   # "auto" | "cpu" | "cuda"
    dtype: str = "auto"                    # "auto" | "fp32" | "bf16"
    backend: str = "auto"                  # "auto" | "torch" | "native"
    threads: int = 4                       # torch.set_num_threads(4), part1/main.py:16
    use_graph: bool = False                # capture the native train step in a hipGraph
    wgrad_overlap: bool = True             # weight-gradient kernels on a second HIP stream
    fused_optimizer: bool = False          # SGD slices inside backward (needs wgrad_overlap)
    # first conv layer whose update (with everything after it) runs on the side stream next to
    # the following forward's first layers; 0 = off, -1 = from the first 4x4 layer on
    update_overlap: int = 0
    wgrad_group: int = 0                   # layers per hand-off to that stream; 0 = auto

    # --- data ---
    synthetic: bool = True
    data_root: str = "./data"
    train_size: int = 50000
    test_size: int = 10000
    eval: bool = True
    eval_batches: Optional[int] = None     # None -> the whole test set

    # --- output ---
    summary: bool = True
    avg_colon_spaces: int = 1              # part3 prints "time is :", 2a/2b print "time is  :"
    announce_epoch: bool = False           # part3/main.py:43
    json: bool = False
    profile_ranges: bool = False
    report_overlap: bool = False           # after the run: comm/compute split of a step
    report_comm_volume: bool = False       # after the epoch: bytes this rank sent / received
    checkpoint_out: Optional[str] = None
    checkpoint_in: Optional[str] = None

    def resolved_grad_reduce(self) -> str:
        """SUM for the two manual modes, MEAN for the DDP mode, as the reference does
        (part2/2a/main.py:104-109, part2/2b/main.py:106, DDP averaging in part3)."""
        if self.grad_reduce is not None:
            return self.grad_reduce
        return "mean" if self.sync == SYNC_DDP else "sum"

    def replace(self, **kw) -> "TrainConfig":
        return dataclasses.replace(self, **kw)


def part_defaults(part: str) -> TrainConfig:
    """The constants each reference script hard-codes."""
    if part == "part1":
        return TrainConfig(sync=SYNC_NONE, batch_size=256, batch_norm=False,
                           avg_colon_spaces=2)
    if part == "part2a":
        return TrainConfig(sync=SYNC_GATHER_SCATTER, batch_size=64, batch_norm=False,
                           avg_colon_spaces=2)
    if part == "part2b":
        return TrainConfig(sync=SYNC_ALL_REDUCE, batch_size=64, batch_norm=False,
                           avg_colon_spaces=2)
    if part == "part3":
        return TrainConfig(sync=SYNC_DDP, batch_size=64, batch_norm=True,
                           avg_colon_spaces=1, announce_epoch=True)
    raise ValueError(f"unknown part {part!r}")
