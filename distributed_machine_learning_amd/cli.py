"""Command line and process entry point.

Keeps the reference's three flags with their names and defaults
(``--master-ip 127.0.1.1:8000 --rank 0 --num-nodes 1``:
/root/reference/part2/2a/main.py:210-218, part2/2b/main.py:179-187,
part3/main.py:165-173) and its start-up / shut-down sequence (``fun_main``:
part2/2a/main.py:194-207): print the rendezvous line, join the group, seed, print the
identity line, run, leave the group.  Everything else is opt-in.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Optional, Sequence

import torch

from .config import SYNC_MODES, SYNC_NONE, TrainConfig, part_defaults


def build_parser(defaults: TrainConfig, distributed_flags: bool = True) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="MI355X-native data-parallel VGG training")
    if distributed_flags:
        p.add_argument("--master-ip", type=str, default=defaults.master_ip)
        p.add_argument("--rank", type=int, default=defaults.rank)
        p.add_argument("--num-nodes", type=int, default=defaults.num_nodes)
    p.add_argument("--sync", choices=SYNC_MODES, default=defaults.sync)
    p.add_argument("--grad-reduce", choices=("sum", "mean"), default=defaults.grad_reduce)
    p.add_argument("--bucket-cap-mb", type=float, default=defaults.bucket_cap_mb)
    p.add_argument("--first-bucket-cap-mb", type=float, default=defaults.first_bucket_cap_mb)
    p.add_argument("--tail-bucket-cap-mb", type=float, default=defaults.tail_bucket_cap_mb)
    p.add_argument("--batch-size", type=int, default=defaults.batch_size)
    p.add_argument("--iters", type=int, default=defaults.iters)
    p.add_argument("--arch", choices=("VGG11", "VGG13", "VGG16", "VGG19"), default=defaults.arch)
    bn = p.add_mutually_exclusive_group()
    bn.add_argument("--batch-norm", dest="batch_norm", action="store_true",
                    default=defaults.batch_norm)
    bn.add_argument("--no-batch-norm", dest="batch_norm", action="store_false")
    p.add_argument("--dtype", choices=("auto", "fp32", "bf16"), default=defaults.dtype)
    p.add_argument("--device", choices=("auto", "cpu", "cuda"), default=defaults.device)
    p.add_argument("--backend", choices=("auto", "torch", "native"), default=defaults.backend)
    p.add_argument("--seed", type=int, default=defaults.seed)
    p.add_argument("--lr", type=float, default=defaults.lr)
    p.add_argument("--momentum", type=float, default=defaults.momentum)
    p.add_argument("--weight-decay", type=float, default=defaults.weight_decay)
    syn = p.add_mutually_exclusive_group()
    syn.add_argument("--synthetic", dest="synthetic", action="store_true",
                     default=defaults.synthetic)
    syn.add_argument("--real-data", dest="synthetic", action="store_false")
    p.add_argument("--data-root", type=str, default=defaults.data_root)
    p.add_argument("--train-size", type=int, default=defaults.train_size)
    p.add_argument("--test-size", type=int, default=defaults.test_size)
    p.add_argument("--no-eval", dest="eval", action="store_false", default=defaults.eval)
    p.add_argument("--eval-batches", type=int, default=defaults.eval_batches)
    p.add_argument("--no-summary", dest="summary", action="store_false", default=defaults.summary)
    p.add_argument("--json", action="store_true", default=defaults.json)
    p.add_argument("--graph", dest="use_graph", action="store_true", default=defaults.use_graph)
    wo = p.add_mutually_exclusive_group()
    wo.add_argument("--wgrad-overlap", dest="wgrad_overlap", action="store_true",
                    default=defaults.wgrad_overlap,
                    help="run backward's weight-gradient kernels on a second HIP stream")
    wo.add_argument("--no-wgrad-overlap", dest="wgrad_overlap", action="store_false")
    p.add_argument("--wgrad-group", type=int, default=defaults.wgrad_group)
    p.add_argument("--update-overlap", dest="update_overlap", type=int,
                   default=defaults.update_overlap,
                   help="first conv layer whose SGD update (with all later layers) runs on the side "
                        "stream next to the next forward's first layers (0 off, -1 auto)")
    p.add_argument("--fused-optimizer", dest="fused_optimizer", action="store_true",
                   default=defaults.fused_optimizer,
                   help="optimizer update in slices inside backward (behind each group's sums / "
                        "each bucket's all-reduce) instead of one launch after it")
    p.add_argument("--serialize-streams", action="store_true",
                   default=defaults.serialize_streams,
                   help="debug: issue collectives on the compute stream (no overlap)")
    p.add_argument("--grad-comm-dtype", choices=("fp32", "bf16"), default=defaults.grad_comm_dtype,
                   help="ddp mode: element type of the gradient buckets on the wire")
    p.add_argument("--inline-last-bucket", dest="inline_last_bucket", type=int, choices=(0, 1),
                   default=defaults.inline_last_bucket,
                   help="ddp mode: all-reduce the bucket that ends backward on the compute stream "
                        "(default: only with one rank)")
    p.add_argument("--no-broadcast-buffers", dest="broadcast_buffers", action="store_false",
                   default=defaults.broadcast_buffers)
    p.add_argument("--comm-timeout", dest="comm_timeout_s", type=float,
                   default=defaults.comm_timeout_s)
    p.add_argument("--profile-ranges", action="store_true", default=defaults.profile_ranges)
    p.add_argument("--report-overlap", action="store_true", default=defaults.report_overlap,
                   help="after training: time the step with and without its collectives and "
                        "the collectives alone; print the exposed time and overlap fraction")
    p.add_argument("--report-comm-volume", action="store_true",
                   default=defaults.report_comm_volume,
                   help="after the training iterations: bytes this rank put on / took off its "
                        "links and the collectives it issued (the reference's report samples "
                        "this per node, group25.pdf p4, p6, p8)")
    p.add_argument("--save", dest="checkpoint_out", type=str, default=None)
    p.add_argument("--load", dest="checkpoint_in", type=str, default=None)
    p.add_argument("--threads", type=int, default=defaults.threads)
    return p


def config_from_args(defaults: TrainConfig, argv: Optional[Sequence[str]] = None,
                     distributed_flags: bool = True) -> TrainConfig:
    ns = build_parser(defaults, distributed_flags).parse_args(argv)
    return defaults.replace(**vars(ns))


def _apply_launcher_env(cfg: TrainConfig) -> TrainConfig:
    """When started by ``torchrun`` / ``launch.py`` the rank triple comes from the
    environment instead of the flags."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and cfg.num_nodes == 1 \
            and int(os.environ["WORLD_SIZE"]) > 1:
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = os.environ.get("MASTER_PORT", "8000")
        cfg = cfg.replace(rank=int(os.environ["RANK"]), num_nodes=int(os.environ["WORLD_SIZE"]),
                          master_ip=f"{addr}:{port}")
    return cfg


def fun_main(cfg: TrainConfig):
    """Distributed entry point; ``sync == none`` runs without a process group like the
    reference's part1."""
    from .engine.trainer import Trainer, resolve_device
    from .parallel.comm import init_communicator

    cfg = _apply_launcher_env(cfg)
    torch.set_num_threads(cfg.threads)
    local_rank = int(os.environ.get("LOCAL_RANK", cfg.rank))
    device = resolve_device(cfg, local_rank)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    comm = None
    solo_cpu = False
    if cfg.sync != SYNC_NONE:
        print(f"master_ip = {cfg.master_ip} and nodes = {cfg.num_nodes}")
        # a group of one on the CPU device needs no transport (every strategy is the identity);
        # with more ranks the CPU device has none to offer and init_communicator says so
        solo_cpu = device.type != "cuda" and cfg.num_nodes == 1
        if not solo_cpu:
            try:
                comm = init_communicator(cfg.master_ip, cfg.rank, cfg.num_nodes, device,
                                         cfg.comm_timeout_s)
            except Exception as exc:     # rendezvous failure: clean message, non-zero exit
                print(f"[rank {cfg.rank}] could not join the process group at {cfg.master_ip}: "
                      f"{exc}", file=sys.stderr)
                raise SystemExit(2)
    if comm is not None and device.type == "cuda":
        # collectives are enqueued on the compute stream in the two manual modes (and for the
        # last bucket of the DDP mode): keep RCCL's kernels off the legacy null stream
        torch.cuda.set_stream(torch.cuda.Stream(device=device))
    # identical initial weights on every rank: seed after joining, before the model exists
    torch.manual_seed(cfg.seed)
    if comm is not None:
        print(f"[{os.getpid()}] world_size = {comm.world_size}, "
              + f"rank = {comm.rank}, backend={comm.backend}")
    elif solo_cpu:
        print(f"[{os.getpid()}] world_size = 1, rank = 0, backend=none")
    trainer = Trainer(cfg, device, comm)
    # which engine and which precision actually run (the reference is fp32 on the CPU)
    print(f"Engine: backend={trainer.backend.name}, dtype={trainer.cfg.dtype}, device={device}")
    try:
        result = trainer.run()
    finally:
        trainer.close()
        if comm is not None:
            comm.destroy()
    return result


def main(argv: Optional[Sequence[str]] = None, part: Optional[str] = None):
    defaults = part_defaults(part) if part else TrainConfig()
    cfg = config_from_args(defaults, argv, distributed_flags=(part != "part1"))
    return fun_main(cfg)
