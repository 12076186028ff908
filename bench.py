"""Headline benchmark: VGG-11 training throughput on synthetic CIFAR-10-shaped data.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

Flagship configuration = the reference's fastest variant (part3,
/root/reference/part3/main.py): VGG-11 with BatchNorm, SGD lr 0.1 / momentum 0.9 /
weight decay 1e-4, bucketed all-reduce (25 MB buckets, mean) overlapped with backward,
here with 256 images per GPU (weak scaling), bf16 compute with fp32 master weights.
The timed region is the reference's (/root/reference/part3/main.py:50-67): zero_grad,
forward, loss, backward (+ gradient sync), optimizer step; batches are prepared before.
Other reference configurations: ``--mode part1|part2a|part2b|part3`` and
``--no-batch-norm``; ``--backend torch`` times stock PyTorch-ROCm on the same GPU.

Rank 0 prints ONE JSON line; ``value`` is whole-job images/sec, ``vs_baseline`` divides by
the best number the reference publishes (part3 on its 4-node CPU cluster, 305.4 images/s).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from distributed_machine_learning_amd.config import SEED, TrainConfig  # noqa: E402

REFERENCE_IMAGES_PER_SEC = {"part1": 106.8, "part2a": 211.4, "part2b": 273.9, "part3": 305.4}
MODE_SYNC = {"part1": "none", "part2a": "gather_scatter", "part2b": "all_reduce", "part3": "ddp"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=tuple(MODE_SYNC), default="part3")
    ap.add_argument("--batch-size", type=int, default=256, help="images per GPU")
    bn = ap.add_mutually_exclusive_group()
    bn.add_argument("--batch-norm", dest="batch_norm", action="store_true", default=None)
    bn.add_argument("--no-batch-norm", dest="batch_norm", action="store_false")
    ap.add_argument("--backend", choices=("native", "torch"), default="native")
    ap.add_argument("--dtype", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (1 GPU)")
    ap.add_argument("--bucket-cap-mb", type=float, default=10.0,
                    help="gradient bucket size.  10 MiB (first bucket included) cuts VGG-11's "
                         "37 MB of gradients into {fc1, conv8, conv7} - on the wire a quarter of "
                         "the way into backward -, {conv6, conv5} and {conv4 .. conv1}, which "
                         "--tail-bucket-cap-mb splits again.  Buckets leave from the weight-"
                         "gradient stream, so an extra bucket costs the compute stream nothing "
                         "(measured at one rank: 2 .. 24 MiB caps are within 2 %% of each other); "
                         "smaller buckets put the first bytes on the wire earlier")
    ap.add_argument("--first-bucket-cap-mb", type=float, default=None,
                    help="defaults to --bucket-cap-mb")
    ap.add_argument("--tail-bucket-cap-mb", type=float, default=1.0,
                    help="split the last bucket so that at most this much completes with the "
                         "final wgrad ({conv4, conv3} then hide behind conv2/conv1 backward and "
                         "only {conv2, conv1}, 0.3 MB, is exposed)")
    ap.add_argument("--serialize-streams", action="store_true")
    ap.add_argument("--compute-stream", choices=("auto", "default", "new"), default="auto",
                    help="run the step on torch's default (null) stream or on a created one; auto: "
                         "a created stream whenever collectives are issued (RCCL kernels then never "
                         "sit on the legacy null stream), the default stream otherwise - no "
                         "difference in step time was measured")
    ap.add_argument("--wgrad-overlap", type=int, default=1, choices=(0, 1),
                    help="weight-gradient kernels on a second HIP stream")
    ap.add_argument("--wgrad-group", type=int, default=0, help="0 = auto")
    ap.add_argument("--fused-optimizer", type=int, default=0, choices=(0, 1),
                    help="SGD slices inside backward, behind each group's / bucket's gradients")
    ap.add_argument("--distinct-batches", type=int, default=4)
    ap.add_argument("--report-overlap", action="store_true",
                    help="with communication on: also time the step without collectives and the "
                         "collectives alone, and report the exposed communication time and the "
                         "overlap fraction (extra untimed steps after the measured region)")
    ap.add_argument("--force-comm", action="store_true",
                    help="build the RCCL communicator and sync strategy even with one rank")
    return ap.parse_args()


def main():
    ns = parse()
    # stdout carries exactly one line, the JSON record; everything else a library prints there
    # (RCCL's version banner at its first collective) is sent to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != ns.gpus:
        if world == 1 and ns.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run "
                             "(one process per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {ns.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if ns.compute_stream == "new" or (ns.compute_stream == "auto" and (world > 1 or ns.force_comm)):
        torch.cuda.set_stream(torch.cuda.Stream(device=device))

    from distributed_machine_learning_amd.data import synthetic_cifar
    from distributed_machine_learning_amd.data.transforms import BatchTransform
    from distributed_machine_learning_amd.engine.trainer import make_backend
    from distributed_machine_learning_amd.parallel.comm import RcclCommunicator

    batch_norm = ns.batch_norm if ns.batch_norm is not None else (ns.mode == "part3")
    use_comm = world > 1 or ns.force_comm
    sync = MODE_SYNC[ns.mode] if use_comm else "none"
    master = f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}:{os.environ.get('MASTER_PORT', '29500')}"
    comm = RcclCommunicator(master, rank, world, local_rank, timeout_s=120) if use_comm else None

    cfg = TrainConfig(master_ip=master, rank=rank, num_nodes=world, sync=sync,
                      batch_size=ns.batch_size, batch_norm=batch_norm, dtype=ns.dtype,
                      backend=ns.backend, device="cuda", use_graph=ns.graph,
                      bucket_cap_mb=ns.bucket_cap_mb,
                      first_bucket_cap_mb=(ns.first_bucket_cap_mb if ns.first_bucket_cap_mb
                                           is not None else ns.bucket_cap_mb),
                      tail_bucket_cap_mb=ns.tail_bucket_cap_mb,
                      serialize_streams=ns.serialize_streams,
                      wgrad_overlap=bool(ns.wgrad_overlap), wgrad_group=ns.wgrad_group,
                      fused_optimizer=bool(ns.fused_optimizer),
                      summary=False, eval=False)
    torch.manual_seed(SEED)
    backend = make_backend(cfg, device, comm)
    backend.train_mode(True)

    # synthetic CIFAR-shaped batches, rank-strided like the reference's sampler, prepared
    # (augmented, normalised, laid out) before the clock starts
    n_img = ns.batch_size * world * ns.distinct_batches
    ds = synthetic_cifar(n_img, SEED).to(device)
    tf = BatchTransform(True, backend.loader_layout, SEED)
    batches = []
    for i in range(ns.distinct_batches):
        idx = torch.arange(ns.batch_size, device=device) * world + rank + i * ns.batch_size * world
        imgs, labels = ds.take(idx)
        batches.append((tf(imgs, idx).contiguous(), labels.contiguous()))

    def barrier():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    # RCCL builds its channels on first use of each collective: do that outside any timed or
    # counted step (the gradient buffer is still all zeros here)
    if comm is not None and hasattr(backend, "comm_only_step"):
        for _ in range(2):
            backend.comm_only_step()
        barrier()

    loss = None
    for i in range(ns.warmup):
        loss = backend.train_step(*batches[i % len(batches)])
    barrier()
    t0 = time.perf_counter()
    for i in range(ns.steps):
        loss = backend.train_step(*batches[i % len(batches)])
    barrier()
    elapsed = time.perf_counter() - t0

    # outside the timed region: host cost of enqueueing a step into an empty queue
    probe = 4
    t1 = time.perf_counter()
    for i in range(probe):
        backend.train_step(*batches[i % len(batches)])
    host_elapsed = time.perf_counter() - t1
    barrier()

    overlap = None
    if ns.report_overlap and comm is not None:
        from distributed_machine_learning_amd.utils.overlap import measure_comm_split
        overlap = measure_comm_split(backend, comm, batches, n=max(10, min(ns.steps, 50)))

    if comm is not None:       # MAX over ranks
        times = torch.zeros(world, device=device)
        times[rank] = elapsed
        comm.all_reduce(times, "sum")
        torch.cuda.synchronize()
        elapsed = float(times.max())
    final_loss = float(loss.item()) if loss is not None else float("nan")

    if rank == 0:
        ms = elapsed / ns.steps * 1e3
        value = ns.batch_size * world * ns.steps / elapsed
        par = "single" if world == 1 else f"dp{world}"
        record = json.dumps({
            "metric": "images/sec, VGG-11 CIFAR-10-shaped training, batch 256 per GPU",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": ns.steps,
            "warmup": ns.warmup, "ms_per_step": round(ms, 4), "s_per_iter": round(ms / 1e3, 7),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / REFERENCE_IMAGES_PER_SEC["part3"], 2),
            "dtype": ns.dtype, "data": "synthetic",
            "config": {"model": "VGG-11" + (" + BatchNorm" if batch_norm else ""),
                       "mode": ns.mode, "sync": sync, "global_batch": ns.batch_size * world,
                       "batch_per_gpu": ns.batch_size, "image": "3x32x32", "parallelism": par,
                       "optimizer": "SGD lr0.1 m0.9 wd1e-4 (fused)", "backend": backend.name,
                       "graph": bool(ns.graph), "bucket_cap_mb": ns.bucket_cap_mb, "tail_bucket_cap_mb": ns.tail_bucket_cap_mb,
                       "host_enqueue_ms_per_step": round(host_elapsed / probe * 1e3, 4),
                       "final_loss": round(final_loss, 4), "overlap": overlap,
                       "baseline": "reference part3: 305.4 images/s on 4 CPU nodes "
                                   "(group25.pdf p7); part1: 106.8 images/s"}})
        print(record, file=json_out, flush=True)
    backend.close()
    if comm is not None:
        comm.destroy()


if __name__ == "__main__":
    main()
