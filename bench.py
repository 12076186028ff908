"""Headline benchmark: VGG-11 training throughput on synthetic CIFAR-10-shaped data.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus 8 --steps 200 --warmup 20          # starts its own 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

Flagship configuration = the reference's fastest variant (part3,
/root/reference/part3/main.py): VGG-11 with BatchNorm, SGD lr 0.1 / momentum 0.9 /
weight decay 1e-4, bucketed all-reduce (mean) overlapped with backward, here with 256
images per GPU (weak scaling), bf16 compute with fp32 master weights.
The timed region is the reference's (/root/reference/part3/main.py:50-67): zero_grad,
forward, loss, backward (+ gradient sync), optimizer step; batches are prepared before.
Other reference configurations: ``--mode part1|part2a|part2b|part3`` and
``--no-batch-norm``; ``--backend torch`` times stock PyTorch-ROCm on the same GPU.

Launching.  One process per GPU, like the reference's one process per node
(/root/reference/part2/2a/main.py:194-218).  With ``--gpus N > 1`` and no launcher
environment (RANK / WORLD_SIZE) this script starts the N ranks itself: the parent never
touches the GPU, never replaces itself with another program, only spawns children, relays
rank 0's record and enforces a deadline.  Under torchrun it is simply one of the ranks.

Rank 0 prints ONE JSON line; ``value`` is whole-job images/sec, ``vs_baseline`` divides by
the number the reference publishes for the same mode (part3: 305.4 images/s on its 4-node
CPU cluster).  Beyond the harness fields the record proves what ran: the rank count the
transport itself reports (``rccl_nranks``), every rank's own step time, whether the
parameters of all ranks are bit-identical after the timed loop (``params_in_sync``), the
gradient bucket plan, the communication / compute split, and the min / median over further
timing windows.
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REFERENCE_IMAGES_PER_SEC = {"part1": 106.8, "part2a": 211.4, "part2b": 273.9, "part3": 305.4}
MODE_SYNC = {"part1": "none", "part2a": "gather_scatter", "part2b": "all_reduce", "part3": "ddp"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prime-steps", type=int, default=100,
                    help="set-up, before the warm-up steps: this many steps on a snapshot of the "
                         "training state (restored afterwards), so that code objects are loaded "
                         "and the GPU is at its steady clocks whatever --warmup is")
    ap.add_argument("--mode", choices=tuple(MODE_SYNC), default="part3")
    ap.add_argument("--batch-size", type=int, default=256, help="images per GPU")
    bn = ap.add_mutually_exclusive_group()
    bn.add_argument("--batch-norm", dest="batch_norm", action="store_true", default=None)
    bn.add_argument("--no-batch-norm", dest="batch_norm", action="store_false")
    ap.add_argument("--backend", choices=("native", "torch"), default="native")
    ap.add_argument("--device", choices=("cuda", "cpu"), default="cuda",
                    help="cpu: the reference's own part1 set-up (one process, torch on the host, "
                         "/root/reference/part1/main.py:15-16) - stock PyTorch only, fp32, for the "
                         "baseline table's CPU row")
    ap.add_argument("--dtype", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (1 GPU)")
    ap.add_argument("--bucket-cap-mb", type=float, default=4.5,
                    help="gradient bucket size.  A bucket closes once it has reached the cap; "
                         "4.5 MiB gives every large layer of VGG-11 a bucket of its own - {fc1, "
                         "conv8}, {conv7}, {conv6} (9 MB each), {conv5} (4.5 MB) - and leaves "
                         "{conv4 .. conv1}, which --tail-bucket-cap-mb splits again, so the first "
                         "bytes are on the wire as soon as the first weight gradient exists and "
                         "nothing waits for a neighbour layer.  Buckets depart from the weight-"
                         "gradient stream through flag words and cost the compute side about a "
                         "microsecond each (one rank, multi-rank schedule: 3 buckets 0.6266, 4 "
                         "buckets 0.6265, 6 buckets 0.627, 7 buckets 0.629 ms per step)")
    ap.add_argument("--first-bucket-cap-mb", type=float, default=None,
                    help="defaults to --bucket-cap-mb")
    ap.add_argument("--tail-bucket-cap-mb", type=float, default=1.0,
                    help="split the last bucket so that at most this much completes with the "
                         "final wgrad ({conv4, conv3} then hide behind conv2/conv1 backward and "
                         "only {conv2, conv1}, 0.3 MB, is exposed)")
    ap.add_argument("--grad-comm-dtype", choices=("fp32", "bf16"), default="fp32",
                    help="part3: element type of the gradient buckets on the wire")
    ap.add_argument("--inline-last-bucket", type=int, choices=(0, 1), default=None,
                    help="part3: the bucket that ends backward on the compute stream (default: "
                         "only with one rank)")
    ap.add_argument("--serialize-streams", action="store_true")
    ap.add_argument("--compute-stream", choices=("auto", "default", "new"), default="auto",
                    help="run the step on torch's default (null) stream or on a created one; auto: "
                         "a created stream whenever collectives are issued (RCCL kernels then never "
                         "sit on the legacy null stream), the default stream otherwise - no "
                         "difference in step time was measured")
    ap.add_argument("--wgrad-overlap", type=int, default=1, choices=(0, 1),
                    help="weight-gradient kernels on a second HIP stream")
    ap.add_argument("--wgrad-group", type=int, default=0, help="0 = auto")
    ap.add_argument("--update-overlap", type=int, default=None,
                    help="first conv layer updated on the side stream next to the next forward "
                         "(0 off, -1 auto)")
    ap.add_argument("--fused-optimizer", type=int, default=0, choices=(0, 1),
                    help="SGD slices inside backward, behind each group's / bucket's gradients")
    ap.add_argument("--distinct-batches", type=int, default=4)
    ap.add_argument("--windows", type=int, default=5,
                    help="further timing windows of --steps steps after the measured one; their "
                         "min / median go into the record's `windows` key")
    ap.add_argument("--report-overlap", action="store_true",
                    help="also with one rank (needs --force-comm): time the step without "
                         "collectives and the collectives alone, and report the exposed "
                         "communication time and the overlap fraction (extra untimed steps after "
                         "the measured region).  Always on with more than one rank")
    ap.add_argument("--force-comm", action="store_true",
                    help="build the communicator and sync strategy even with one rank")
    ap.add_argument("--transport", choices=("rccl", "loopback"), default="rccl",
                    help="loopback: the TEST transport that lets several ranks share one GPU "
                         "(RCCL refuses that); its numbers say nothing about xGMI")
    ap.add_argument("--comm-timeout", type=float, default=120.0,
                    help="deadline of every host-side wait on a collective")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="self-launch: kill all ranks and exit non-zero after this many seconds")
    ap.add_argument("--dry-launch", action="store_true",
                    help="self-launch check: every rank prints its launcher environment and "
                         "exits without touching a GPU")
    ap.add_argument("--self-checks", choices=("auto", "on", "off"), default="auto",
                    help="numerical self-checks on the live communicator, outside the timed region "
                         "(auto: with more than one rank): flag-word against event hand-offs bit for "
                         "bit, data parallelism against rank 0 alone on the global batch in all "
                         "three sync modes, an fp32 window and a bf16-wire window")
    ap.add_argument("--bucket-departure", choices=("auto", "event", "value"), default="auto",
                    help="part3: how a gradient bucket is handed to the communication stream.  auto: "
                         "the reducer's own default, upgraded to flag words when the self-check has "
                         "just shown them bit-identical to event pairs on this communicator")
    ap.add_argument("--rccl-max-channels", type=int, default=0,
                    help="cap on RCCL's channels (NCCL_MAX_NCHANNELS, set before the communicator "
                         "exists): fewer channels leave more compute units to backward while the "
                         "buckets are in flight; 0 keeps the library's choice")
    ap.add_argument("--plan-policy", choices=("auto", "tall", "tall128", "streaming"), default="auto",
                    help="convolution tile plans of the native bf16 engine: the tall patch-resident "
                         "tiles (one block per CU, nearly all of its LDS; tall128: no tile taller than 128 "
                         "pixels, twice the blocks) or the streaming tiles only. "
                         "auto: tall on one GPU; with more than one rank both are timed on the live job "
                         "before the measured region (a communication library's resident blocks can "
                         "push a one-block-per-CU grid into a second round) and the faster one runs")
    ap.add_argument("--extras-deadline", type=float, default=300.0,
                    help="seconds the checks and extra windows AFTER the measured region may take in "
                         "total; past it rank 0 prints the record it has (extras_error says which "
                         "phase was running) and every rank exits non-zero instead of hanging")
    ap.add_argument("--profile-ranges", action="store_true",
                    help="emit roctx ranges for the phases of the step (rocprofv3 --marker-trace)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------- self launch
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(ns, argv) -> int:
    """Parent of a self-launched run.  Makes no GPU call and no exec of its own."""
    port = _free_port()
    procs = []
    # (the loopback transport names its shared-memory segment after port and launch)
    launch_id = f"{os.getpid():x}{int(time.time() * 1e3) & 0xffffffff:x}"
    for rank in range(ns.gpus):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ns.gpus),
                   LOCAL_WORLD_SIZE=str(ns.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DML_BENCH_SELF_LAUNCHED="1",
                   DML_LAUNCH_ID=launch_id)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), *argv],
                                      env=env))
    deadline = time.time() + ns.launch_timeout
    code = 0
    try:
        alive = set(range(ns.gpus))
        while alive:
            for r in list(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0 and code == 0:
                    code = rc
                    print(f"[bench] rank {r} exited with code {rc}; stopping the others",
                          file=sys.stderr)
            if code != 0:
                break
            if time.time() > deadline:
                print(f"[bench] ranks still running after {ns.launch_timeout:.0f} s; giving up",
                      file=sys.stderr)
                code = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                p.kill()
    return code


# ------------------------------------------------------------------------------ a rank
def run_rank(ns) -> None:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    master = f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}:{os.environ.get('MASTER_PORT', '29500')}"
    if world != ns.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {ns.gpus}")
    if ns.dry_launch:
        # the bootstrap exchange of the real run (rank 0 publishes the communicator's id, the
        # others fetch it) with a stand-in id, through the store the real run would use
        boot = None
        if world > 1:
            import hashlib
            from distributed_machine_learning_amd.parallel.comm import exchange_bootstrap_bytes
            blob, _store = exchange_bootstrap_bytes(master, rank, world, 60.0,
                                                    lambda: os.urandom(128), key="dml_dry_launch_id")
            boot = hashlib.sha256(blob).hexdigest()[:16]
        # one write per line: the ranks share the parent's stdout
        os.write(1, (json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank,
                                 "world_size": world, "master": master, "bootstrap_id": boot,
                                 "agent_store": os.environ.get("TORCHELASTIC_USE_AGENT_STORE",
                                                               "").lower() == "true",
                                 "self_launched": os.environ.get("DML_BENCH_SELF_LAUNCHED") == "1"})
                     + "\n").encode())
        return

    import torch
    from distributed_machine_learning_amd.config import SEED, TrainConfig

    # stdout carries exactly one line, the JSON record; everything else a library prints there
    # (RCCL's version banner at its first collective) is sent to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    on_cpu = ns.device == "cpu"
    if on_cpu:
        if ns.backend != "torch" or world != 1 or ns.force_comm:
            raise SystemExit("--device cpu runs one process of the stock PyTorch backend "
                             "(--backend torch, --gpus 1, no --force-comm)")
        ns.dtype = "fp32"
        ns.prime_steps = 0
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU")
    n_dev = 1 if on_cpu else torch.cuda.device_count()
    if on_cpu:
        dev_index = 0
    elif ns.transport == "loopback":
        dev_index = local_rank % n_dev
    elif local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPUs are visible "
                         "(RCCL needs one GPU per rank)")
    else:
        dev_index = local_rank
    device = torch.device("cpu") if on_cpu else torch.device("cuda", dev_index)
    if not on_cpu:
        torch.cuda.set_device(device)
    use_comm = world > 1 or ns.force_comm
    if not on_cpu and (ns.compute_stream == "new" or (ns.compute_stream == "auto" and use_comm)):
        torch.cuda.set_stream(torch.cuda.Stream(device=device))

    from distributed_machine_learning_amd.data import synthetic_cifar
    from distributed_machine_learning_amd.data.transforms import BatchTransform
    from distributed_machine_learning_amd.engine.trainer import make_backend
    from distributed_machine_learning_amd.parallel.comm import RcclCommunicator

    batch_norm = ns.batch_norm if ns.batch_norm is not None else (ns.mode == "part3")
    sync = MODE_SYNC[ns.mode] if use_comm else "none"
    comm = None
    if ns.rccl_max_channels > 0:
        os.environ["NCCL_MAX_NCHANNELS"] = str(ns.rccl_max_channels)
    if use_comm:
        comm = RcclCommunicator(master, rank, world, dev_index, timeout_s=ns.comm_timeout,
                                transport=ns.transport)

    cfg = TrainConfig(master_ip=master, rank=rank, num_nodes=world, sync=sync,
                      batch_size=ns.batch_size, batch_norm=batch_norm, dtype=ns.dtype,
                      backend=ns.backend, device=ns.device, use_graph=ns.graph,
                      bucket_cap_mb=ns.bucket_cap_mb,
                      first_bucket_cap_mb=(ns.first_bucket_cap_mb if ns.first_bucket_cap_mb
                                           is not None else ns.bucket_cap_mb),
                      tail_bucket_cap_mb=ns.tail_bucket_cap_mb,
                      grad_comm_dtype=ns.grad_comm_dtype,
                      inline_last_bucket=(None if ns.inline_last_bucket is None
                                          else bool(ns.inline_last_bucket)),
                      serialize_streams=ns.serialize_streams, comm_timeout_s=ns.comm_timeout,
                      wgrad_overlap=bool(ns.wgrad_overlap), wgrad_group=ns.wgrad_group,
                      fused_optimizer=bool(ns.fused_optimizer),
                      update_overlap=(TrainConfig().update_overlap if ns.update_overlap is None
                                      else int(ns.update_overlap)),
                      summary=False, eval=False)
    torch.manual_seed(SEED)
    backend = make_backend(cfg, device, comm)
    backend.train_mode(True)
    if ns.profile_ranges and backend.name == "native":
        from distributed_machine_learning_amd.ops import native
        native.lib().set_trace_ranges(True)

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
        # bounded: a lost rank raises here instead of hanging the survivors
        if comm is not None:
            comm.barrier(ns.comm_timeout)
        if not on_cpu:
            torch.cuda.synchronize()

    host_loop = []      # per timed window: seconds until the LAST train_step call returned

    def timed(n, first=0):
        barrier()
        t = time.perf_counter()
        last = None
        for i in range(n):
            last = backend.train_step(*batches[(first + i) % len(batches)])
        host_loop.append(time.perf_counter() - t)
        barrier()
        return time.perf_counter() - t, last

    # RCCL builds its channels on first use of each collective: do that outside any timed or
    # counted step (the gradient buffer is still all zeros here)
    if comm is not None and hasattr(backend, "comm_only_step"):
        for _ in range(2):
            backend.comm_only_step()
        barrier()

    # ---- which form of hand-off the timed run uses (outside the timed region).  With RCCL and
    # more than one rank the reducer starts on event pairs; when flag words reproduce their result
    # bit for bit on THIS communicator the run switches to them.
    checks_on = ns.self_checks == "on" or (ns.self_checks == "auto" and world > 1)
    self_checks = {}
    checks_failed = []
    native_ddp = backend.name == "native" and getattr(backend, "reducer", None) is not None
    if native_ddp and ns.bucket_departure != "auto":
        backend.reducer.set_departure_by_value(ns.bucket_departure == "value")
    if native_ddp and checks_on and backend.reducer.can_depart_by_value:
        try:
            from distributed_machine_learning_amd.utils.selfcheck import handoff_forms_agree
            res = handoff_forms_agree(backend, comm, batches, steps=3)
            self_checks["handoff_forms"] = res
            if res["ok"] is None:
                # inconclusive (see the record): event pairs stay, nothing failed
                if ns.bucket_departure == "auto":
                    backend.reducer.set_departure_by_value(False)
            elif not res["ok"]:
                checks_failed.append("handoff_forms")
                backend.reducer.set_departure_by_value(False)
            elif ns.bucket_departure == "auto" and not os.environ.get("DML_REDUCER_HANDOFF"):
                # proven equal on this very job: the cheaper form carries the timed steps
                # (an explicit DML_REDUCER_HANDOFF stays what the caller asked for)
                backend.reducer.set_departure_by_value(True)
        except Exception as exc:                # noqa: BLE001 - reported in the record
            self_checks["handoff_forms"] = {"ok": False, "error": f"{type(exc).__name__}: {exc}"}
            checks_failed.append("handoff_forms")
        barrier()

    # ---- convolution plan policy (outside the timed region)
    plan_policy = None
    if backend.name == "native" and ns.dtype == "bf16":
        from distributed_machine_learning_amd.ops import native as _native
        _C = _native.lib()
        def set_policy(name):
            _C.set_conv_tall_tiles(name != "streaming")
            _C.set_conv_tall_max_rows(128 if name == "tall128" else 256)

        if ns.plan_policy != "auto":
            set_policy(ns.plan_policy)
            plan_policy = {"chosen": ns.plan_policy, "how": "flag"}
        elif world > 1 and comm is not None:
            tried = {}
            for name in ("tall", "tall128", "streaming"):
                set_policy(name)
                for i in range(10):
                    backend.train_step(*batches[i % len(batches)])
                t_pol, _ = timed(40)
                host_loop.pop()
                times = comm.all_gather_scalars(t_pol, torch.float32)
                tried[name] = round(float(times.max()) / 40 * 1e3, 4)
            chosen = min(tried, key=tried.get)
            set_policy(chosen)
            plan_policy = {"chosen": chosen, "how": "timed on this job", "ms_per_step": tried}
        else:
            now = ("streaming" if not _C.conv_tall_tiles()
                   else "tall128" if _C.conv_tall_max_rows() == 128 else "tall")
            plan_policy = {"chosen": now, "how": "default"}

    if ns.prime_steps > 0 and hasattr(backend, "prime"):
        backend.prime(batches, ns.prime_steps)
        barrier()

    loss = None
    for i in range(ns.warmup):
        loss = backend.train_step(*batches[i % len(batches)])
    if comm is not None:
        comm.reset_volume()                     # (host-side counters of what gets issued)
    elapsed, last = timed(ns.steps)             # THE measured region: exactly --steps steps
    volume = comm.volume() if comm is not None else None
    loss = last if last is not None else loss
    final_loss = float(loss.item()) if loss is not None else float("nan")

    # ---- everything below is outside the measured region.  The extras (further windows, host
    # probe, self checks, communication split) must never cost the run its record: a failure in
    # one of them is reported in the record's `extras_error` instead.
    extras_error = None
    phase = ["further windows"]
    extra_windows = {}
    windows = []
    host_elapsed = float("nan")
    probe = 20                                  # host cost of enqueueing steps into an empty queue
    per_rank_ms = [round(elapsed / ns.steps * 1e3, 4)]
    nranks_seen = comm.peer_count if comm is not None else 1
    in_sync = None
    overlap = None
    def make_record(error):
        ms = elapsed / ns.steps * 1e3
        value = ns.batch_size * world * ns.steps / elapsed
        par = "single" if world == 1 else f"dp{world}"
        win_ms = sorted(w / ns.steps * 1e3 for w in windows)
        record = {
            "metric": "images/sec, VGG-11 CIFAR-10-shaped training, batch 256 per GPU",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": ns.steps,
            "warmup": ns.warmup, "ms_per_step": round(ms, 4), "s_per_iter": round(ms / 1e3, 7),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / REFERENCE_IMAGES_PER_SEC[ns.mode], 2),
            "dtype": ns.dtype, "data": "synthetic",
            "rccl_nranks": nranks_seen, "transport": ns.transport if comm is not None else "none",
            "per_rank_ms_per_step": per_rank_ms, "params_in_sync": in_sync,
            "extras_error": error,
            # numerical self-checks on the live communicator (utils/selfcheck.py): None = not run
            "handoff_bitwise_ok": (self_checks.get("handoff_forms") or {}).get("ok"),
            "dp_vs_single_rel_err": ({m: v["rel_err_vs_single"] for m, v in
                                      self_checks["data_parallel"]["modes"].items()}
                                     if "data_parallel" in self_checks else None),
            "self_checks": self_checks or None,
            "self_checks_failed": checks_failed or None,
            "extras": extra_windows or None,
            # MODELLED bytes rank 0 puts on / takes off its links per step - ring all-reduce,
            # direct gather / scatter - computed from the collectives issued inside the measured
            # region (its two barriers' 4-byte tokens included); not a hardware link counter, and
            # RCCL may run another algorithm
            "comm_bytes_per_step": ({"model": "ring all-reduce, direct gather/scatter",
                                     "tx": volume["tx_bytes"] // ns.steps,
                                     "rx": volume["rx_bytes"] // ns.steps,
                                     "calls": {k: v for k, v in volume["calls"].items() if v}}
                                    if volume is not None else None),
            "windows": ({"n": len(win_ms), "steps_each": ns.steps,
                         "min_ms_per_step": round(win_ms[0], 4),
                         "median_ms_per_step": round(win_ms[len(win_ms) // 2], 4),
                         "max_ms_per_step": round(win_ms[-1], 4)} if win_ms else None),
            "config": {"model": "VGG-11" + (" + BatchNorm" if batch_norm else ""), "device": ns.device,
                       "rccl_max_channels": ns.rccl_max_channels or None,
                       "plan_policy": plan_policy,
                       "mode": ns.mode, "sync": sync, "global_batch": ns.batch_size * world,
                       "batch_per_gpu": ns.batch_size, "image": "3x32x32", "parallelism": par,
                       "optimizer": "SGD lr0.1 m0.9 wd1e-4 (fused)", "backend": backend.name,
                       "graph": bool(ns.graph), "prime_steps": ns.prime_steps,
                       "bucket_cap_mb": ns.bucket_cap_mb,
                       "tail_bucket_cap_mb": ns.tail_bucket_cap_mb,
                       "sync_plan": backend.sync_plan() if hasattr(backend, "sync_plan") else None,
                       "host_enqueue_ms_per_step": round(host_elapsed / probe * 1e3, 4),
                       # the measured window's calls had all returned after this long: equal to
                       # ms_per_step = the host is the bottleneck, well below = the GPU is
                       "host_loop_ms_per_step": round(host_loop[0] / ns.steps * 1e3, 4),
                       "final_loss": round(final_loss, 4), "overlap": overlap,
                       "baseline": f"reference {ns.mode}: {REFERENCE_IMAGES_PER_SEC[ns.mode]} "
                                   "images/s on its CPU cluster (group25.pdf)"}}
        return record

    import threading

    def deadline_passed():
        # a collective that never returns (a peer lost after the measured region) must not cost the
        # run its record: print what there is, say where it stopped, leave without teardown
        msg = (f"the work after the measured region did not finish within {ns.extras_deadline:.0f} s "
               f"(phase: {phase[0]})")
        print(f"[bench] rank {rank}: {msg}", file=sys.stderr, flush=True)
        if rank == 0:
            try:
                print(json.dumps(make_record(msg)), file=json_out, flush=True)
            except Exception as exc:            # noqa: BLE001
                print(f"[bench] no record: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
            finally:
                os._exit(4)
        os._exit(4)

    watchdog = threading.Timer(ns.extras_deadline, deadline_passed)
    watchdog.daemon = True
    watchdog.start()
    try:
        windows = [timed(ns.steps)[0] for _ in range(max(ns.windows, 0))]
        phase[0] = "host enqueue probe"
        t1 = time.perf_counter()
        for i in range(probe):
            backend.train_step(*batches[i % len(batches)])
        host_elapsed = time.perf_counter() - t1
        barrier()

        # ---- cross-rank self checks
        phase[0] = "maximum over ranks and parameter checksum"
        if comm is not None:
            times = comm.all_gather_scalars(elapsed, torch.float32)
            per_rank_ms = [round(float(t) / ns.steps * 1e3, 4) for t in times]
            elapsed = float(times.max())                        # MAX over ranks
            if windows:
                w = torch.zeros(world, len(windows), device=device)
                w[rank] = torch.tensor(windows, device=device)
                comm.all_reduce(w)
                comm.guard_sync(ns.comm_timeout)
                windows = [float(v) for v in w.max(dim=0).values.cpu()]
            # bit pattern checksum of the master weights: equal on every rank, or training diverged
            flat = backend.parameters_flat()
            checksum = int(flat.view(torch.int32).to(torch.int64).sum().item())
            sums = comm.all_gather_scalars(checksum, torch.int64)
            in_sync = bool((sums == sums[0]).all())

        phase[0] = "communication / compute split"
        if comm is not None and (world > 1 or ns.report_overlap):
            from distributed_machine_learning_amd.utils.overlap import measure_comm_split
            overlap = measure_comm_split(backend, comm, batches, n=max(10, min(ns.steps, 50)))

        # ---- correctness against an independent computation (see utils/selfcheck.py)
        phase[0] = "data-parallel steps against the single-process global batch"
        if comm is not None and checks_on and backend.name == "native":
            from distributed_machine_learning_amd.utils import selfcheck
            res = selfcheck.dp_and_manual_modes(cfg.replace(dtype=ns.dtype), comm, device,
                                                backend.loader_layout)
            self_checks["data_parallel"] = res
            if not res["ok"]:
                checks_failed.append("data_parallel")
    except Exception as exc:                    # noqa: BLE001 - reported, then the run fails
        extras_error = f"{type(exc).__name__}: {exc}"
        print(f"[bench] rank {rank}: after the measured region: {extras_error}", file=sys.stderr)
        if comm is not None and len(per_rank_ms) == 1 and world > 1:
            # the MAX over ranks could not be taken: say so instead of passing off one rank's time
            extras_error += " (ms_per_step is rank 0's own time, not the maximum over ranks)"

    # ---- optional extra windows (never fatal): the reference's precision and bf16 on the wire
    if comm is not None and checks_on and backend.name == "native" and extras_error is None:
        from distributed_machine_learning_amd.utils import selfcheck

        def batches_for(layout):
            t2 = BatchTransform(True, layout, SEED)
            out = []
            for i in range(2):
                idx = torch.arange(ns.batch_size, device=device) * world + rank + i * ns.batch_size * world
                imgs, labels = ds.take(idx)
                out.append((t2(imgs, idx).contiguous(), labels.contiguous()))
            return out

        for name, over in (("fp32", {"dtype": "fp32"}),
                           ("bf16_wire", {"dtype": "bf16", "grad_comm_dtype": "bf16"})):
            if name == "bf16_wire" and (sync != "ddp" or ns.dtype != "bf16"):
                continue
            if name == "fp32" and ns.dtype == "fp32":
                continue
            phase[0] = f"extra window {name}"
            try:
                extra_windows[name + "_ms_per_step"] = selfcheck.timed_window(
                    cfg, comm, device, batches_for, steps=20, warmup=5, **over)
            except Exception as exc:            # noqa: BLE001
                extra_windows[name + "_error"] = f"{type(exc).__name__}: {exc}"
        try:
            extra_windows["transport"] = selfcheck.transport_facts()
        except Exception as exc:                # noqa: BLE001
            extra_windows["transport_error"] = f"{type(exc).__name__}: {exc}"


    watchdog.cancel()
    if rank == 0:
        print(json.dumps(make_record(extras_error)), file=json_out, flush=True)
    if extras_error is not None:
        # The measured region itself completed on every rank (its closing barrier returned), so
        # the record stands and says what failed afterwards.  A failed collective may have left
        # the communicator unusable: no orderly teardown.  With more than one rank a run whose
        # cross-rank checks could not be completed is NOT a verified number: non-zero exit.
        sys.stderr.flush()
        os._exit(1 if (in_sync is False or (world > 1 and in_sync is not True)) else 0)
    backend.close()
    if comm is not None:
        comm.destroy()
    if in_sync is False:
        raise SystemExit("parameters differ between ranks after the timed loop")
    if checks_failed:
        raise SystemExit(f"self-checks failed: {', '.join(checks_failed)} (see the record)")


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    ns = parse(argv)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if ns.gpus > 1 and not launched:
        return self_launch(ns, argv)
    run_rank(ns)
    return 0


if __name__ == "__main__":
    sys.exit(main())
