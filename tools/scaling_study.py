"""Scalability study: the reference report's last experiment (group25.pdf p10 - time per
iteration against the number of nodes, speed-up and the share of it lost to communication),
run on the GPUs of one node.

    python tools/scaling_study.py                          # part3 at 1, 2, 4, 8 GPUs
    python tools/scaling_study.py --modes part2a part2b part3 --gpus 1 2 4 8 --dtype fp32
    python tools/scaling_study.py --gpus 1 2 --transport loopback --batch-size 32   # one GPU, test transport

One `bench.py` run per (mode, GPU count): bench.py starts its own ranks, so this script never
touches a GPU itself.  Per run the table shows the step time (maximum over ranks), whole-job
images/sec, the weak-scaling efficiency against the same mode's first GPU count, what
communication adds to a step and how much of it backward hides (bench.py's `overlap` block), the
bytes rank 0 moved per step, and the reference's published s/iter at that node count where the
report has one (1-4 CPU nodes; it scales the global batch strongly, 256 images in total, where
this study keeps 256 images per GPU).  `--json FILE` also writes the raw records.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

# s/iter the report publishes: parts at 4 nodes (p3, p5, p7), part1 on one node (p2), and the
# DDP curve read off the chart on p10
REFERENCE_S_PER_ITER = {
    "part1": {1: 2.39596},
    "part2a": {4: 1.21111},
    "part2b": {4: 0.93453},
    "part3": {1: 2.35, 2: 1.20, 3: 1.00, 4: 0.83813},
}


def run_bench(mode, gpus, ns):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--mode", mode,
           "--steps", str(ns.steps), "--warmup", str(ns.warmup), "--windows", str(ns.windows),
           "--dtype", ns.dtype, "--batch-size", str(ns.batch_size), "--transport", ns.transport]
    if gpus == 1 and mode != "part1" and ns.comm_at_one:
        cmd += ["--force-comm", "--report-overlap"]
    cmd += ns.bench_args
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=ns.timeout)
    if out.returncode != 0:
        sys.stderr.write(out.stderr[-2000:])
        raise SystemExit(f"bench.py failed for mode {mode} at {gpus} GPUs (exit {out.returncode})")
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


def table(records):
    head = (f"{'mode':7s} {'GPUs':>4s} {'ms/step':>9s} {'images/s':>11s} {'efficiency':>10s} "
            f"{'exposed comm':>13s} {'overlap':>8s} {'MB/step tx':>11s} {'in sync':>8s} "
            f"{'reference s/iter':>17s} {'speed-up':>9s}")
    rows = [head, "-" * len(head)]
    first = {}
    for rec in records:
        mode, n = rec["config"]["mode"], rec["n_gpus"]
        per_gpu = rec["value"] / n
        first.setdefault(mode, per_gpu)
        eff = per_gpu / first[mode]
        ov = rec["config"].get("overlap") or {}
        exposed = f"{ov['exposed_comm_ms']:.3f} ms" if ov.get("exposed_comm_ms") is not None else "-"
        frac = f"{100 * ov['overlap_fraction']:.0f} %" if ov.get("overlap_fraction") is not None else "-"
        vol = rec.get("comm_bytes_per_step")
        tx = f"{vol['tx'] / 1e6:.1f}" if vol else "-"
        sync = {True: "yes", False: "NO", None: "-"}[rec.get("params_in_sync")]
        ref = REFERENCE_S_PER_ITER.get(mode, {}).get(n)
        ref_s = f"{ref:.3f}" if ref else "-"
        speed = f"{ref / rec['s_per_iter']:.0f}x" if ref else "-"
        rows.append(f"{mode:7s} {n:4d} {rec['ms_per_step']:9.4f} {rec['value']:11.1f} "
                    f"{100 * eff:9.1f}% {exposed:>13s} {frac:>8s} {tx:>11s} {sync:>8s} "
                    f"{ref_s:>17s} {speed:>9s}")
    return "\n".join(rows)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    ap.add_argument("--gpus", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--modes", nargs="+", default=["part3"],
                    choices=("part1", "part2a", "part2b", "part3"))
    ap.add_argument("--dtype", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--batch-size", type=int, default=256, help="images per GPU (weak scaling)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=3)
    ap.add_argument("--transport", choices=("rccl", "loopback"), default="rccl")
    ap.add_argument("--comm-at-one", action="store_true",
                    help="at one GPU build the communicator and the sync strategy anyway, so that "
                         "the first row carries the strategy's fixed cost")
    ap.add_argument("--timeout", type=float, default=1500.0)
    ap.add_argument("--json", type=str, default=None)
    ap.add_argument("bench_args", nargs=argparse.REMAINDER,
                    help="after `--`: passed to every bench.py run")
    ns = ap.parse_args(argv)
    if ns.bench_args and ns.bench_args[0] == "--":
        ns.bench_args = ns.bench_args[1:]
    records = []
    for mode in ns.modes:
        for n in ns.gpus:
            if mode == "part1" and n > 1:
                continue                        # the single-process configuration
            rec = run_bench(mode, n, ns)
            records.append(rec)
            print(f"[scaling] {mode} at {n} GPU(s): {rec['ms_per_step']:.4f} ms/step, "
                  f"{rec['value']:.0f} images/s", file=sys.stderr, flush=True)
    print(table(records))
    if ns.json:
        with open(ns.json, "w") as f:
            json.dump(records, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
