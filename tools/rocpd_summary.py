"""Summarise a rocprofv3 `--kernel-trace` result database (rocpd sqlite) as a text table.

    python tools/rocpd_summary.py gpurun_out/prof_bn/bn_results.db --iters 60 > profiles/x.txt

Prints, per kernel: calls, total / mean / min / max duration and share of GPU time, plus
the per-training-step cost when ``--iters`` (warmup + timed steps of the profiled run) is
given.  Template arguments are kept (they identify the tile plan); return types and
argument lists are dropped.
"""
from __future__ import annotations

import argparse
import re
import sqlite3
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):       # cut the argument list: first '(' outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            name = name[:i]
            break
    name = name.replace("dml::", "")
    return name if len(name) <= 88 else name[:85] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--top", type=int, default=60)
    ns = ap.parse_args()
    con = sqlite3.connect(ns.db)
    rows = con.execute("select name, start, end from kernels").fetchall()
    agg = defaultdict(list)
    for name, start, end in rows:
        agg[short(name)].append((end - start) / 1e3)
    total = sum(sum(v) for v in agg.values())
    print(f"# {ns.db}: {len(rows)} kernel dispatches, {total / 1e3:.2f} ms GPU time"
          + (f", {total / ns.iters:.1f} us per step over {ns.iters} steps" if ns.iters else ""))
    hdr = f"{'kernel':88s} {'calls':>7s} {'total_us':>10s} {'mean_us':>8s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}"
    if ns.iters:
        hdr += f" {'us/step':>8s}"
    print(hdr)
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:ns.top]:
        line = (f"{name:88s} {len(v):7d} {sum(v):10.1f} {sum(v) / len(v):8.2f} {min(v):8.2f} "
                f"{max(v):8.2f} {100 * sum(v) / total:6.2f}")
        if ns.iters:
            line += f" {sum(v) / ns.iters:8.2f}"
        print(line)


if __name__ == "__main__":
    main()
