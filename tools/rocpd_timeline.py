"""One training step of a rocprofv3 kernel trace (rocpd sqlite) as a per-stream timeline.

    python tools/rocpd_timeline.py results.db [--anchor conv1_fprop] [--step -3]

A step is delimited by consecutive dispatches of the anchor kernel (the first layer's forward).
Prints every kernel of the chosen step with its stream, start offset, duration and the idle
time on its stream since the previous kernel there, then per-stream busy time and the step's
span - the view that shows which stream is the critical path and where hand-offs stall.
"""
from __future__ import annotations

import argparse
import sqlite3

from rocpd_summary import short


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--anchor", default="conv1_fprop")
    ap.add_argument("--step", type=int, default=-3, help="which step (negative: from the end)")
    ap.add_argument("--summary", action="store_true", help="per-stream totals only")
    ns = ap.parse_args()
    con = sqlite3.connect(ns.db)
    rows = sorted(con.execute("select start, end, name, stream_id from kernels").fetchall())
    marks = [i for i, r in enumerate(rows) if ns.anchor in r[2]]
    if len(marks) < 3:
        raise SystemExit(f"fewer than 3 dispatches of {ns.anchor!r}")
    k = ns.step if ns.step >= 0 else len(marks) - 1 + ns.step
    lo, hi = marks[k], marks[k + 1]
    step = rows[lo:hi]
    t0 = step[0][0]
    streams = sorted({r[3] for r in step}, key=lambda s: min(r[0] for r in step if r[3] == s))
    label = {s: "ABCDEFGH"[i] for i, s in enumerate(streams)}
    last_end = {}
    busy = {s: 0 for s in streams}
    print(f"# step {k} of {len(marks) - 1}: {len(step)} kernels on {len(streams)} streams, "
          f"span {(rows[hi][0] - t0) / 1e3:.1f} us (anchor to anchor)")
    if not ns.summary:
        print(f"{'stream':>6s} {'start_us':>9s} {'dur_us':>8s} {'idle_us':>8s}  kernel")
    for s, e, n, st in step:
        idle = (s - last_end[st]) / 1e3 if st in last_end else 0.0
        last_end[st] = e
        busy[st] += e - s
        if not ns.summary:
            print(f"{label[st]:>6s} {(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {idle:8.2f}  {short(n)[:90]}")
    for st in streams:
        ks = [r for r in step if r[3] == st]
        print(f"# stream {label[st]}: {len(ks)} kernels, busy {busy[st] / 1e3:.1f} us, "
              f"first start {(ks[0][0] - t0) / 1e3:.1f} us, last end {(ks[-1][1] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
