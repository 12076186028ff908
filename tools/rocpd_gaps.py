"""Idle gaps between consecutive kernels in a rocprofv3 kernel trace (rocpd sqlite).

    python tools/rocpd_gaps.py results.db [--skip 0.5] [--top 25]

Reports GPU busy time, the span, and the largest idle gaps with the kernels on either side
- the way to see launch bubbles, cross-stream event hand-offs and exposed collectives.
"""
from __future__ import annotations

import argparse
import sqlite3

from rocpd_summary import short


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip", type=float, default=0.5, help="ignore this leading fraction (warmup)")
    ap.add_argument("--top", type=int, default=25)
    ns = ap.parse_args()
    con = sqlite3.connect(ns.db)
    rows = sorted(con.execute("select start, end, name from kernels").fetchall())
    rows = rows[int(len(rows) * ns.skip):]
    busy = 0
    cur_end = rows[0][0]
    gaps = []
    for i, (s, e, n) in enumerate(rows):
        if s > cur_end:
            gaps.append((s - cur_end, short(rows[i - 1][2]), short(n)))
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    span = cur_end - rows[0][0]
    print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms "
          f"({100 * busy / span:.1f}%), idle {(span - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
    agg = {}
    for g, a, b in gaps:
        k = (a[:50], b[:50])
        t = agg.setdefault(k, [0, 0])
        t[0] += g
        t[1] += 1
    print(f"{'after':50s} {'before':50s} {'count':>6s} {'total_us':>9s} {'mean_us':>8s}")
    for (a, b), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:ns.top]:
        print(f"{a:50s} {b:50s} {c:6d} {t / 1e3:9.1f} {t / c / 1e3:8.2f}")


if __name__ == "__main__":
    main()
