"""Per-range summary of a rocprofv3 marker trace (rocpd sqlite): count, mean / total time.

    python tools/rocpd_ranges.py results.db

The engine's ranges (csrc/engine/trace.cpp) bracket HOST-side phases of a step - forward,
backward, backward/wgrad-side, backward/buckets, sync/*, optimizer - so their durations are the
host time spent enqueueing each phase, the split behind bench.py's host_loop_ms_per_step.
"""
from __future__ import annotations

import json
import sqlite3
import sys
from collections import defaultdict


def main():
    con = sqlite3.connect(sys.argv[1])
    views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
    src = next((v for v in ("regions", "markers", "rocpd_region") if v in views), None)
    if src is None:
        raise SystemExit(f"no marker table in {sys.argv[1]}: {views}")
    cols = [r[1] for r in con.execute(f"pragma table_info({src})")]
    name = "name" if "name" in cols else next(c for c in cols if "name" in c)
    extra = ", extdata" if "extdata" in cols else ""
    rows = con.execute(f"select {name}, start, end{extra} from {src}").fetchall()
    agg = defaultdict(list)
    for row in rows:
        n, s, e = row[0], row[1], row[2]
        if extra and row[3]:        # roctx ranges carry their text as {"message": "..."}
            try:
                n = json.loads(row[3]).get("message", n)
            except ValueError:
                pass
        if e is not None and s is not None and e >= s:
            agg[str(n)].append((e - s) / 1e3)
    print(f"# {sys.argv[1]}: {len(rows)} ranges from `{src}`")
    print(f"{'range':40s} {'count':>7s} {'mean_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s}")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)[:len(v) - len(v) // 20] if len(v) > 20 else v      # drop the cold 5 %
        print(f"{n[:40]:40s} {len(v):7d} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} "
              f"{sum(v) / 1e3:10.2f}")


if __name__ == "__main__":
    main()
