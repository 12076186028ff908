"""Per-kernel hardware-counter table from a rocprofv3 --pmc result database (rocpd sqlite).

    python tools/rocpd_pmc.py out/pmc_results.db [--last 1]

Averages each counter over the dispatches of a kernel (optionally only the last N
dispatches of each distinct (kernel, grid) pair, to drop cold runs).
"""
from __future__ import annotations

import argparse
import sqlite3
from collections import defaultdict

from rocpd_summary import short


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--filter", default="conv")
    ap.add_argument("--derive", action="store_true",
                    help="ratios instead of raw counters (needs GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, "
                         "SQ_WAIT_ANY, SQ_WAVE_CYCLES, SQ_LDS_BANK_CONFLICT, SQ_BUSY_CU_CYCLES)")
    ns = ap.parse_args()
    con = sqlite3.connect(ns.db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    want = [c for c in ("dispatch_id", "kernel_name", "name", "counter_name", "value", "grid_size",
                        "grid_size_x", "workgroup_size", "start", "end") if c in cols]
    rows = con.execute(f"select {', '.join(want)} from counters_collection").fetchall()
    idx = {c: i for i, c in enumerate(want)}
    kcol = idx.get("kernel_name", idx.get("name"))
    ccol = idx["counter_name"]
    gcol = idx.get("grid_size", idx.get("grid_size_x"))
    table = defaultdict(lambda: defaultdict(list))
    seen = set()
    for r in rows:
        key = (short(str(r[kcol])), r[gcol] if gcol is not None else 0)
        table[key][r[ccol]].append(float(r[idx["value"]]))
        # wall time of the dispatch (ns -> us), once per dispatch, where the database has it
        if "start" in idx and "end" in idx and "dispatch_id" in idx:
            d = r[idx["dispatch_id"]]
            if d not in seen and r[idx["start"]] and r[idx["end"]]:
                seen.add(d)
                table[key]["~wall_us"].append((float(r[idx["end"]]) - float(r[idx["start"]])) / 1e3)
    if ns.derive:
        def mean(v, c):
            xs = v.get(c, [])
            xs = xs[len(xs) // 2:] if len(xs) > 1 else xs
            return sum(xs) / len(xs) if xs else float("nan")
        print("# time = GRBM_GUI_ACTIVE / 8 XCDs / 2.4 GHz; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / "
              "(GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
              "LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_BUSY_CU_CYCLES; wall = dispatch end - start")
        print(f"{'kernel':66s} {'grid':>8s} {'time us':>8s} {'wall us':>8s} {'MFMA busy %':>11s} "
              f"{'LDS conflict %':>14s} {'waiting %':>9s}")
        for (name, grid), v in sorted(table.items()):
            if ns.filter and ns.filter not in name:
                continue
            gui = mean(v, "GRBM_GUI_ACTIVE")
            print(f"{name[:66]:66s} {grid:8d} {gui / 8 / 2400:8.1f} {mean(v, '~wall_us'):8.1f} "
                  f"{100 * mean(v, 'SQ_VALU_MFMA_BUSY_CYCLES') / (gui / 8 * 1024):11.1f} "
                  f"{100 * mean(v, 'SQ_LDS_BANK_CONFLICT') / mean(v, 'SQ_BUSY_CU_CYCLES'):14.1f} "
                  f"{100 * mean(v, 'SQ_WAIT_ANY') / mean(v, 'SQ_WAVE_CYCLES'):9.1f}")
        return
    counters = sorted({c for v in table.values() for c in v})
    print(f"{'kernel':60s} {'grid':>9s} " + " ".join(f"{c[-22:]:>22s}" for c in counters))
    for (name, grid), v in sorted(table.items()):
        if ns.filter and ns.filter not in name:
            continue
        vals = []
        for c in counters:
            xs = v.get(c, [])
            xs = xs[len(xs) // 2:] if len(xs) > 1 else xs        # drop the cold half
            vals.append(sum(xs) / len(xs) if xs else float("nan"))
        print(f"{name[:60]:60s} {grid:9d} " + " ".join(f"{x:22.4g}" for x in vals))


if __name__ == "__main__":
    main()
