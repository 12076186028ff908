"""One summary line of a bench.py JSON record read from stdin (used by tools/baseline_table.sh)."""
import json
import sys

d = json.loads(sys.stdin.readlines()[-1])
c = d["config"]
print(f"{d['dtype']:5s} {c['mode']:7s} bn={int('BatchNorm' in c['model'])} sync={c['sync']:15s} "
      f"{d['ms_per_step']:8.4f} ms/step {d['value']:10.1f} img/s  median "
      f"{d['windows']['median_ms_per_step']:.4f}  host_loop {c['host_loop_ms_per_step']:.3f}")
