"""Coordinate descent over the per-layer pixel splits of the weight-gradient kernels inside the
full training step (DML_WGRAD_SPLITS), one bench.py run per candidate.

    python tools/tune_wgrad_splits.py [bench args...]
    KNOB=DML_WGRAD_F32_SPLITS START=57,29,15,22,11,11,11 python tools/tune_wgrad_splits.py --dtype fp32
"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ARGS = sys.argv[1:]


def run(splits, waves):
    env = dict(os.environ, DML_WGRAD_WAVES=str(waves))
    env[os.environ.get("KNOB", "DML_WGRAD_SPLITS")] = ",".join(map(str, splits))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup",
                          "30", "--windows", "3"] + ARGS, capture_output=True, text=True, env=env,
                         timeout=200)
    if out.returncode:
        print(out.stderr[-2000:])
        raise SystemExit(1)
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    return rec["windows"]["median_ms_per_step"]


def main():
    waves = int(os.environ.get("WAVES", "8"))
    best = [int(v) for v in os.environ.get("START", "40,20,10,5,3,1,1").split(",")]
    best_t = run(best, waves)
    print(f"start {best} -> {best_t:.4f} ms", flush=True)
    for rnd in range(2):
        for layer in (4, 3, 2, 1, 0, 5, 6):
            cur = best[layer]
            if os.environ.get("FINE"):      # neighbours of a list that is already tuned
                step = max(1, cur // 10)
                cands = sorted({max(1, cur - 2 * step), max(1, cur - step), cur + step,
                                cur + 2 * step} - {cur})
            else:
                cands = sorted({max(1, cur // 2), max(1, cur * 3 // 4), cur + max(1, cur // 4),
                                cur + max(1, cur // 2), cur * 2} - {cur})
            for c in cands:
                trial = list(best)
                trial[layer] = c
                t = run(trial, waves)
                mark = ""
                if t < best_t - float(os.environ.get("THRESH", "0.0015")):
                    best, best_t, mark = trial, t, "  <- better"
                print(f"round {rnd} layer {layer + 1} splits {c:3d}: {t:.4f} ms{mark}", flush=True)
        print(f"after round {rnd}: {best} -> {best_t:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
