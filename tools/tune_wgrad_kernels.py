"""Per-layer choice between the per-tap and the tap-resident weight-gradient kernel (and the
latter's co tile, ring depth and pixel splits) inside the full training step, one bench.py run per
candidate (DML_WGRAD_RESIDENT + DML_WGRAD_SPLITS), starting from the shipped per-tap lists.

    python tools/tune_wgrad_kernels.py [bench args...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ARGS = sys.argv[1:]
# per layer: pixel-split candidates for the tap-resident kernel (its tiles are 64 ci wide, all taps)
SPLITS = [[4, 8, 16, 32], [2, 4, 8, 16], [1, 2, 4, 8], [1, 2, 4], [1, 2, 4], [1, 2], [1, 2]]


def run(resident, splits):
    env = dict(os.environ, DML_WGRAD_RESIDENT=",".join(resident),
               DML_WGRAD_SPLITS=",".join(map(str, splits)))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup",
                          "30", "--windows", "3"] + ARGS, capture_output=True, text=True, env=env,
                         timeout=200)
    if out.returncode:
        return None
    return json.loads(out.stdout.strip().splitlines()[-1])["windows"]["median_ms_per_step"]


def main():
    resident = ["0"] * 7
    splits = [int(v) for v in os.environ.get("START", "40,10,10,2,3,1,1").split(",")]
    best_t = run(resident, splits)
    print(f"start (per-tap, {splits}) -> {best_t:.4f} ms", flush=True)
    for layer in (0, 6, 5, 4, 3, 2, 1):
        for cfg in ("64:2", "64:3", "128:3"):
            for sp in SPLITS[layer]:
                r, s = list(resident), list(splits)
                r[layer], s[layer] = cfg, sp
                t = run(r, s)
                if t is None:
                    print(f"layer {layer + 1} resident {cfg} splits {sp}: refused", flush=True)
                    continue
                mark = ""
                if t < best_t - 0.0015:
                    resident, splits, best_t, mark = r, s, t, "  <- better"
                print(f"layer {layer + 1} resident {cfg} splits {sp}: {t:.4f} ms{mark}", flush=True)
        print(f"after layer {layer + 1}: resident {resident} splits {splits} -> {best_t:.4f} ms",
              flush=True)


if __name__ == "__main__":
    main()
