"""Per-layer choice of the backward-data convolution plan inside the full training step
(DML_DGRAD_PLANS), one bench.py run per candidate; a candidate the launcher refuses is skipped.

    python tools/tune_dgrad_plans.py [bench args...]
    KNOB=DML_FWD_PLANS python tools/tune_dgrad_plans.py          # the forward convolutions instead
"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ARGS = sys.argv[1:]
CANDS = ["128:128:1:0:8", "128:128:1:0:9", "128:128:1:0:10", "128:128:3:-1:10", "128:128:1:0:4", "128:64:1:0:4", "64:64:1:0:4",
         "64:64:1:0:16", "128:64:1:0:16", "128:128:1:0:16", "128:128:3:-1:9", "128:128:3:-1:8",
         "64:64:3:-1:4", "64:64:9:-1:4", "128:64:3:-1:4",
         # tall patch-resident tiles (waves = 20 + variant, see conv_patch.hip)
         "256:64:1:0:20", "256:64:1:0:21", "128:64:1:0:20", "128:64:1:0:21", "128:64:1:0:25",
         "128:64:2:0:20"]


def run(plans):
    env = dict(os.environ)
    env[os.environ.get("KNOB", "DML_DGRAD_PLANS")] = ",".join(plans)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup",
                          "30", "--windows", "3"] + ARGS, capture_output=True, text=True, env=env,
                         timeout=200)
    if out.returncode:
        return None
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    return rec["windows"]["median_ms_per_step"]


def main():
    best = ["0"] * 7
    best_t = run(best)
    print(f"start (default plans) -> {best_t:.4f} ms", flush=True)
    for layer in (6, 5, 4, 3, 2, 1, 0):
        for c in CANDS:
            trial = list(best)
            trial[layer] = c
            t = run(trial)
            if t is None:
                print(f"layer {layer + 1} plan {c}: refused", flush=True)
                continue
            mark = ""
            if t < best_t - 0.0015:
                best, best_t, mark = trial, t, "  <- better"
            print(f"layer {layer + 1} plan {c}: {t:.4f} ms{mark}", flush=True)
        print(f"after layer {layer + 1}: {best} -> {best_t:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
