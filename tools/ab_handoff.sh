# A/B of the main -> side hand-off mechanism inside the training step (bench.py, 200 steps x 4 windows)
set -e
out=gpurun_out/handoff; mkdir -p $out
timeout -k 10 300 python -u -m pytest tests/test_engine_gpu.py -x -q --timeout 120 --timeout-method thread -k "side_stream or overlap or graph or reducer or zero" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for mode in event value event value; do
  for extra in "" "--no-batch-norm --mode part1" "--force-comm"; do
    tag=$(echo "$mode$extra" | tr -d ' -')
    DML_HANDOFF=$mode timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 $extra > $out/$tag.$RANDOM.json 2>>$out/err.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/handoff/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["ms_per_step"], d["windows"]["median_ms_per_step"])
PY
