# fp32 engine: hand-off mechanism A/B, after the engine and multi-rank tests
set -e
out=gpurun_out/handoff_f32; mkdir -p $out
timeout -k 10 400 python -u -m pytest tests/test_engine_f32_gpu.py tests/test_multirank_gpu.py -x -q --timeout 120 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for mode in event value event value; do
  for extra in "" "--no-batch-norm --mode part1"; do
    tag=$(echo "$mode$extra" | tr -d ' -')
    DML_HANDOFF=$mode timeout -k 10 120 python bench.py --dtype fp32 --steps 100 --warmup 10 --windows 3 $extra > $out/$tag.$RANDOM.json 2>>$out/err.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/handoff_f32/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["ms_per_step"], d["windows"]["median_ms_per_step"])
PY
