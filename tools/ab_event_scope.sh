set -e
mkdir -p gpurun_out/evscope
for sc in system device nofence system device; do
  DML_EVENT_SCOPE=$sc timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 > gpurun_out/evscope/$sc.$RANDOM.json 2>gpurun_out/evscope/err.log
done
for f in gpurun_out/evscope/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d["ms_per_step"], d["windows"])
PY
done
