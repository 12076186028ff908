# A/B of side-stream priority and weight-gradient kernel choice inside the full step (1 GPU).
set -e
show='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("windows",{}).get("median_ms_per_step"))'
python -c "import torch; print(torch.cuda.get_device_name(0))"
for pr in "" "1" "-1"; do
for v in "1 0,2 0" "0 64,2 192" "0 64,2 256"; do
  set -- $v
  for args in "" "--force-comm"; do
    echo "side_priority=[$pr] variant=$1 wgrad9=$2 target=$3 args=[$args]"
    DML_SIDE_PRIORITY=$pr DML_WGRAD_VARIANT=$1 DML_WGRAD9=$2 DML_WGRAD_TARGET=$3 timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 $args 2>/dev/null | python -c "$show"
  done
done
done
