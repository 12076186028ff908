# quick A/B inside the full step: per-tap weight-gradient kernel against the tap-resident one
set -e
for v in "1 0,2 0" "2 64,2 128" "2 64,2 192" "2 64,2 256" "2 64,3 192"; do
  set -- $v
  echo "variant=$1 wgrad9=$2 target=$3"
  DML_WGRAD_VARIANT=$1 DML_WGRAD9=$2 DML_WGRAD_TARGET=$3 timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 2>/dev/null | python tools/baseline_row.py
done
