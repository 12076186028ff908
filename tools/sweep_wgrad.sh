# A/B of the weight-gradient kernel choice inside the full training step (1 GPU):
# per-tap kernel (variant 1) against the tap-resident one (variant 2) for a few block targets
set -e
for v in "1 0,2 0" "2 64,2 128" "2 64,2 192" "2 64,2 256" "2 128,3 256"; do
  set -- $v
  for args in "" "--force-comm" "--no-batch-norm"; do
    echo "variant=$1 wgrad9=$2 target=$3 args=[$args]"
    DML_WGRAD_VARIANT=$1 DML_WGRAD9=$2 DML_WGRAD_TARGET=$3 timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 $args 2>/dev/null | python tools/baseline_row.py
  done
done
