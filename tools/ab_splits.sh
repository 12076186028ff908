# A/B of per-layer pixel-split lists of the weight-gradient kernels inside the full step, three modes
set -e
for args in "" "--force-comm" "--mode part1"; do
  for v in "8 40,10,10,2,3,1,1" "8 43,11,11,3,3,1,1" "8 39,10,10,2,2,1,1" "8 32,8,8,2,2,1,1" "4 40,10,10,2,3,1,1" "4 43,11,11,3,3,1,1" "8 0"; do
    set -- $v
    echo "waves=$1 splits=$2 args=[$args]"
    DML_WGRAD_WAVES=$1 DML_WGRAD_SPLITS=$2 timeout -k 10 120 python bench.py --steps 300 --warmup 30 --windows 3 $args 2>/dev/null | python tools/baseline_row.py
  done
done
