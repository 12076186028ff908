# A/B inside the full step: per-tap weight-gradient kernel with 4 / 8 waves per block, a few block targets, three modes
set -e
for args in "" "--force-comm" "--mode part1"; do
  for v in "4 0" "8 0" "8 320" "8 360" "8 400" "4 360"; do
    set -- $v
    echo "waves=$1 target=$2 args=[$args]"
    DML_WGRAD_WAVES=$1 DML_WGRAD_TARGET=$2 timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 $args 2>/dev/null | python tools/baseline_row.py
  done
done
