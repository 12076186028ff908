# in-step A/B of the per-tap weight-gradient kernel's K step (64 against 32 pixels) and block targets
set -e
for v in "64 0" "32 0" "32 432" "32 576" "32 288"; do
  set -- $v
  echo "kstep=$1 target=$2"
  DML_WGRAD_KSTEP=$1 DML_WGRAD_TARGET=$2 timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 2>/dev/null | python tools/baseline_row.py
done
