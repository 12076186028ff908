# fused optimizer (SGD slices on the side stream inside backward): from which layer on per-layer slices pay
set -e
echo "fused off"; timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 2>/dev/null | python tools/baseline_row.py
for from in 1 2 3 4 5 6 7 8; do
  echo "fused on, slices from layer $from"
  DML_OPT_SLICE_FROM=$from timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 --fused-optimizer 1 2>/dev/null | python tools/baseline_row.py
done
echo "no BN: fused off"; timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 --no-batch-norm 2>/dev/null | python tools/baseline_row.py
for from in 4 6; do
  echo "no BN: fused on, slices from layer $from"
  DML_OPT_SLICE_FROM=$from timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 --fused-optimizer 1 --no-batch-norm 2>/dev/null | python tools/baseline_row.py
done
