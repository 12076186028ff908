# in-step sweep of the per-tap weight-gradient kernel's block target (fewer blocks = fewer pixel splits)
set -e
for t in ${TARGETS:-100 144 192 224 256 288}; do
  echo "target=$t"
  DML_WGRAD_TARGET=$t timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 $@ 2>/dev/null | python tools/baseline_row.py
done
