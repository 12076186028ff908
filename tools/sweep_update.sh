# effect of the deep-layer update overlap (first layer updated on the side stream) on the step
set -e
show='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("windows",{}).get("median_ms_per_step"), d.get("params_in_sync"))'
for late in 0 1 2 3 4 5 6; do
  for args in "" "--force-comm" "--mode part2a --force-comm" "--no-batch-norm"; do
    echo "update_overlap=$late args=[$args]"
    timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 3 --update-overlap $late $args 2>/dev/null | python -c "$show"
  done
done
