# host time to enqueue one step (bench.py's probe) for the three sync modes at one GPU
set -e
show='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], "host probe", d["config"]["host_enqueue_ms_per_step"], "host loop", d["config"]["host_loop_ms_per_step"])'
for args in "" "--force-comm" "--mode part2a --force-comm" "--mode part2b --force-comm" "--no-batch-norm"; do
  echo "args=[$args]"
  timeout -k 10 120 python bench.py --steps 200 --warmup 30 --windows 1 $args 2>/dev/null | python -c "$show"
done
