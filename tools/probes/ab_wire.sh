#!/bin/bash
# one-rank cost of the bf16 wire against the fp32 wire (same box, alternating)
out=gpurun_out/ab_wire.log
: > $out
run() { timeout -k 10 150 python bench.py --force-comm --steps 300 --warmup 30 --windows 2 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['ms_per_step'], r.get('windows'))" | tee -a $out; }
run --grad-comm-dtype fp32
run --grad-comm-dtype bf16
run --grad-comm-dtype fp32
run --grad-comm-dtype bf16
run --grad-comm-dtype fp32 --no-batch-norm
run --grad-comm-dtype bf16 --no-batch-norm
