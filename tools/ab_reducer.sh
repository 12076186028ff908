# where the reducer's overhead at one rank comes from: bench.py --force-comm under variations
out=gpurun_out/reducer; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-28s %.4f  median %.4f  plan %s' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step'], (d['config'].get('sync_plan') or {}).get('buckets') if isinstance(d['config'].get('sync_plan'), dict) else d['config'].get('sync_plan')))"; }
run nocomm
run forcecomm --force-comm
run onebucket --force-comm --bucket-cap-mb 1000 --tail-bucket-cap-mb 1000
run noinline --force-comm --inline-last-bucket 0
run serialized --force-comm --serialize-streams
run nobn_nocomm --no-batch-norm
run nobn_forcecomm --no-batch-norm --force-comm
run bf16payload --force-comm --grad-comm-dtype bf16
