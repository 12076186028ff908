# fixed cost of the reducer against the number of buckets (one rank, multi-rank schedule)
out=gpurun_out/buckets; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-24s %.4f  median %.4f  %s' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step'], (d['config'].get('sync_plan') or {}).get('buckets_mb')))"; }
run nocomm
for cap in 2 5 9 10 25; do run cap$cap --force-comm --inline-last-bucket 0 --bucket-cap-mb $cap; done
run cap10_group2 --force-comm --inline-last-bucket 0 --wgrad-group 2
run cap9_first9 --force-comm --inline-last-bucket 0 --bucket-cap-mb 10 --first-bucket-cap-mb 9
