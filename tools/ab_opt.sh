# optimizer placement options re-measured under the flag-word hand-offs
out=gpurun_out/optab; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-28s %.4f  median %.4f' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step']))"; }
run base
run fusedopt --fused-optimizer 1
run update_auto --update-overlap -1
run update_3 --update-overlap 3
run update_5 --update-overlap 5
run base2
DML_OPT_SLICE_FROM=2 run fusedopt_from2 --fused-optimizer 1
DML_OPT_SLICE_FROM=7 run fusedopt_from7 --fused-optimizer 1
