# a few schedule knobs re-measured under the flag-word hand-offs (same box, back to back)
out=gpurun_out/knobs; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 300 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-24s %.4f  median %.4f' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step']))"; }
run base
DML_BN_EPI_SUMS=0 run no_epi_sums
run newstream --compute-stream new
DML_WGRAD_TUNED=0 run untuned_splits
DML_WGRAD_WAVES=4 run wgrad_4waves
run base2
