# schedule knobs re-measured with the tall conv tiles and without packed fp32 pairs (one box, back to back)
out=gpurun_out/knobs_sweep; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 300 --warmup 20 --windows 2 "$@" 2>>$out/err.log > $out/$tag.json || { echo "$tag failed"; return; }
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-28s %.4f  median %.4f' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step']))"; }
run base
run fused_opt --fused-optimizer 1
for late in 1 3 5 6 7; do run update_overlap_$late --update-overlap $late; done
DML_WGRAD_TUNED=0 run untuned_splits
DML_WGRAD_WAVES=4 run wgrad_4waves
DML_WGRAD_KSTEP=32 run wgrad_kstep32
for t in 192 288 360; do DML_WGRAD_TARGET=$t run wgrad_target_$t; done
run group2 --wgrad-group 2
DML_BN_EPI_SUMS=0 run no_epi_sums
run base2
