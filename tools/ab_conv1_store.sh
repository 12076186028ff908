# first layer's forward: store order of the epilogue (DML_CONV1_STORE=interleaved|halves), in the step
out=gpurun_out/conv1store; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 300 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-24s %.4f  median %.4f' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step']))"; }
for i in 1 2 3; do
DML_CONV1_STORE=interleaved run interleaved$i
run halves$i
done
DML_CONV1_STORE=interleaved run interleaved_nobn --no-batch-norm --mode part1
run halves_nobn --no-batch-norm --mode part1
