# backward-data plan of the fifth conv layer (256 -> 512 @ 4x4): default (tap split, position-major) against the
# 64 x 64 patch-resident kernel, which owns its pixel tiles and so leaves the BatchNorm backward sums in its epilogue
out=gpurun_out/dgrad_l4; mkdir -p $out
PLAN="0,0,0,64:64:1:0:16"
DML_DGRAD_PLANS=$PLAN timeout -k 10 400 python -u -m pytest tests/test_engine_gpu.py -x -q --timeout 120 --timeout-method thread -k "layerwise or training_steps or forty or 40 or reproducible or side_stream" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 300 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-24s %.4f  median %.4f' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step']))"; }
for i in 1 2 3; do
run base$i
DML_DGRAD_PLANS=$PLAN run patch$i
done
run base_comm --force-comm --inline-last-bucket 0
DML_DGRAD_PLANS=$PLAN run patch_comm --force-comm --inline-last-bucket 0
