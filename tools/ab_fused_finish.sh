# forward of the two 2x2 layers under BatchNorm: split-K finish + statistics + apply in one launch
# (DML_FUSED_FINISH_BN=1; DML_FINISH_BN_CPB = channels per block, 8 or 16) against
# finish-with-statistics + fold/apply (DML_FUSED_FINISH_BN=0)
out=gpurun_out/fused_finish; mkdir -p $out
for cpb in 8 16; do
DML_FUSED_FINISH_BN=1 DML_FINISH_BN_CPB=$cpb timeout -k 10 500 python -u -m pytest tests/test_engine_gpu.py -x -q --timeout 200 --timeout-method thread -k "layerwise or training or forty or reproducible or side_stream or eval or checkpoint or ddp" > $out/tests$cpb.log 2>&1 || { tail -40 $out/tests$cpb.log; exit 1; }
tail -1 $out/tests$cpb.log
done
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 300 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-24s %.4f  median %.4f  loss %s' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step'], d['config']['final_loss']))"; }
for i in 1 2 3; do
DML_FUSED_FINISH_BN=0 run separate$i
DML_FUSED_FINISH_BN=1 DML_FINISH_BN_CPB=16 run fused16_$i
DML_FUSED_FINISH_BN=1 DML_FINISH_BN_CPB=8 run fused8_$i
done
