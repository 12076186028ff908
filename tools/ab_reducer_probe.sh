out=gpurun_out/reducer_probe; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-28s %.4f  median %.4f' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step']))"; }
run nocomm
run forcecomm --force-comm
DML_REDUCER_PROBE=skip run skip --force-comm
DML_REDUCER_PROBE=marker run marker --force-comm
DML_REDUCER_PROBE=nodone run nodone --force-comm
run forcecomm2 --force-comm
