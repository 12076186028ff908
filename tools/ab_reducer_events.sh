out=gpurun_out/reducer_ev; mkdir -p $out
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-28s %.4f  median %.4f' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step']))"; }
run nocomm
run forcecomm --force-comm
DML_REDUCER_EVENT_SCOPE=ready-nofence run ready_nofence --force-comm
DML_REDUCER_EVENT_SCOPE=done-nofence run done_nofence --force-comm
DML_REDUCER_EVENT_SCOPE=ready-nofence,done-nofence run both_nofence --force-comm
run noinline --force-comm --inline-last-bucket 0
DML_REDUCER_EVENT_SCOPE=ready-nofence,done-nofence run noinline_both_nofence --force-comm --inline-last-bucket 0
