# bucket departures: event pair against flag word + wait-value (one rank, RCCL forced on)
out=gpurun_out/reducer_handoff; mkdir -p $out
timeout -k 10 400 python -u -m pytest tests/test_engine_gpu.py tests/test_multirank_gpu.py -x -q --timeout 120 --timeout-method thread -k "ddp or reducer or overlap or bench" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-28s %.4f  median %.4f  %s' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step'], (d['config'].get('sync_plan') or {}).get('bucket_departure')))"; }
run nocomm
DML_REDUCER_HANDOFF=event run event --force-comm
run value --force-comm
DML_REDUCER_HANDOFF=event run event_noinline --force-comm --inline-last-bucket 0
run value_noinline --force-comm --inline-last-bucket 0
DML_REDUCER_HANDOFF=event run event_fp32 --force-comm --dtype fp32
run value_fp32 --force-comm --dtype fp32
