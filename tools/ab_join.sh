# the side stream's "done" join: event against flag word (bf16 engine)
out=gpurun_out/join; mkdir -p $out
timeout -k 10 400 python -u -m pytest tests/test_engine_gpu.py -x -q --timeout 120 --timeout-method thread -k "side_stream or flag_word or overlap or reproducible or zero" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() { tag=$1; shift; timeout -k 10 120 python bench.py --steps 200 --warmup 20 --windows 3 "$@" 2>>$out/err.log > $out/$tag.json || exit 1
  python -c "import json,sys; d=json.load(open('$out/$tag.json')); print('%-28s %.4f  median %.4f' % ('$tag', d['ms_per_step'], d['windows']['median_ms_per_step']))"; }
DML_JOIN=event run event
run value
DML_JOIN=event run event2
run value2
DML_JOIN=event run event_nobn --no-batch-norm --mode part1
run value_nobn --no-batch-norm --mode part1
DML_JOIN=event run event_comm --force-comm --inline-last-bucket 0
run value_comm --force-comm --inline-last-bucket 0
