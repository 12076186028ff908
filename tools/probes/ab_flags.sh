#!/bin/bash
# A/B on one box: the tree as built against a copy rebuilt with other compiler flags.
# usage: ab_flags.sh '<sed expression applied to buildinfo.py>'
set -e
out=gpurun_out/ab_flags.log
: > $out
run() { (cd $1 && timeout -k 10 150 python bench.py --steps 300 --warmup 30) 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', r['ms_per_step'])" | tee -a $out; }
run . A
rm -rf /tmp/b && mkdir /tmp/b && cp -r bench.py build.py distributed_machine_learning_amd /tmp/b/
(cd /tmp/b && sed -i "$1" distributed_machine_learning_amd/buildinfo.py && grep -n "ARCH =\|COMMON_FLAGS = " -A1 distributed_machine_learning_amd/buildinfo.py | head -8 && timeout -k 10 600 python build.py > /tmp/b/build.log 2>&1; tail -1 /tmp/b/build.log) | tee -a $out
run /tmp/b B
run . A
run /tmp/b B
