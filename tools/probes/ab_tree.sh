#!/bin/bash
# A/B on one box: HEAD's sources (git archive shipped as gpurun_out/head.tar by the caller is not
# available on the box, so: the tree as built = B, against a copy with a file list reverted from
# ./.ab_base/) - usage: ab_tree.sh   (expects ./.ab_base/<same layout> holding the A versions)
set -e
out=gpurun_out/ab_tree.log
: > $out
run() { (cd $1 && timeout -k 10 150 python bench.py --steps 300 --warmup 30 $EXTRA) 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', r['ms_per_step'])" | tee -a $out; }
rm -rf /tmp/a && mkdir /tmp/a && cp -r bench.py build.py distributed_machine_learning_amd /tmp/a/
cp -r .ab_base/. /tmp/a/
(cd /tmp/a && timeout -k 10 600 python build.py > /tmp/a/build.log 2>&1; tail -1 /tmp/a/build.log) | tee -a $out
run /tmp/a A
run . B
run /tmp/a A
run . B
