# HBM traffic of the flagship step per kernel: two rocprofv3 --pmc passes (fetch, write) over a
# short bench run; tables under gpurun_out/.   usage: bash tools/pmc_step.sh <tag> [bench args...]
set -e
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE TCC_HIT_sum" "write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  set -- $pass
  name=$1; shift
  rm -rf /tmp/pmc_${tag}_$name
  timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format rocpd -d /tmp/pmc_${tag}_$name -o p -- python3 $root/bench.py --steps 10 --warmup 2 --windows 1 ${BENCH_ARGS} > $root/gpurun_out/${tag}_pmc_$name.log 2>&1 || { tail -20 $root/gpurun_out/${tag}_pmc_$name.log; exit 1; }
  db=$(find /tmp/pmc_${tag}_$name -name "*.db" | head -1)
  python3 $root/tools/rocpd_pmc.py $db --filter "" > $root/gpurun_out/${tag}_pmc_$name.txt
done
