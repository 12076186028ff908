# rocprofv3 kernel trace of the flagship step -> per-stream timeline + kernel table under gpurun_out/.
# usage: bash tools/profile_step.sh <tag> [bench args...]   (kernel-tuning env vars pass through)
set -e
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$tag -o $tag -- python3 $root/bench.py --steps 50 --warmup 4 --windows 1 "$@" > /tmp/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_timeline.py $db > $root/gpurun_out/${tag}_timeline.txt
python3 $root/tools/rocpd_summary.py $db --iters ${ITERS:-124} > $root/gpurun_out/${tag}_kernels.txt
tail -n 3 $root/gpurun_out/${tag}_timeline.txt
