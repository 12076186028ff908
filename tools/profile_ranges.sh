# rocprofv3 marker + kernel trace of the flagship step with the engine's phase ranges on.
# usage: bash tools/profile_ranges.sh <tag> [bench args...]
set -e
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rng_$tag
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --output-format rocpd -d /tmp/rng_$tag -o $tag -- python3 $root/bench.py --steps 100 --warmup 10 --windows 0 --profile-ranges "$@" > /tmp/rng_$tag.log 2>&1 || { tail -n 20 /tmp/rng_$tag.log; exit 1; }
db=$(find /tmp/rng_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_ranges.py $db > $root/gpurun_out/${tag}_ranges.txt
cat $root/gpurun_out/${tag}_ranges.txt
