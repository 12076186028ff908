# 1-GPU numbers of every reference configuration (parts 1 / 2a / 2b / 3) in both precisions,
# one JSON line each under gpurun_out/baseline_r2/ plus a summary on stdout.
set -e
out=gpurun_out/baseline_r2
mkdir -p $out
show='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d["config"]; print(f"{d[\"dtype\"]:5s} {c[\"mode\"]:7s} bn={int(\"BatchNorm\" in c[\"model\"])} sync={c[\"sync\"]:15s} {d[\"ms_per_step\"]:8.4f} ms/step {d[\"value\"]:10.1f} img/s  median {d[\"windows\"][\"median_ms_per_step\"]:.4f}  host_loop {c[\"host_loop_ms_per_step\"]:.3f}")'
for dtype in bf16 fp32; do
  for cfg in "part1" "part2a --force-comm" "part2b --force-comm" "part3" "part3 --force-comm"; do
    tag=$(echo "${dtype}_$cfg" | tr -d ' -')
    timeout -k 10 200 python bench.py --dtype $dtype --mode $cfg --steps 200 --warmup 30 --windows 5 2>/dev/null > $out/$tag.json
    python -c "$show" < $out/$tag.json
  done
done
