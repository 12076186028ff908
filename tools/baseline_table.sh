# 1-GPU numbers of every reference configuration (parts 1 / 2a / 2b / 3) in both precisions,
# one JSON line each under gpurun_out/baseline_table/ plus a summary on stdout.
set -e
out=gpurun_out/baseline_table
mkdir -p $out
for dtype in bf16 fp32; do
  for cfg in "part1" "part2a --force-comm" "part2b --force-comm" "part3" "part3 --force-comm"; do
    tag=$(echo "${dtype}_$cfg" | tr -d ' -')
    timeout -k 10 200 python bench.py --dtype $dtype --mode $cfg --steps 200 --warmup 30 --windows 5 2>/dev/null > $out/$tag.json
    python tools/baseline_row.py < $out/$tag.json
  done
done
# stock PyTorch-ROCm on the same GPU (the torch backend: eager autograd, MIOpen / hipBLASLt)
for dtype in bf16 fp32; do
  for cfg in "part1" "part3"; do
    tag=$(echo "torch_${dtype}_$cfg" | tr -d ' -')
    timeout -k 10 300 python bench.py --backend torch --dtype $dtype --mode $cfg --steps 60 --warmup 10 --windows 2 2>/dev/null > $out/$tag.json
    python tools/baseline_row.py < $out/$tag.json
  done
done
