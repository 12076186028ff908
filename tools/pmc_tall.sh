# Hardware counters of one conv kernel alone on the GPU (tools/pmc_one.py), two passes.
# usage: bash tools/pmc_tall.sh <tag> <layer> <plan>      e.g.  bash tools/pmc_tall.sh c4 2 256,64,1,0,20
set -e
tag=$1; layer=$2; plan=$3
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
export LAYER=$layer KIND=fwd PLAN=$plan
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmct_${tag}_$i
  timeout -s KILL 120 rocprofv3 --pmc $pass --kernel-trace --output-format rocpd -d /tmp/pmct_${tag}_$i -o p -- python3 $root/tools/pmc_one.py > $root/gpurun_out/pmct_${tag}_$i.log 2>&1 || { tail -20 $root/gpurun_out/pmct_${tag}_$i.log; exit 1; }
  db=$(find /tmp/pmct_${tag}_$i -name "*.db" | head -1)
  python3 $root/tools/rocpd_pmc.py $db --filter conv > $root/gpurun_out/pmct_${tag}_$i.txt
  cat $root/gpurun_out/pmct_${tag}_$i.txt
done
