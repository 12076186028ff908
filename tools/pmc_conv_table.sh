# MFMA / LDS / wait ratios of every conv kernel of the flagship step, each alone on the GPU.
# usage: bash tools/pmc_conv_table.sh <tag>     -> gpurun_out/<tag>_conv_kernels_pmc.txt
set -e
tag=$1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcc_$tag
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES --kernel-trace --output-format rocpd -d /tmp/pmcc_$tag -o p -- python3 $root/tools/pmc_conv.py > $root/gpurun_out/pmcc_$tag.log 2>&1 || { tail -20 $root/gpurun_out/pmcc_$tag.log; exit 1; }
db=$(find /tmp/pmcc_$tag -name "*.db" | head -1)
python3 $root/tools/rocpd_pmc.py $db --filter conv --derive > $root/gpurun_out/${tag}_conv_kernels_pmc.txt
cat $root/gpurun_out/${tag}_conv_kernels_pmc.txt
