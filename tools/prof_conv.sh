#!/bin/bash
# PMC passes over ONE conv shape and pass (tools/one_conv.py), each counter group in its own run (no trace domains beside
# --pmc).  usage: tools/prof_conv.sh <tag> cin cout k s H B mode     -> gpurun_out/pmc_<tag>_<group>/  and  gpurun_out/pmc_<tag>.txt
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
g1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
g2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
g3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM"
g4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE"
i=0
for g in "$g1" "$g2" "$g3" "$g4"; do
  i=$((i+1))
  rocprofv3 --pmc $g --output-format csv -d $out/pmc_${tag}_g$i -o p -- python3 $root/tools/one_conv.py "$@" 6 > $out/pmc_${tag}_g$i.log 2>&1 || echo "group $i failed (see log)"
done
python3 $root/tools/pmc_kernels.py $out/pmc_${tag}_g1 $out/pmc_${tag}_g2 $out/pmc_${tag}_g3 $out/pmc_${tag}_g4 --match conv_ > $out/pmc_${tag}.txt
