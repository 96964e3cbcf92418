#!/bin/bash
# PMC groups (own passes) over the benchmark step for ONE kernel name: tools/prof_kernel_in_step.sh <tag> <kernel substring>
# -> gpurun_out/pmc_<tag>.txt   (counter collection serialises the kernels: SOLO figures of that kernel on real step data)
set -e
tag=$1; match=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
g1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
g2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
g3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
i=0
dirs=""
for g in "$g1" "$g2" "$g3"; do
  i=$((i+1))
  rocprofv3 --pmc $g --output-format csv -d $out/pmc_${tag}_g$i -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-kernel-events > $out/pmc_${tag}_g$i.log 2>&1 || echo "group $i failed (see log)"
  dirs="$dirs $out/pmc_${tag}_g$i"
done
python3 $root/tools/pmc_kernels.py $dirs --match "$match" > $out/pmc_${tag}.txt
rm -rf $dirs
