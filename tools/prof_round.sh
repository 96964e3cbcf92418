#!/bin/bash
# Round profiles (run on the GPU box from the repo root): rocprofv3 kernel statistics of the benchmark command, the two
# HBM-traffic counter passes, the MFMA-busy pass (each --pmc pass alone: no trace domains beside it), and the per-layer
# table.  Raw output under gpurun_out/prof_<tag>/, summaries under gpurun_out/ (copy them into profiles/).
set -e
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --no-infer --no-kernel-events"
echo "[prof] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -o s -- $B --steps 15 --warmup 5 > $out/prof_${tag}_stats.log 2>&1
python3 $root/tools/kernel_stats.py $out/prof_${tag}_stats 20 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-events (per step = / 20 steps incl. warm-up)" > $out/${tag}_kernel_stats.csv
echo "[prof] FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/prof_${tag}_f -o f -- $B --steps 2 --warmup 1 > $out/prof_${tag}_f.log 2>&1
echo "[prof] WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/prof_${tag}_w -o w -- $B --steps 2 --warmup 1 > $out/prof_${tag}_w.log 2>&1
python3 $root/tools/pmc_summary.py $out/prof_${tag}_f $out/prof_${tag}_w > $out/${tag}_pmc_traffic.txt
echo "[prof] MFMA busy"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/prof_${tag}_m -o m -- $B --steps 2 --warmup 1 > $out/prof_${tag}_m.log 2>&1
python3 $root/tools/mfma_util.py $out/prof_${tag}_m > $out/${tag}_mfma_util.txt
echo "[prof] per-layer table"; python3 $root/tools/bench_conv.py > $out/${tag}_conv_layers.txt 2>&1
rm -rf $out/prof_${tag}_stats $out/prof_${tag}_f $out/prof_${tag}_w $out/prof_${tag}_m
echo "[prof] done"
