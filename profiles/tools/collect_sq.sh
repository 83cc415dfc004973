#!/bin/bash
# Occupancy / stall / cache counters of the kernels of one workload: three rocprofv3 --pmc passes (SQ wave accounting,
# SQ instruction mix, TCC hit / miss), each with --kernel-trace only (gpurun refuses --pmc together with the api traces).
# usage (on the GPU box, from the repo root): bash profiles/tools/collect_sq.sh <workload> <out dir>
set -e
w=$1; out=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {   # <tag> <counters...>
  tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/sq_${w}_$tag -o s -- python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --steps 20 --warmup 5 --workload $w > $out/sq_${w}_$tag.json 2> $out/sq_${w}_$tag.log
}
run waves SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run insts SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python profiles/tools/sq_summary.py $out/sq_${w}_waves/s_counter_collection.csv $out/sq_${w}_insts/s_counter_collection.csv $out/sq_${w}_tcc/s_counter_collection.csv > $out/sq_${w}_summary.txt
rm -rf $out/sq_${w}_waves $out/sq_${w}_insts $out/sq_${w}_tcc
