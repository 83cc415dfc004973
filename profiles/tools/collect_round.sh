#!/bin/bash
# Everything the round's DESIGN.md numbers come from, for one workload: the rocprofv3 kernel statistics of the default bench
# command (kernel_stats.csv, by-grid table, the bench line measured under the profiler) and the PMC traffic of the sweeps.
# usage (on the GPU box, from the repo root): bash profiles/tools/collect_round.sh <tag> <workload> [commit]
#   -> gpurun_out/<tag>_<workload>_{kernel_stats.csv,by_grid.txt,bench_under_rocprof.json,pmc_summary.json} + traffic_<workload>.json
set -e
tag=$1; w=$2; commit=${3:-unknown}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_${tag}_$w -o s -- python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --workload $w > $out/${tag}_${w}_bench_under_rocprof.json 2> $out/${tag}_${w}_rocprof.log
cp $(find $out/kt_${tag}_$w -name "s_kernel_stats.csv" | head -1) $out/${tag}_${w}_kernel_stats.csv
python profiles/tools/trace_by_grid.py $(find $out/kt_${tag}_$w -name "s_kernel_trace.csv" | head -1) > $out/${tag}_${w}_by_grid.txt
rm -rf $out/kt_${tag}_$w
bash profiles/tools/collect_traffic.sh $w $out $commit
cp $out/pmc_${w}_summary.json $out/${tag}_${w}_pmc_summary.json
