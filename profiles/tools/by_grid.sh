#!/bin/bash
# kernel statistics by (kernel, grid) of bench.py on the listed workloads: usage by_grid.sh <tag> <workload> ...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$w -o s -- python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --no-alternatives --workload $w > gpurun_out/$tag/${tag}_${w}_bench_under_rocprof.json 2> gpurun_out/$tag/${tag}_${w}_rocprof.log
  python profiles/tools/trace_by_grid.py $(find gpurun_out/kt_$w -name "s_kernel_trace.csv" | head -1) > gpurun_out/$tag/${tag}_${w}_by_grid.txt
  cp $(find gpurun_out/kt_$w -name "s_kernel_stats.csv" | head -1) gpurun_out/$tag/${tag}_${w}_kernel_stats.csv
  rm -rf gpurun_out/kt_$w
done
