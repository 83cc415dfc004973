# Everything DESIGN.md's round-4 numbers come from, on ONE box: usage  bash profiles/tools/collect_r04.sh <commit>
set -e
C=$1
for w in knot sphere10k torus100k; do bash profiles/tools/collect_round.sh r04 $w $C; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# torus65k_T127 and knot63: kernel statistics only
for w in torus65k_T127 knot63; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_r04_$w -o s -- python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --no-alternatives --workload $w > gpurun_out/r04_${w}_bench_under_rocprof.json 2> gpurun_out/r04_${w}_rocprof.log
  python profiles/tools/trace_by_grid.py $(find gpurun_out/kt_r04_$w -name "s_kernel_trace.csv" | head -1) > gpurun_out/r04_${w}_by_grid.txt
  cp $(find gpurun_out/kt_r04_$w -name "s_kernel_stats.csv" | head -1) gpurun_out/r04_${w}_kernel_stats.csv
  rm -rf gpurun_out/kt_r04_$w
done
# kernel timelines by kind of iteration (quiet / validation / penalty update)
for w in knot torus100k; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o s -- python profiles/tools/iteration_timeline.py run $w > gpurun_out/r04_tl_$w.log 2>&1
  python profiles/tools/iteration_timeline.py show $(find gpurun_out/tl -name s_kernel_trace.csv | head -1) 330 > gpurun_out/r04_${w}_timeline.txt
  rm -rf gpurun_out/tl
done
# the slab formulation on one rank (rehearsal): kernels by grid
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kts -o s -- python profiles/tools/slab_profile.py torus100k 100 > gpurun_out/r04_slab.log 2>&1
python profiles/tools/trace_by_grid.py $(find gpurun_out/kts -name s_kernel_trace.csv | head -1) > gpurun_out/r04_torus100k_slab_by_grid.txt
grep "slab iteration" gpurun_out/r04_slab.log > gpurun_out/r04_torus100k_slab_ms.txt
rm -rf gpurun_out/kts
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_style.json 2> gpurun_out/r04_bench_driver_style.log
python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r04_bench_n2_gloo_one_gpu.json 2> gpurun_out/r04_bench_n2.log
python bench.py --workload torus500k --no-cpu-baseline --no-configs --no-alternatives > gpurun_out/r04_torus500k_bench.json 2> gpurun_out/r04_torus500k.log
bash profiles/tools/window_compare.sh gpurun_out/r04_window_compare.txt 2 scratch/old_r2 scratch/old_r3 . > /dev/null
