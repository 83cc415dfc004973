set -e
C=$1
for w in knot sphere10k torus100k; do bash profiles/tools/collect_round.sh r03 $w $C; done
bash profiles/tools/collect_sq.sh torus100k gpurun_out
# torus65k_T127: kernel statistics only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_r03_T127 -o s -- python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --workload torus65k_T127 > gpurun_out/r03_torus65k_T127_bench_under_rocprof.json 2> gpurun_out/r03_T127_rocprof.log
python profiles/tools/trace_by_grid.py $(find gpurun_out/kt_r03_T127 -name "s_kernel_trace.csv" | head -1) > gpurun_out/r03_torus65k_T127_by_grid.txt
rm -rf gpurun_out/kt_r03_T127
# the staged right-hand side (study): traffic of the launch it replaces (the switch is exported: no `env` hop under rocprofv3)
export DOTS_RHS_TILES=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_tiles_$c -o s -- python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --steps 20 --warmup 5 --workload torus100k > gpurun_out/pmc_tiles_$c.json 2> gpurun_out/pmc_tiles_$c.log
done
unset DOTS_RHS_TILES
calib=$(python -c "import json;print(json.load(open('gpurun_out/pmc_tiles_FETCH_SIZE.json'))['roofline']['pmc_calibration_bytes_each_way'])")
python profiles/tools/pmc_summary.py gpurun_out/pmc_tiles_FETCH_SIZE/s_counter_collection.csv gpurun_out/pmc_tiles_WRITE_SIZE/s_counter_collection.csv gpurun_out/r03_torus100k_rhs_tiles_pmc_summary.json --calib k_calib_stream $calib $calib > /dev/null
rm -rf gpurun_out/pmc_tiles_FETCH_SIZE gpurun_out/pmc_tiles_WRITE_SIZE
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_style.json 2> gpurun_out/r03_bench_driver_style.log
python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r03_bench_n2_gloo_one_gpu.json 2> gpurun_out/r03_bench_n2.log
