#!/bin/bash
# HBM traffic of the direct solve from the PMC counters: two rocprofv3 passes per workload (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), calibrated on the k_calib_stream launch of the same run.
# usage (on the GPU box, from the repo root): bash profiles/tools/collect_traffic.sh <workload> <out dir> [commit]
# (.git does not travel to the GPU box: pass the commit the measurement belongs to)
set -e
w=$1; out=$2; commit=${3:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${w}_$c -o s -- python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --steps 20 --warmup 5 --workload $w > $out/pmc_${w}_$c.json 2> $out/pmc_${w}_$c.log
done
levels=$(python -c "import json;print(json.load(open('$out/pmc_${w}_FETCH_SIZE.json'))['roofline']['launches_per_solve'] // 2)")   # forward launches of one solve = bands of tree heights
calib=$(python -c "import json;print(json.load(open('$out/pmc_${w}_FETCH_SIZE.json'))['roofline']['pmc_calibration_bytes_each_way'])")
python profiles/tools/pmc_summary.py $out/pmc_${w}_FETCH_SIZE/s_counter_collection.csv $out/pmc_${w}_WRITE_SIZE/s_counter_collection.csv $out/pmc_${w}_summary.json --calib k_calib_stream $calib $calib --front-levels $levels > $out/pmc_${w}_summary.txt
python - <<PY
import json
s = json.load(open("$out/pmc_${w}_summary.json"))
b = json.load(open("$out/pmc_${w}_FETCH_SIZE.json"))
json.dump({"bytes_per_solve": s["front_solve"]["bytes_per_solve"], "read_bytes_per_solve": s["front_solve"]["read_bytes_per_solve"],
           "written_bytes_per_solve": s["front_solve"]["written_bytes_per_solve"],
           "algorithmic_bytes_per_solve": b["roofline"]["algorithmic_bytes_per_solve"],
           "factor_bytes_per_solve_as_installed": b["roofline"]["factor"].get("bytes_per_solve_as_installed"),
           "bands": b["roofline"]["factor"].get("bands"),
           "kernel_sources_sha256": b["roofline"]["kernel_sources_sha256"],
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) of 'bench.py --workload $w --steps 20 --warmup 5 --no-configs', commit $commit, "
                     "calibrated on k_calib_stream of the same run (profiles/tools/collect_traffic.sh)"},
          open("$out/traffic_${w}.json", "w"), indent=1)
PY
rm -rf $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE
