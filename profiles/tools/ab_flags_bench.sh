#!/bin/bash
# A/B of compile-time variants on ONE box by bench.py's steady-state rate: usage ab_flags_bench.sh "<flags A>" "<flags B>" ...
out=gpurun_out/ab_flags_bench.txt
for rep in 1 2; do
for flags in "$@"; do
  DOTS_HIPCC_FLAGS="$flags" python -m dots_socp_amd.build > /dev/null 2>&1 || { echo "build failed: $flags" >> $out; exit 1; }
  for w in ${AB_WORKLOADS:-knot sphere10k torus100k}; do
    DOTS_HIPCC_FLAGS="$flags" python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --no-alternatives --workload $w > /tmp/ab_bench.json 2>/dev/null || { echo "[$flags] $w FAILED" >> $out; continue; }
    python - "$flags" $w >> $out <<'PY'
import json, sys
d = json.load(open("/tmp/ab_bench.json"))
print(f"[{sys.argv[1]}] {sys.argv[2]}: steady {d['steady_state']['iterations_per_s']:.1f} it/s, value {d['value']:.1f}, solve {d['roofline']['ms_per_solve']*1e3:.1f} us")
PY
  done
done
done
python -m dots_socp_amd.build > /dev/null 2>&1
cat $out
