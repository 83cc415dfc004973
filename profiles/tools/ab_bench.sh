#!/bin/bash
# A/B of run-time switches on ONE box by the steady-state rate of bench.py: usage ab_bench.sh "<VAR=val ...>" ... ; two rounds
out=gpurun_out/ab_bench.txt
for rep in 1 2; do
  for setting in "$@"; do
    for w in ${AB_WORKLOADS:-knot sphere10k torus100k}; do
      env $setting python bench.py --no-cpu-baseline --no-time-to-tol --no-configs --no-alternatives --workload $w > /tmp/ab_bench.json 2>/dev/null || { echo "[$setting] $w FAILED" >> $out; continue; }
      python - "$setting" $w >> $out <<'PY'
import json, sys
d = json.load(open("/tmp/ab_bench.json"))
print(f"[{sys.argv[1]}] {sys.argv[2]}: steady {d['steady_state']['iterations_per_s']:.1f} it/s, value {d['value']:.1f}, solve {d['roofline']['ms_per_solve']*1e3:.1f} us")
PY
    done
  done
done
cat $out
