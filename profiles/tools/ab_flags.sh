#!/bin/bash
# A/B of compile-time variants on ONE box: usage ab.sh "<flags A>" "<flags B>" ... ; prints the solve times of the workloads per variant
out=gpurun_out/ab.txt
for flags in "$@"; do
  DOTS_HIPCC_FLAGS="$flags" python -m dots_socp_amd.build > /dev/null 2>&1 || { echo "build failed: $flags" >> $out; exit 1; }
  for rep in 1 2; do
    for w in ${AB_WORKLOADS:-knot sphere10k torus100k torus65k_T127}; do
      echo -n "[$flags] " >> $out
      python profiles/tools/front_tune.py $w 2>&1 | grep solve >> $out
    done
  done
done
python -m dots_socp_amd.build > /dev/null 2>&1
cat $out
