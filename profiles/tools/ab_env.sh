#!/bin/bash
# A/B of run-time switches on ONE box: usage ab_env.sh "<VAR=val ...>" "<VAR=val ...>" ... ; solve times of the workloads per setting, two rounds
out=gpurun_out/ab_env.txt
for rep in 1 2; do
  for setting in "$@"; do
    for w in ${AB_WORKLOADS:-knot sphere10k torus100k torus65k_T127}; do
      echo -n "[$setting] " >> $out
      env $setting python profiles/tools/front_tune.py $w 2>&1 | grep solve >> $out
    done
  done
done
cat $out
