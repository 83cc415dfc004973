#!/bin/bash
# usage: window_compare.sh <out file> <reps> <tree or VAR=val:tree> ... : profiles/tools/window_compare.py on every tree, interleaved, on this box
out=$1; reps=$2; shift; shift
for rep in $(seq $reps); do for t in "$@"; do
  if [[ "$t" == *:* ]]; then setting=${t%%:*}; tree=${t##*:}; else setting="DOTS_NOOP_=1"; tree=$t; fi
  if [[ "$setting" == "DOTS_NOOP_=1" ]]; then python profiles/tools/window_compare.py $tree >> $out 2>/dev/null || echo "$t FAILED" >> $out
  else echo -n "[$setting] " >> $out; env $setting python profiles/tools/window_compare.py $tree >> $out 2>/dev/null || echo "$t FAILED" >> $out; fi
done; done
cat $out
