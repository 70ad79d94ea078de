#!/bin/bash
# GPU box, repo root: socket power and shader clock (rocm-smi, twice a second) beside a steady loop of the forward kernel, the backward and a
# plain device copy -- scripts/exp/power_trace.sh <outfile>.  Direct evidence for (or against) "the forward is power-limited".
out=$1; : > $out
rocm-smi --showmaxpower >> $out 2>&1
for what in copy fwd bwd; do
  echo "== $what" >> $out
  python3 scripts/exp/power_loop.py $what 8 >> $out 2>&1 &
  pid=$!
  sleep 3
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level" | sed 's/^GPU\[0\]\s*: //' | tr '\n' '|' >> $out
    echo >> $out
    sleep 0.5
  done
  wait $pid
done
cat $out
