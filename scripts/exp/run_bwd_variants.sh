#!/bin/bash
# on the GPU box: scripts/exp/run_bwd_variants.sh <outfile> <variant>...   -> one line per variant: backward ms per call
out=$1; shift
for v in "$@"; do
  python3 scripts/exp/bench_variant.py $v --no-cpu-baseline --no-check --steps 10 --warmup 3 $BENCH_FLAGS > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err || { echo "$v FAILED" >> $out; continue; }
  python3 - "$v" gpurun_out/exp_$v.json >> $out <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-14s backward %.3f ms  (forward kernel %.4f)" % (sys.argv[1], j["backward"]["ms"], j["roofline"]["kernel_ms"]))
PY
done
cat $out
