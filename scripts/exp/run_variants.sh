#!/bin/bash
# on the GPU box: scripts/exp/run_variants.sh <outfile> <variant>...   -> one line per variant: kernel ms, step ms
out=$1; shift
for v in "$@"; do
  python3 scripts/exp/bench_variant.py $v --no-cpu-baseline --no-check --no-backward --steps 20 --warmup 5 $BENCH_FLAGS > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err || { echo "$v FAILED" >> $out; continue; }
  python3 - "$v" gpurun_out/exp_$v.json >> $out <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-12s kernel %.4f ms  step %.4f ms  auto %.4f" % (sys.argv[1], j["roofline"]["kernel_ms"], j["ms_per_step"], j.get("auto_call", {}).get("ms", 0)))
PY
done
cat $out
