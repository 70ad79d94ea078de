#!/bin/bash
# on the GPU box: scripts/exp/run_variants4.sh <outfile> <variant>...  -> one line per variant: kernel ms (avg / min), step ms, max-abs vs the C oracle (sample 0)
out=$1; shift
for v in "$@"; do
  python3 scripts/exp/bench_variant.py $v --no-cpu-baseline --no-backward --steps 20 --warmup 5 $BENCH_FLAGS > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err || { echo "$v FAILED" >> $out; tail -3 gpurun_out/exp_$v.err >> $out; continue; }
  python3 - "$v" gpurun_out/exp_$v.json >> $out <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j["roofline"]
print("%-14s kernel %.4f ms (min %.4f)  frac %.4f  step %.4f ms  auto %.4f  max_abs %.2e" % (sys.argv[1], r["kernel_ms"], r["kernel_ms_min"], r["frac"], j["ms_per_step"], j.get("auto_call", {}).get("ms", 0), j.get("max_abs_vs_ref", -1)))
PY
done
cat $out
