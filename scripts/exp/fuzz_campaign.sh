#!/bin/bash
# a hardening run of scripts/fuzz_parity.py over many seeds (GPU box, repo root): scripts/exp/fuzz_campaign.sh <first seed> <count>
# Stops at the first failing seed and exits non-zero (the old form piped python into tail, whose status hid every failure).
set -o pipefail
s0=$1; n=$2; out=gpurun_out/fuzz_campaign_$s0.txt; : > $out
for ((s = s0; s < s0 + n; ++s)); do
  case $((s % 4)) in 0) extra="";; 1) extra="--dtype f16";; 2) extra="--dtype bf16";; 3) extra="--big";; esac
  tmp=$(mktemp)
  timeout -k 10 300 python scripts/fuzz_parity.py --seed $s --cases 40 $extra > $tmp 2>&1
  rc=$?
  tail -1 $tmp >> $out
  if [ $rc -ne 0 ]; then echo "seed $s FAILED rc=$rc ($extra)" >> $out; tail -5 $tmp >> $out; rm -f $tmp; cat $out; exit 1; fi
  rm -f $tmp
done
# every line must be a success line
bad=$(grep -vc "^seed [0-9]*: [0-9]* runs" $out)
cat $out
if [ "$bad" -ne 0 ]; then echo "$bad line(s) are not success lines"; exit 1; fi
