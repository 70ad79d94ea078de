#!/bin/bash
# a hardening run of scripts/fuzz_parity.py over many seeds (GPU box, repo root): scripts/exp/fuzz_campaign.sh <first seed> <count>
s0=$1; n=$2; out=gpurun_out/fuzz_campaign_$s0.txt; : > $out
for ((s = s0; s < s0 + n; ++s)); do
  case $((s % 4)) in 0) extra="";; 1) extra="--dtype f16";; 2) extra="--dtype bf16";; 3) extra="--big";; esac
  timeout -k 10 300 python scripts/fuzz_parity.py --seed $s --cases 40 $extra 2>&1 | tail -1 >> $out || { echo "seed $s FAILED ($extra)" >> $out; tail -5 $out; exit 1; }
done
cat $out
