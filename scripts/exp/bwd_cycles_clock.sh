#!/bin/bash
# backward kernel: event time + cycles for the shipped build and timing-only variants
R=$GRAFT_REPO_ROOT
for v in shipped b_noadd b_noatom b_neither; do
  bash scripts/pmc_bwd_var.sh $v "e" > gpurun_out/r05_pmcb_$v.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for v in shipped b_noadd b_noatom b_neither; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_bstats_$v -- python3 $R/scripts/exp/bench_variant.py $v --steps 3 --warmup 1 --no-cpu-baseline --no-check > $R/gpurun_out/r05_bstats_$v.log 2>&1
  f=$(ls $R/gpurun_out/r05_bstats_$v/*/*kernel_stats.csv | head -1); grep -i "k_bwd_brick" $f | cut -c1-200 > $R/gpurun_out/r05_bstats_$v.txt
done
cd $R
for v in shipped b_noadd b_noatom b_neither; do echo "== $v"; grep -A1 k_bwd gpurun_out/r05_pmcb_$v.txt | tail -1; cat gpurun_out/r05_bstats_$v.txt; done
