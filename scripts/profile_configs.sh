#!/bin/bash
# usage: scripts/profile_configs.sh <tag>   (GPU box, repo root): rocprofv3 --kernel-trace --stats of BASELINE configs[1], [3], [4] (per-GPU
# shards; forward + backward legs of bench.py) -> gpurun_out/<tag>_cfgstats/<cfg>_kernel_stats.csv
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_cfgstats
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { t=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$t -- python3 $R/bench.py --no-cpu-baseline --no-check --steps 10 --warmup 3 "$@" > $out/$t.log 2>&1
  f=$(ls $out/$t/*/*kernel_stats.csv | head -1); cp $f $out/${t}_kernel_stats.csv; echo "== $t"; cut -c1-140 $f | head -8; }
run c1 --batch 8 --grid 32 --channels 256 --views 4
run c3 --batch 16 --grid 64 --channels 256 --views 8
run c4 --batch 16 --grid 128 --channels 512 --views 4
