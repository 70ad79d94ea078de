#!/bin/bash
# usage: scripts/measure_shipped.sh <tag>   (GPU box, repo root): the reference's SHIPPED configuration (cfg/defaults.py:18,24-25,
# cfg/baseline.yaml:28-34: VOLUME_SIZE 16, DECONV_LAYERS 0 -> 2048 input channels, stride-32 maps of 12x12 / 8x8, 256 output channels, 4 views)
# forward bench lines, train-step lines and the kernel trace of one train step -> gpurun_out/<tag>_shipped/
tag=$1
out=gpurun_out/${tag}_shipped
mkdir -p $out
for feat in 12 8; do for b in 8 32; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --grid 16 --channels 256 --feat $feat --batch $b > $out/fwd_f${feat}_b$b.json 2> $out/fwd_f${feat}_b$b.err && python scripts/show_bench.py $out/fwd_f${feat}_b$b.json
  timeout -k 10 300 python bench.py --train-step --steps 20 --warmup 5 --grid 16 --channels 256 --in-channels 2048 --feat $feat --batch $b > $out/train_f${feat}_b$b.json 2> $out/train_f${feat}_b$b.err && cat $out/train_f${feat}_b$b.json
done; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- python3 $R/bench.py --train-step --steps 20 --warmup 5 --grid 16 --channels 256 --in-channels 2048 --feat 12 --batch 32 > $R/$out/stats.log 2>&1
cd $R
f=$(ls $out/stats/*/*kernel_stats.csv | head -1); cut -c1-160 $f | head -24
