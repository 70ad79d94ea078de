#!/bin/bash
# forward ms of the brick and of the gather kernels against the number of bricks in a launch (the AUTO rule's threshold)
# usage (GPU box, repo root): bash scripts/exp/sweep_brick_count.sh  -> gpurun_out/brick_count_sweep.txt
out=gpurun_out/brick_count_sweep.txt; mkdir -p gpurun_out; : > $out
for cfg in "32 128 96" "32 128 24" "32 32 96" "16 256 12" "24 128 48"; do
  set -- $cfg; S=$1; C=$2; F=$3
  for B in 2 4 8 16 32; do
    for var in brick gather; do
      line=$(python bench.py --grid $S --channels $C --feat $F --batch $B --variant $var --no-cpu-baseline --no-check --no-backward --steps 50 --warmup 10 2>/dev/null | tail -1) || exit 1
      echo "grid $S ch $C feat $F batch $B $var $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step %.4f kernel %s" % (d["ms_per_step"], d["roofline"].get("kernel")))')" >> $out
    done
  done
done
cat $out
