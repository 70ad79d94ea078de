#!/bin/bash
# usage: scripts/measure_round.sh <tag>   (GPU box, repo root): BASELINE configs 1-4 per-GPU shards + the two train-step lines
tag=$1
mkdir -p gpurun_out/${tag}_configs
run() { t=$1; shift; timeout -k 10 500 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/${tag}_configs/cfg_$t.json 2> gpurun_out/${tag}_configs/cfg_$t.err && python scripts/show_bench.py gpurun_out/${tag}_configs/cfg_$t.json && python -c "
import json; d=json.load(open('gpurun_out/${tag}_configs/cfg_$t.json')); print('   bwd', d.get('backward'))"; }
run c1 --batch 8 --grid 32 --channels 256 --views 4 &&
run c2 --batch 32 --grid 64 --channels 256 --views 4 &&
run c2h --batch 32 --grid 64 --channels 256 --views 4 --dtype f16 &&
run c3 --batch 16 --grid 64 --channels 256 --views 8 &&
run c4 --batch 16 --grid 128 --channels 512 --views 4 &&
timeout -k 10 500 python bench.py --train-step --steps 10 --warmup 3 > gpurun_out/${tag}_train_step.json 2> gpurun_out/${tag}_train_step.err && cat gpurun_out/${tag}_train_step.json &&
timeout -k 10 900 python bench.py --train-step --grid 128 --channels 512 --batch 16 --steps 5 --warmup 2 > gpurun_out/${tag}_train_step_c4.json 2> gpurun_out/${tag}_train_step_c4.err && cat gpurun_out/${tag}_train_step_c4.json
