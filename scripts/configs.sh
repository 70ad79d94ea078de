# BASELINE.json configs 1-4 (per-GPU shards): forward kernel + step + backward, one JSON per config.
# bench.py asks mvhmr_unproject_query_variant which variant the geometry gate selects (configs[1]: gather, the others: brick).
run() { tag=$1; shift; timeout -k 10 500 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/cfg_$tag.json && python scripts/show_bench.py gpurun_out/cfg_$tag.json && python -c "
import json; d=json.load(open('gpurun_out/cfg_$tag.json')); print('   bwd', d.get('backward'))"; }
run c1 --batch 8 --grid 32 --channels 256 --views 4 &&
run c2 --batch 32 --grid 64 --channels 256 --views 4 &&
run c2h --batch 32 --grid 64 --channels 256 --views 4 --dtype f16 &&
run c3 --batch 16 --grid 64 --channels 256 --views 8 &&
run c4 --batch 16 --grid 128 --channels 512 --views 4
