#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <counters...>   (run on the GPU box from the repo root; one counter set per call)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-backward > $out.log 2>&1
echo "pmc $tag exit $?"
