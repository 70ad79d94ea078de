#!/bin/bash
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/scripts/bench_bwd.py --iters 1 > $out.log 2>&1
echo "pmc $tag exit $?"
