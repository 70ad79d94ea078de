#!/bin/bash
# one rocprofv3 --pmc pass over scripts/time_volgen.py (the fused 1x1 conv + forward); usage: pmc_conv.sh TAG COUNTER...
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/scripts/time_volgen.py > $out.log 2>&1
echo "pmc $tag exit $?"
