#!/bin/bash
# Experimental builds of the forward kernel: scripts/exp/build_variant.sh <name> "<extra hipcc flags>"
# compiles the softmax forward unit (unproject_brick_fwd_m0.hip) with the flags and links it with the shipped objects into
# multiviewhmr_amd/lib_exp/<name>/libmvhmr_unproject.so (git-ignored, travels to the GPU box).  Not part of the product.
set -e
name=$1; shift
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/multiviewhmr_amd/csrc; LIB=$ROOT/multiviewhmr_amd/lib; OUT=$ROOT/multiviewhmr_amd/lib_exp/$name
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -I$ROOT/include -I$CS -Wall -Wno-unused-function"
UNIT=${UNIT:-unproject_brick_fwd_m0}
/opt/rocm/bin/hipcc $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -save-temps=obj -c $CS/$UNIT.hip -o $OUT/$UNIT.o 2> $OUT/resources.txt || { cat $OUT/resources.txt | grep -v remark | head -30; exit 1; }
OBJS=""
for o in $LIB/*.o; do b=$(basename $o); if [ "$b" != "$UNIT.o" ]; then OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmvhmr_unproject.so $OUT/$UNIT.o $OBJS
rm -f $OUT/*.bc $OUT/*.hipi $OUT/*.out $OUT/*.hipfb $OUT/*host*.s $OUT/*.o
grep -A12 "Function Name: .*${KERN:-k_fwd_brickILi0ELi4ELi1024EfLi2E}" $OUT/resources.txt | grep -E "VGPRs:|Spill|ScratchSize|Occupancy" | tr '\n' ' '; echo " <- $name"
