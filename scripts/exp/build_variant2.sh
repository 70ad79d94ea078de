#!/bin/bash
# Experimental build that recompiles TWO units (host side of the forward + one method unit): scripts/exp/build_variant2.sh <name> <flags...>
set -e
name=$1; shift
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/multiviewhmr_amd/csrc; LIB=$ROOT/multiviewhmr_amd/lib; OUT=$ROOT/multiviewhmr_amd/lib_exp/$name
mkdir -p $OUT/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -I$ROOT/include -I$CS -Wall -Wno-unused-function"
for u in unproject_brick_fwd unproject_brick_fwd_m0; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/$u.hip -o $OUT/obj/$u.o 2> $OUT/obj/$u.log & done; wait
OBJS=""
for o in $LIB/*.o; do b=$(basename $o); if [ "$b" != "unproject_brick_fwd.o" ] && [ "$b" != "unproject_brick_fwd_m0.o" ]; then OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmvhmr_unproject.so $OUT/obj/unproject_brick_fwd.o $OUT/obj/unproject_brick_fwd_m0.o $OBJS
echo "built $name"
