#!/bin/bash
# Experimental build that recompiles TWO units (host side of the forward + one method unit): scripts/exp/build_variant2.sh <name> <flags...>
set -e
name=$1; shift
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/multiviewhmr_amd/csrc; LIB=$ROOT/multiviewhmr_amd/lib; OUT=$ROOT/multiviewhmr_amd/lib_exp/$name
mkdir -p $OUT/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -I$ROOT/include -I$CS -Wall -Wno-unused-function"
# the product's own flags (csrc/Makefile), -save-temps=obj included: hipcc's code generation differs without it (r05: the same header spilled in
# k_fwd_ws's hot loops and ran at half speed when built without), and the same loop gate
for u in unproject_brick_fwd unproject_brick_fwd_m0; do mkdir -p $OUT/obj/tmp_$u; /opt/rocm/bin/hipcc $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -save-temps=obj -c $CS/$u.hip -o $OUT/obj/tmp_$u/unit.o 2> $OUT/obj/$u.log & done; wait
for u in unproject_brick_fwd unproject_brick_fwd_m0; do mv $OUT/obj/tmp_$u/unit.o $OUT/obj/$u.o; done
python3 $CS/check_loops.py 11k_fwd_brickI:ds_read_b128 k_fwd_brick_groups:ds_read_b128 k_fwd_ws:ds_read_b128 -- $OUT/obj/tmp_unproject_brick_fwd_m0/*gfx950*.s || echo "LOOP GATE FAILED for $name"
rm -rf $OUT/obj/tmp_unproject_brick_fwd $OUT/obj/tmp_unproject_brick_fwd_m0
OBJS=""
for o in $LIB/*.o; do b=$(basename $o); if [ "$b" != "unproject_brick_fwd.o" ] && [ "$b" != "unproject_brick_fwd_m0.o" ]; then OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmvhmr_unproject.so $OUT/obj/unproject_brick_fwd.o $OUT/obj/unproject_brick_fwd_m0.o $OBJS
echo "built $name"
