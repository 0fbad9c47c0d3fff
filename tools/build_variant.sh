#!/bin/bash
# Timing-only variant of the library: render_fused.hip recompiled with extra -D flags, linked with the regular objects.
#   bash tools/build_variant.sh <name> <hipcc flags...>   ->  tools/scratch/variants/libcropnerf_<name>.so
# Select it with CROPNERF_HIP_LIB=<path> (cropnerf_amd/_lib.py).  tools/scratch/ is git-ignored and travels with gpurun.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/cropnerf-a-neural-radiance-field-based-framework_amd
OUT=$ROOT/tools/scratch/variants
mkdir -p $OUT
SRC=${VARIANT_SRC:-render_fused}
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -x hip -Wno-unused-result "$@" -c $PKG/csrc/$SRC.hip -o $OUT/${SRC}_$NAME.o
OBJS=$(ls $PKG/build/*.o | grep -v "/$SRC.o" | grep -v "\.det\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libcropnerf_$NAME.so $OBJS $OUT/${SRC}_$NAME.o
echo $OUT/libcropnerf_$NAME.so
