#!/bin/bash
# usage: [VARIANT_SRC=fused_sweep_f32] tools/build_variant.sh <name> [extra hipcc flags...]
# Builds variants/<name>/libarmon_hip.so: fused_sweep_f64.hip (or $VARIANT_SRC.hip) recompiled with the extra flags, every
# other object taken from the regular in-tree build. Run with ARMON_HIP_LIB=variants/<name>/libarmon_hip.so.
set -e
name=$1; shift
src=${VARIANT_SRC:-fused_sweep_f64}
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/armon.jl_amd
mkdir -p $root/variants/$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
  -DARMON_BUILDING_LIB "$@" -c $pkg/csrc/$src.hip -o $root/variants/$name/$src.o
objs=$(ls $pkg/build/*.o | grep -v "$src.o\|_alt.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/variants/$name/libarmon_hip.so \
  $root/variants/$name/$src.o $objs -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo built variants/$name/libarmon_hip.so
