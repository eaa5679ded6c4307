#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]
# Builds variants/<name>/libarmon_hip.so: fused_sweep_f64.hip recompiled with the extra flags, every other
# object taken from the regular in-tree build. Run with ARMON_HIP_LIB=variants/<name>/libarmon_hip.so.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/armon.jl_amd
mkdir -p $root/variants/$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
  -DARMON_BUILDING_LIB "$@" -c $pkg/csrc/fused_sweep_f64.hip -o $root/variants/$name/fused_sweep_f64.o
objs=$(ls $pkg/build/*.o | grep -v 'fused_sweep_f64.o\|_alt.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/variants/$name/libarmon_hip.so \
  $root/variants/$name/fused_sweep_f64.o $objs -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo built variants/$name/libarmon_hip.so
