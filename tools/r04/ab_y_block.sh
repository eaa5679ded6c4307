#!/bin/bash
# Y march: columns per workgroup (ARMON_Y_BLOCK = 128 / 256 / 512), each variant first once; 16384² and the strong-scaling tiles
V=variants
for shape in 16384x16384 8192x8192 4096x8192; do
  echo "== $shape"
  python3 tools/ab_sweep.py --rounds 15 --shape $shape b256=$V/r4final/libarmon_hip.so b512=$V/r4yb512/libarmon_hip.so b128=$V/r4yb128/libarmon_hip.so | grep sweep_Y
  python3 tools/ab_sweep.py --rounds 15 --shape $shape b512=$V/r4yb512/libarmon_hip.so b128=$V/r4yb128/libarmon_hip.so b256=$V/r4final/libarmon_hip.so | grep sweep_Y
done
