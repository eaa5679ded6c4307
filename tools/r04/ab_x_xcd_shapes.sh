#!/bin/bash
# ARMON_X_XCD at the smaller shapes (config 2's grid, the strong-scaling tiles), each variant first once; exact flavour at 16384²
L=armon.jl_amd/libarmon_hip.so
E="xcd:ARMON_X_XCD=1"
for shape in 8192x8192 8192x16384 4096x8192 2048x2048; do
  echo "== $shape"
  python3 tools/ab_sweep.py --rounds 30 --shape $shape --env "$E" plain=$L xcd=$L | grep sweep_X
  python3 tools/ab_sweep.py --rounds 30 --shape $shape --env "$E" xcd=$L plain=$L | grep sweep_X
done
echo "== exact 16384x16384"
python3 tools/ab_sweep.py --rounds 15 --exact --env "$E" plain=$L xcd=$L | grep sweep_X
python3 tools/ab_sweep.py --rounds 15 --exact --env "$E" xcd=$L plain=$L | grep sweep_X
echo "== tuned + dt tracking on X, 16384x16384"
python3 tools/ab_sweep.py --rounds 15 --track-x --env "$E" plain=$L xcd=$L | grep sweep_X
python3 tools/ab_sweep.py --rounds 15 --track-x --env "$E" xcd=$L plain=$L | grep sweep_X
echo "== config 2 (Godunov) 8192x8192"
python3 tools/ab_sweep.py --rounds 30 --shape 8192x8192 --scheme Godunov --env "$E" plain=$L xcd=$L | grep sweep_X
python3 tools/ab_sweep.py --rounds 30 --shape 8192x8192 --scheme Godunov --env "$E" xcd=$L plain=$L | grep sweep_X
