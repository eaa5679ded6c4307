#!/bin/bash
# Y march: the nearly empty last workgroup of a row folded into the first one's idle lanes (tools/r04/patches/y_wrap_experiment.patch, NOT in the tree: measured, no gain) and the run
# length that then fills the device in whole rounds; 16384², config 2's 8192², the strong-scaling tiles. Bits first.
V=armon.jl_amd/libarmon_hip.so
ARMON_HIP_LIB=$V ARMON_Y_WRAP=0 python3 tools/r04/run_case.py /tmp/w0.npz --test Sod_circ --n 512 --ny 300 --maxcycle 12
ARMON_HIP_LIB=$V python3 tools/r04/run_case.py /tmp/w1.npz --test Sod_circ --n 512 --ny 300 --maxcycle 12
python3 tools/r04/run_case.py --compare /tmp/w0.npz /tmp/w1.npz
E="nowrap:ARMON_Y_WRAP=0;wrap:ARMON_Y_WRAP=1;w512:ARMON_Y_WRAP=1,ARMON_Y_SEG=512;w683:ARMON_Y_WRAP=1,ARMON_Y_SEG=683;w1024:ARMON_Y_WRAP=1,ARMON_Y_SEG=1024"
echo "== 16384x16384"
python3 tools/ab_sweep.py --rounds 15 --env "$E" nowrap=$V wrap=$V w512=$V w683=$V w1024=$V | grep sweep_Y
python3 tools/ab_sweep.py --rounds 15 --env "$E" w1024=$V w683=$V w512=$V wrap=$V nowrap=$V | grep sweep_Y
E2="nowrap:ARMON_Y_WRAP=0;wrap:ARMON_Y_WRAP=1"
for shape in 8192x8192 8192x16384 4096x8192; do
  echo "== $shape"
  python3 tools/ab_sweep.py --rounds 25 --shape $shape --env "$E2" nowrap=$V wrap=$V | grep sweep_Y
  python3 tools/ab_sweep.py --rounds 25 --shape $shape --env "$E2" wrap=$V nowrap=$V | grep sweep_Y
done
