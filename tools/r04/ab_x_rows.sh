#!/bin/bash
# X sweep workgroup shape / XCD mapping after the prologue repairs (fp64): time at three shapes, then traffic by counters.
L=armon.jl_amd/libarmon_hip.so
for shape in 16384x16384 8192x8192 4096x8192; do
  echo "== $shape"; python3 tools/ab_sweep.py --rounds 15 --shape $shape --env "rows2:ARMON_X_ROWS=2;xcd:ARMON_X_XCD=1" rows1=$L rows2=$L xcd=$L | grep sweep_X
  echo "-- rows2 first"; python3 tools/ab_sweep.py --rounds 15 --shape $shape --env "rows2:ARMON_X_ROWS=2" rows2=$L rows1=$L | grep sweep_X
done
for rows in 1 2; do
  echo "== traffic with ARMON_X_ROWS=$rows"
  ARMON_X_ROWS=$rows python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['per_kernel_ms'], 'traffic', r['traffic'], r['traffic']/(64*16384*16384) if r['traffic'] else None, r['traffic_source'][:40])"
done
