#!/bin/bash
# XCD-aware workgroup mapping of the X sweep (ARMON_X_XCD) against the plain one and the 4-strips-per-workgroup shape,
# every variant first once (the first build of a round follows another kernel and pays for it), then traffic by counters.
L=armon.jl_amd/libarmon_hip.so
E="rows2:ARMON_X_ROWS=2;xcd:ARMON_X_XCD=1"
for t in Sod Bizarrium; do
  echo "== $t 16384x16384"
  python3 tools/ab_sweep.py --rounds 20 --test $t --env "$E" rows1=$L rows2=$L xcd=$L | grep sweep_X
  python3 tools/ab_sweep.py --rounds 20 --test $t --env "$E" xcd=$L rows1=$L rows2=$L | grep sweep_X
  python3 tools/ab_sweep.py --rounds 20 --test $t --env "$E" rows2=$L xcd=$L rows1=$L | grep sweep_X
done
echo "== traffic with ARMON_X_XCD=1"
ARMON_X_XCD=1 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['per_kernel_ms'], 'traffic', r['traffic'], r['traffic']/(64*16384*16384) if r['traffic'] else None, r['traffic_source'][:40])"
for i in 1 2 3; do for x in 0 1; do ARMON_X_XCD=$x python3 bench.py --no-cpu-baseline --no-measure-traffic | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('ARMON_X_XCD=$x', d['value'], r['per_kernel_ms'], d['config']['hbm_placement']['chosen_ms'])"; done; done
