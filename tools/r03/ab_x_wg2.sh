#!/bin/bash
# same A/B with the order reversed (the first build of a round follows the copy kernel and pays for it on small grids)
L=armon.jl_amd/libarmon_hip.so
echo "== 4096x8192, rows first"; python tools/ab_sweep.py --rounds 30 --shape 4096x8192 --copy --env 'rows:ARMON_X_ROWS=1;rows2:ARMON_X_ROWS=1' rows=$L alongx=$L rows2=$L alongx2=$L | grep "sweep_X"
echo "== f32, rows first"; python tools/ab_sweep.py --rounds 15 --f32 --copy --env 'rows:ARMON_X_ROWS=1;rows2:ARMON_X_ROWS=1' rows=$L alongx=$L rows2=$L alongx2=$L | grep "sweep_X"
echo "== 8192x8192, rows first"; python tools/ab_sweep.py --rounds 20 --shape 8192x8192 --copy --env 'rows:ARMON_X_ROWS=1;rows2:ARMON_X_ROWS=1' rows=$L alongx=$L rows2=$L alongx2=$L | grep "sweep_X"
