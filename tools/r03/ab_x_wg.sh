#!/bin/bash
# workgroup shape of the single-strip X sweep: 4 strips of one row (default) / one strip of 4 rows / the same with the XCD-aware order
L=armon.jl_amd/libarmon_hip.so
for extra in "" "--exact" "--track-x"; do
echo "== $extra"; python tools/ab_sweep.py --rounds 15 --copy $extra --env 'rows:ARMON_X_ROWS=1;xcd:ARMON_X_XCD=1' alongx=$L rows=$L xcd=$L | grep "sweep_X"
done
echo "== 4096x8192"; python tools/ab_sweep.py --rounds 30 --shape 4096x8192 --copy --env 'rows:ARMON_X_ROWS=1;xcd:ARMON_X_XCD=1' alongx=$L rows=$L xcd=$L | grep "sweep_X"
echo "== f32"; python tools/ab_sweep.py --rounds 15 --f32 --copy --env 'rows:ARMON_X_ROWS=1;xcd:ARMON_X_XCD=1' alongx=$L rows=$L xcd=$L | grep "sweep_X"
