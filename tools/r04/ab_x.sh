#!/bin/bash
# usage: tools/r04/ab_x.sh <tag> (variants r4base[_biz] against <tag>[_biz]): bits first, then interleaved timing at 16384²
tag=$1
V=variants
set -e
for t in Sod_circ:"" Bizarrium:_biz; do
  test=${t%%:*}; sfx=${t##*:}
  ARMON_HIP_LIB=$V/r4base$sfx/libarmon_hip.so python3 tools/r04/run_case.py /tmp/a$sfx.npz --test $test --n 384 --ny 200 --maxcycle 25
  ARMON_HIP_LIB=$V/$tag$sfx/libarmon_hip.so python3 tools/r04/run_case.py /tmp/b$sfx.npz --test $test --n 384 --ny 200 --maxcycle 25
  python3 tools/r04/run_case.py --compare /tmp/a$sfx.npz /tmp/b$sfx.npz || true
done
echo "== Sod 16384², tuned"; python3 tools/ab_sweep.py --rounds 15 --copy base=$V/r4base/libarmon_hip.so new=$V/$tag/libarmon_hip.so
echo "== Bizarrium 16384², tuned"; python3 tools/ab_sweep.py --rounds 15 --test Bizarrium base=$V/r4base_biz/libarmon_hip.so new=$V/${tag}_biz/libarmon_hip.so
