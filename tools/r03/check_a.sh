#!/bin/bash
# forced one-rank RCCL bench through the native exchange, the dt / NaN tests, the staged bench and its kernel statistics
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
ARMON_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cells 8192 --no-cpu-baseline --require-native 2>gpurun_out/r03_bench_forcedist.err | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('force-dist', j['value'], j['config']['halo_exchange'], j['config']['halo_exchange_downgraded'], j['config']['halo_exchange_error'])"; echo rc=$?; tail -2 gpurun_out/r03_bench_forcedist.err
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py -x -q -m gpu -k 'dtCFL or nan or invalid or golden' 2>&1 | tail -3
python bench.py --staged --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('staged', j['value'], j['ms_per_step'], j['roofline']['per_kernel_ms'])"
cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $root/bench.py --staged --no-cpu-baseline --steps 5 > /dev/null 2>&1
cp $(find /tmp/st -name '*kernel_stats.csv' | head -1) $root/gpurun_out/r03_staged_kernel_stats.csv
head -25 $root/gpurun_out/r03_staged_kernel_stats.csv | cut -c1-150
