#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on ONE GPU (every rank on cuda:0 over gloo, host staging: code path only, not a
# measurement). The pool allows at most 6 processes on the card, so the 8-rank launch the driver makes is rehearsed at 6
# ranks (3x2: middle tiles with two remote sides along x, remote sides on both axes) and at 4 (2x2). Both workloads of the
# N > 1 line (weak, then strong) run; the lines are kept under profiles/.
set -e
mkdir -p gpurun_out
export ARMON_BENCH_REHEARSAL=1
for spec in "4 2x2 29541" "6 3x2 29542"; do
  set -- $spec
  timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $3 \
    bench.py --gpus $1 --grid $2 --cells 2048 --steps 5 --warmup 2 > gpurun_out/r04_bench_rehearsal_$1rank.json 2> gpurun_out/r04_bench_rehearsal_$1rank.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r04_bench_rehearsal_$1rank.json'))
print('$1 ranks', d['scaling'], d['value'], d['config']['workload'])
print('   strong', d.get('strong'))
"
done
