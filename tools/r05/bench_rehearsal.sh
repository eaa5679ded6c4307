#!/bin/bash
# Rehearsal of bench.py's N > 1 lines on ONE GPU (code path only, not a measurement: every rank / tile shares cuda:0).
#   tools/r05/bench_rehearsal.sh        (run on the GPU box from the repo root)
# (1) bare `python bench.py --gpus N`: the parent starts its own ranks (child torch.distributed.run, gloo between the
#     ranks here), N = 2, 4, 6 (the pool allows at most 6 processes on a card);
# (2) `--transport peer`: one process, 8 tiles through the library's in-process group, one host thread per tile.
export ARMON_BENCH_REHEARSAL=1
out=gpurun_out
for n in 2 4 6; do
  python3 bench.py --gpus $n --cells 2048 --steps 5 --warmup 2 > $out/r05_bench_${n}rank_rehearsal.json 2> $out/r05_bench_${n}rank_rehearsal.err
  echo "ranks $n: rc $?"
done
python3 bench.py --gpus 8 --cells 4096 --steps 5 --warmup 2 --transport peer > $out/r05_bench_8tile_peer_rehearsal.json 2> $out/r05_bench_8tile_peer_rehearsal.err
echo "peer 8: rc $?"
for f in $out/r05_bench_*rehearsal.json; do python3 -c "
import json,sys
d=json.load(open('$f'))
print('$f', d['n_gpus'], d['scaling'], d['value'], d['config']['transport'], d['config'].get('launched_by'), 'weak:', d.get('weak',{}).get('value'), 'slowest', d['config']['slowest_rank'])
"; done
