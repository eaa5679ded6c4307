#!/bin/bash
# How often does a fresh process find a fast placement, as a function of the number of spare vectors?
# (each line: spares, the X+Y times of the draws, the chosen one)   usage: tools/placement_runs.sh "4 8 16 24" 3
for sp in $1; do for rep in $(seq 1 $2); do
python3 - $sp <<'PY'
import sys; sys.path.insert(0, '.')
import armon_amd
from armon_amd.solver import BlockGrid
sp = int(sys.argv[1])
p = armon_amd.ArmonParameters(test="Sod", N=(16384, 16384), silent=5, placement_tries=24)
g = BlockGrid(p)
r = g.tune_placement(spare=sp, keep_state=False)
print(f"spares {sp:2d}: chosen {r['chosen_ms']:.3f} ms after {r['tries']:2d} draws  {r['x_plus_y_ms']}", flush=True)
PY
done; done
