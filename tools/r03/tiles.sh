#!/bin/bash
# strong-scaling tiles alone (the tile one GPU owns when Sod 16384² is split over 1, 2, 4, 8 GPUs) and the tile structure
# on ONE GPU (tools/tile_overhead.py), re-collected for round 3
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
echo "# tile          Gcells/s   ms/cycle   frac    X ms    Y ms"
for g in 16384x16384 8192x16384 8192x8192 4096x8192; do
python bench.py --global $g --grid 1x1 --no-cpu-baseline --no-measure-traffic --steps 40 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; k=r['per_kernel_ms']; print('$g'.ljust(14), round(j['value']/1e3,1), j['ms_per_step'], r['frac'], k['sweep_x'], k['sweep_y'])"
done
python tools/tile_overhead.py
python tools/tile_overhead.py --global 32768x16384 --grids 1x1,2x2 --cycles 10
python tools/tile_overhead.py --global 32768x32768 --test Bizarrium --grids 1x1,4x2 --cycles 6
