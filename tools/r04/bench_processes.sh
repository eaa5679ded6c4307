#!/bin/bash
# The headline line in N fresh processes on one box: value, frac, X / Y ms, the placement search's outcome, the same-device copy.
N=${1:-10}; shift
echo "# python bench.py --no-cpu-baseline --no-measure-traffic $@  x $N processes"
echo "# run  Gcells/s  frac    X ms    Y ms   copy GB/s  chosen ms  tries rounds"
for i in $(seq 1 $N); do
python3 bench.py --no-cpu-baseline --no-measure-traffic "$@" 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; k=r['per_kernel_ms']; p=j['config']['hbm_placement']
print('%3d   %7.1f  %.4f  %.4f  %.4f  %7.1f   %.3f    %3d   %d' % ($i, j['value']/1e3, r['frac'], k['sweep_x'], k['sweep_y'], r.get('stream_copy_GBps_this_device') or 0, p['chosen_ms'], p['tries'], p['rounds']))"
done
