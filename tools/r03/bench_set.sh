#!/bin/bash
# bench lines of the main configurations (one process each), summary per line
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$*'.ljust(44), j['value'], 'ms/step', j['ms_per_step'], r['per_kernel_ms'], 'frac', r['frac'], 'copy', r.get('stream_copy_GBps_this_device'), 'of copy', r.get('frac_of_stream_copy'), 'placed', (j['config']['hbm_placement'] or {}).get('chosen_ms'))"; }
run
run --exact
run --f32
run --config 2
run --config 3
run --test Bizarrium
run --global 8192x16384 --grid 1x1
run --global 8192x8192 --grid 1x1
run --global 4096x8192 --grid 1x1 --steps 100
