#!/bin/bash
# sample clocks and power every 0.5 s while the bench runs
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|junction|memory\)" | tr '\n' ' ' ; echo; sleep 0.4; done) > gpurun_out/smi_trace.txt &
MON=$!
python bench.py --no-cpu-baseline --steps 1200 --warmup 5 > gpurun_out/bench_long.json
wait $MON
