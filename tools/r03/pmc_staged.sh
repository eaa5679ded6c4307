#!/bin/bash
# SQ issue counters of the staged kernels (separate --pmc passes, kernel trace only)
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
tools/pmc.sh r03staged SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY -- $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --staged
python3 tools/pmc_summary.py gpurun_out/pmc_r03staged k_ > gpurun_out/r03_pmc_sq_staged_sod16384.txt
rm -rf gpurun_out/pmc_r03staged
