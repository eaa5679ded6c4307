#!/bin/bash
# SQ issue counters of the fp32 and the exact-arithmetic fused sweeps (separate --pmc passes, kernel trace only)
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
for mode in f32 exact; do
  tools/pmc.sh r03$mode SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM,SQ_INSTS_VMEM_RD -- $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --$mode
  python3 tools/pmc_summary.py gpurun_out/pmc_r03$mode k_sweep > gpurun_out/r03_pmc_sq_fused_${mode}_sod16384.txt
  rm -rf gpurun_out/pmc_r03$mode
done
