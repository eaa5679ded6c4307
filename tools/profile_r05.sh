#!/bin/bash
# Round-5 profiles (run on the GPU box from the repo root): bench lines, rocprofv3 kernel statistics of the same
# commands, timed-region statistics, PMC traffic and issue counters. Summaries land in gpurun_out/prof_r05/ and are
# copied into profiles/ (profiles/README.md says which command made which file; tools/evidence_table.py reads them).
# bench.py under rocprofv3 always gets --no-measure-traffic: its own child --pmc passes must not start inside a profiler.
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_r05
mkdir -p $out
python3 $root/bench.py > $out/r05_fused_fast_sod16384_bench.json 2> $out/bench_plain.err
echo "plain bench done"
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 $root/bench.py --no-measure-traffic --no-cpu-baseline "$@" > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}.err
  cp "$(find $out/$tag -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
  python3 $root/tools/trace_timed_region.py "$(find $out/$tag -name '*kernel_trace.csv' | head -1)" $out/${tag}_timed_region.json 20 k_sweep k_euler_projection k_acoustic_GAD k_advection_second k_dtCFL k_cell_update k_perfect_gas > /dev/null
  rm -rf $out/$tag
  echo "$tag done"
}
run r05_fused_fast_sod16384
run r05_fused_exact_sod16384 --exact
run r05_fused_fast_sod16384_f32 --f32
run r05_staged_sod16384 --staged
run r05_fused_fast_sedov16384 --config 3
run r05_fused_fast_bizarrium16384 --test Bizarrium
run r05_fused_fast_sod8192_godunov --config 2
# hardware counters: separate passes, kernel trace only
cd $root
tools/pmc.sh r05 FETCH_SIZE WRITE_SIZE -- $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-measure-traffic
python3 tools/pmc_to_traffic.py gpurun_out/pmc_r05 $out/r05_pmc_traffic_fused_fast_sod16384.json "Sod 16384x16384 fp64 GAD+minmod+euler_2nd, fused tuned sweeps, bench.py --steps 5 --warmup 1 (includes the placement-search launches)" > /dev/null
mkdir -p $out/r05_pmc
for f in $(find gpurun_out/pmc_r05 -name '*counter_collection.csv'); do g=$(basename $(dirname $(dirname $(dirname $f)))); grep -E "Kernel_Name|k_sweep" $f > $out/r05_pmc/${g,,}_counter_collection.csv; done
SQ="SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM,SQ_INSTS_VMEM_RD"
tools/pmc.sh r05sq $SQ -- $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic
python3 tools/pmc_summary.py gpurun_out/pmc_r05sq k_sweep > $out/r05_pmc_sq_fused_fast_sod16384.txt
tools/pmc.sh r05sqbiz $SQ -- $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic --test Bizarrium
python3 tools/pmc_summary.py gpurun_out/pmc_r05sqbiz k_sweep > $out/r05_pmc_sq_fused_fast_bizarrium16384.txt
rm -rf gpurun_out/pmc_r05 gpurun_out/pmc_r05sq gpurun_out/pmc_r05sqbiz
ls -la $out
