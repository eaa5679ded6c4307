#!/bin/bash
# the staged bench line under rocprofv3 (kernel trace + stats), re-collected after the aligned-store x forms
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_r03
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
tag=r03_staged_sod16384
rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 $root/bench.py --staged --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}.err
cp "$(find $out/$tag -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
python3 $root/tools/trace_timed_region.py "$(find $out/$tag -name '*kernel_trace.csv' | head -1)" $out/${tag}_timed_region.json 20 k_euler_projection k_acoustic_GAD k_advection_second k_dtCFL k_cell_update k_perfect_gas > /dev/null
rm -rf $out/$tag
