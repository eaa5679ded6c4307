#!/bin/bash
# Round-5 evidence for the Y march on the strong-scaling tiles (run on the GPU box from the repo root):
#   tools/r05/small_tiles.sh [tag-suffix]
# For the tiles one GPU owns when Sod 16384² is split over 8 and 4 GPUs (4096x8192, 8192x8192): the bench line with
# measured traffic, the same command under rocprofv3 --kernel-trace --stats (kernel statistics + the timed region's last
# 100 launches), the four SQ counter groups and an occupancy group. Summaries land in gpurun_out/prof_r05/.
# bench.py under rocprofv3 always gets --no-measure-traffic (no profiler inside a profiled child).
root=${GRAFT_REPO_ROOT:-$PWD}
sfx=$1
out=$root/gpurun_out/prof_r05
mkdir -p $out
SQ="SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM,SQ_INSTS_VMEM_RD SQ_WAVES,SQ_WAIT_ANY,GRBM_GUI_ACTIVE"
for shape in 4096x8192 8192x8192; do
  tag=r05_tile_${shape}${sfx}
  python3 $root/bench.py --global $shape --grid 1x1 --steps 100 --warmup 5 --no-cpu-baseline > $out/${tag}_bench.json 2> $out/${tag}_bench.err
  echo "$tag plain done"
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 $root/bench.py --global $shape --grid 1x1 --steps 100 --warmup 5 --no-measure-traffic --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}.err )
  cp "$(find $out/$tag -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
  python3 $root/tools/trace_timed_region.py "$(find $out/$tag -name '*kernel_trace.csv' | head -1)" $out/${tag}_timed_region.json 100 k_sweep > /dev/null
  rm -rf $out/$tag
  echo "$tag stats done"
  cd $root
  tools/pmc.sh $tag FETCH_SIZE WRITE_SIZE $SQ -- $root/bench.py --global $shape --grid 1x1 --steps 20 --warmup 2 --no-cpu-baseline --no-measure-traffic
  python3 tools/pmc_summary.py gpurun_out/pmc_$tag k_sweep > $out/${tag}_pmc.txt
  rm -rf gpurun_out/pmc_$tag
  echo "$tag pmc done"
done
ls -la $out
