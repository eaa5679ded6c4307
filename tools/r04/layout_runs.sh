#!/bin/bash
# Round 4, review task 1: the layout probe in N fresh processes (tools/probes/probe_layout.hip), output under gpurun_out/.
# usage: tools/r04/layout_runs.sh [N=10] [probe args...]
N=${1:-10}; shift
mkdir -p gpurun_out
out=gpurun_out/r04_layout_probe.txt
: > $out
for i in $(seq 1 $N); do
  echo "## process $i" >> $out
  timeout -k 10 300 tools/probes/probe_layout "$@" >> $out 2>&1 || { echo "probe failed in process $i" >> $out; exit 1; }
done
tail -n 60 $out
