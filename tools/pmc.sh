#!/bin/bash
# usage: tools/pmc.sh <tag> <counter-list...> -- <python script args...>
# Runs one rocprofv3 --pmc pass per counter group (comma-separated counters in one group share a pass)
# with kernel-trace only, output under gpurun_out/pmc_<tag>/<group>/.
set -e
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for g in "${groups[@]}"; do
  out=$root/gpurun_out/pmc_${tag}/$(echo $g | tr ',' '_')
  mkdir -p $out
  rocprofv3 --kernel-trace --pmc $(echo $g | tr ',' ' ') --output-format csv -d $out -- python3 "$@" > $out/run.log 2>&1
  echo "pass $g done"
done
