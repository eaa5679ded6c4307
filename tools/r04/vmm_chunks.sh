#!/bin/bash
# physical chunk size behind the 8 vectors (one valid mapping per process, verified): does the X+Y time follow the size of the
# physically contiguous pieces (translation reach)? chunk:pad pairs in MiB; pad chosen so that every vector starts on a 4-GiB VA boundary
for rep in 1 2 3; do
for cp in 2:2044 16:2032 64:1984 256:2048 1024:2048 2052:2044; do
  c=${cp%%:*}; p=${cp##*:}
  echo "== rep $rep chunk $c MiB pad $p MiB"
  timeout -k 10 200 tools/probes/probe_vmm --chunk=$c --perms=0 --hint=600000000000 --one=0:$p --verify 2>&1 | grep "^#\|verify\|va shift"
done; done
