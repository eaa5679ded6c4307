#!/bin/bash
# one mapping per process (no remapping of handles: stale translations survive hipMemUnmap), at VA = hint + shift, vectors `pad` apart
CH=${CHUNK:-64}
for rep in 1 2 3; do
for sp in "$@"; do
  echo "== rep $rep shift:pad $sp MiB"
  timeout -k 10 120 tools/probes/probe_vmm --chunk=$CH --perms=0 --hint=600000000000 --one=$sp --verify 2>&1 | grep "verify\|va shift"
done; done
