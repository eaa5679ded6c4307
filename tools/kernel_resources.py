#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/scratch/occupancy per kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-DARMON_BUILDING_LIB", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *extra]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+([A-Za-z ]+(?: \[[^\]]+\])?): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4}  kernel")
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"armon::fused::", "", name)
    name = re.sub(r"\(.*$", "", name)
    print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('TotalSGPRs','?'):>5} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8} {r.get('LDS Size [bytes/block]','?'):>7} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>4}  {name}")
