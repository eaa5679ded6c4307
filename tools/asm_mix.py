#!/usr/bin/env python3
"""Instruction mix of selected kernels in a hipcc -S device assembly file.
usage: asm_mix.py file.s substring [substring...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\s*s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    ins = []
    for l in body.splitlines():
        t = l.strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        ins.append(t.split()[0])
    c = collections.Counter(ins)
    groups = collections.Counter()
    for k, v in c.items():
        if k.startswith("v_") and "f64" in k: groups["valu_f64"] += v
        elif k.startswith("v_"): groups["valu_other"] += v
        elif k.startswith("s_"): groups["salu"] += v
        elif k.startswith("ds_"): groups["lds"] += v
        elif k.startswith(("global_", "buffer_", "flat_", "scratch_")): groups["vmem"] += v
        else: groups["other"] += v
    print(name[:90], "total", len(ins), dict(groups))
    print("   ", ", ".join(f"{k}:{v}" for k, v in c.most_common(40)))
