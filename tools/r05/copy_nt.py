#!/usr/bin/env python3
"""The measurement aid armon_hip_stream_copy4 (the same-device ceiling bench.py reports) in its four forms — ARMON_COPY_NT bit 0: non-temporal
loads, bit 1: non-temporal stores — interleaved launch by launch, at 16384² and on the 4096 x 8192 tile.   python tools/r05/copy_nt.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import armon_amd, ctypes as C
from armon_amd.solver import BlockGrid, STATE_VARS
for shape in ((16384, 16384), (4096, 8192)):
    p = armon_amd.ArmonParameters(test="Sod", N=shape, silent=5, placement_tries=0)
    g = BlockGrid(p)
    dev = p.device
    src, dst = [g.data[f] for f in STATE_VARS], [g.alt[f] for f in STATE_VARS]
    nb = src[0].nbytes & ~15
    res = {}
    for rnd in range(12):
        for nt in (0, 1, 2, 3):
            armon_amd.lib().armon_hip_set_tuning(dev.ctx, b"ARMON_COPY_NT", nt)
            dev.event_record(10); dev.stream_copy4(src, dst, nb); dev.event_record(11)
            if rnd >= 2: res.setdefault(nt, []).append(dev.event_elapsed_ms(10, 11))
    for nt, v in res.items():
        v.sort(); m = v[len(v) // 2]
        print(f"{shape[0]}x{shape[1]} copy4 nt={nt}: median {m:.4f} ms  {8 * nb / m / 1e6:.1f} GB/s")
    del g
