#!/usr/bin/env python3
"""Host time to ENQUEUE one solver cycle — no waits — against the GPU time of that cycle.

    python tools/r05/enqueue_time.py [--tile 4096x8192] [--cycles 200]

(a) one tile with two remote sides on BOTH axes over the library's exchange: an in-process 1 x 1 group made periodic
    (its own neighbour on every side: pack, copy, interior, unpack, two strips, join per sweep — the work of a middle
    rank of a 4 x 2 grid and more: a real rank has one remote side along y);
(b) the in-process group driving 8 tiles (4 x 2, every tile on this one device) of that size.
Each through three drivers: the host mirror calling the library step by step (round 4's path), armon_hip_mgpu_cycle from
the calling thread, armon_hip_mgpu_cycle with one host thread per tile. A constant time step (cst_dt) removes the one
host wait of a cycle (the dt read-back), so the loop below only enqueues; the GPU time is the wall time until the device
has drained, per cycle. Enqueue time above the GPU time = a host-bound run.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--tile", default="4096x8192")
ap.add_argument("--cycles", type=int, default=200)
ap.add_argument("--pack-ab", action="store_true", help="every case twice: halo packs on the compute stream (round 4) / on the transfer stream")
ap.add_argument("--prio-ab", action="store_true", help="every case twice: transfer stream at normal / lowest priority")
args = ap.parse_args()
tx, ty = (int(v) for v in args.tile.split("x"))

for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29577")):
    os.environ.setdefault(k, v)
import torch                          # noqa: E402
import torch.distributed as dist      # noqa: E402
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from armon_amd.multi_tile import TileGroup     # noqa: E402


def measure(P, N, periodic, driver):
    group = TileGroup(P, test="Sod", N=N, silent=5, maxcycle=10 ** 9, maxtime=1e9, cst_dt=True, Dt=1e-7, periodic=periodic,
                      native_cycle=driver != "host calls", placement_tries=0)
    try:
        group.set_threads(driver == "native, thread per tile")
        group.init_test()
        gdt = group.global_dt
        gdt.reset()
        for _ in range(5):
            group.solver_cycle(last_cycle=False)
            gdt.next_cycle()
        group.wait()
        t0 = time.perf_counter()
        for _ in range(args.cycles):
            group.solver_cycle(last_cycle=False)
            gdt.next_cycle()
        t1 = time.perf_counter()
        group.wait()
        t2 = time.perf_counter()
        group.drain()
        group.wait()
        return (t1 - t0) / args.cycles * 1e3, (t2 - t0) / args.cycles * 1e3
    finally:
        group.close()


print(f"# tile {tx}x{ty} fp64 tuned, GAD+minmod+euler_2nd, Sequential X,Y; {args.cycles} cycles enqueued back to back (cst_dt: no host wait)")
print(f"# {'case':58s} {'driver':26s} enqueue ms/cycle   until drained ms/cycle")
for label, P, N, periodic in (("(a) 1 tile, remote on all 4 sides (periodic 1x1 group)", (1, 1), (tx, ty), (True, True)),
                              ("(b) 8 tiles (4x2) in one process, all on this device", (4, 2), (4 * tx, 2 * ty), (False, False))):
    for driver in ("host calls", "native, calling thread", "native, thread per tile"):
        if P == (1, 1) and driver == "native, thread per tile":
            continue
        for pack in (("compute", "xfer") * 2 if args.pack_ab else ("xfer",)):
            for prio in (("normal", "lowest") * 2 if args.prio_ab else (None,)):
                os.environ["ARMON_MGPU_PACK"] = pack
                os.environ.pop("ARMON_MGPU_XFER_PRIORITY", None)        # None: the library's own choice (lowest when a tile has its device to itself)
                if prio:
                    os.environ["ARMON_MGPU_XFER_PRIORITY"] = prio
                enq, total = measure(P, N, periodic, driver)
                print(f"  {label:58s} {driver:26s} {enq:10.4f}        {total:10.4f}" + (f"   packs on {pack}" if args.pack_ab else "")
                      + (f"   transfer stream priority {prio}" if args.prio_ab else ""), flush=True)


def measure_rank(N, native):
    """(c) the one-process-per-GPU path: ONE rank over RCCL on a periodic 1 x 1 process grid — ncclSend / ncclRecv to itself on
    all four sides, grouped per sweep on the transfer stream (what a rank of a multi-GPU run enqueues, and more)."""
    import armon_amd
    from armon_amd.halo_exchange import setup
    from armon_amd.solver import BlockGrid, drain_halo, init_test, solver_cycle
    params = armon_amd.ArmonParameters(test="Sod", N=N, use_MPI=True, P=(1, 1), periodic=(True, True), device_id=0, silent=5,
                                       maxcycle=10 ** 9, maxtime=1e9, cst_dt=True, Dt=1e-7, native_cycle=native, placement_tries=0)
    grid = BlockGrid(params)
    comm = setup(params, grid)
    assert getattr(comm, "native", False), "the library's RCCL exchange was not selected"
    init_test(params, grid)
    gdt = grid.global_dt
    gdt.reset()
    for _ in range(5):
        solver_cycle(params, grid, last_cycle=False)
        gdt.next_cycle()
    params.wait()
    t0 = time.perf_counter()
    for _ in range(args.cycles):
        solver_cycle(params, grid, last_cycle=False)
        gdt.next_cycle()
    t1 = time.perf_counter()
    params.wait()
    t2 = time.perf_counter()
    drain_halo(grid)
    params.wait()
    comm.close()
    return (t1 - t0) / args.cycles * 1e3, (t2 - t0) / args.cycles * 1e3


label = "(c) 1 rank over RCCL, send/recv to itself on all 4 sides"
for native in (False, True):
    for pack in (("compute", "xfer") * 2 if args.pack_ab else ("xfer",)):
        for prio in (("normal", "lowest") * 2 if args.prio_ab else (None,)):
            os.environ["ARMON_MGPU_PACK"] = pack
            os.environ.pop("ARMON_MGPU_XFER_PRIORITY", None)
            if prio:
                os.environ["ARMON_MGPU_XFER_PRIORITY"] = prio
            enq, total = measure_rank((tx, ty), native)
            print(f"  {label:58s} {'native, calling thread' if native else 'host calls':26s} {enq:10.4f}        {total:10.4f}"
                  + (f"   packs on {pack}" if args.pack_ab else "") + (f"   transfer stream priority {prio}" if args.prio_ab else ""), flush=True)
dist.destroy_process_group()
