#!/usr/bin/env python3
"""Benchmark of the direction-split sweep hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 4 --global 32768x16384 --grid 2x2            BASELINE config 4 (= --config 4)
    python bench.py --gpus 8 --test Bizarrium --global 32768x32768 --grid 4x2      config 5 (= --config 5)
    python bench.py --gpus N --weak                        --cells² cells PER GPU instead (weak scaling only)

N > 1 needs no launcher: started from a bare shell, this process — before it touches a GPU — starts the N ranks as
children (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1, one rank per GPU over
RCCL), relays rank 0's line and the children's status, kills them on --launch-timeout, and falls back (loudly:
config.halo_exchange, config.launch_fallback) to --transport peer when the rank launch produced no line. Started BY a
launcher (WORLD_SIZE set: the driver's torch.distributed.run command) it is one of the ranks, as before.
--transport peer: ONE process drives the N devices through the library's in-process tile group
(armon_hip_mgpu_init with device_ids 0..N-1: hipMemcpyPeerAsync faces over xGMI, one host thread per device inside
armon_hip_mgpu_cycle) — no RCCL, no launcher, no torch.

Default workload = BASELINE.json's metric: the --cells² (16384²) Sod grid at EVERY N, split over the process grid
1→1x1, 2→2x1, 4→2x2, 8→4x2 by the reference's rule (ref src/parameters.jl:673-697): `value` is the STRONG curve
(`scaling: "strong"`). At N > 1 the weak workload (--cells² per GPU) is timed afterwards in the same processes and
reported next to it (`weak{value, ms_per_step}`), behind a watchdog so that it cannot cost the line.

A "step" is one solver cycle of the reference's time loop (ref src/solver.jl:288-320): the dt/CFL
reduction + one X sweep + one Y sweep over the whole grid. Inputs are the reference's own deterministic
initial conditions (init_test) already resident in HBM. Prints ONE JSON line on rank 0:

 value      = Mcells/s per sweep = cells(all ranks) · 2 sweeps · K / time / 1e6       
 roofline   = dominant kernel (the fused sweep, or euler_projection in --staged mode) timed live with
              HIP events on the kernel's stream; achieved = algorithmic B/cell · cells / mean duration per
              sweep; plus stream_copy_GBps_this_device / frac_of_stream_copy: the same bytes as a plain
              4-in/4-out copy on the same device and vectors, measured right after the timed region
 config.hbm_placement = the draws of BlockGrid.tune_placement (done at init_test, before any timing)
 cpu_baseline = the CPU oracle ("port": OpenMP restatement of the reference's 5-pass CPU path) timed on
              this host's cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic fp64 bytes per cell
B_PER_CELL = {"sweep_x": 64, "sweep_y": 64, "euler_projection": 112, "advection_second_order": 72,
              "advection_first_order": 72, "cell_update": 64, "acoustic_GAD": 48, "acoustic": 48,
              "perfect_gas_EOS": 56, "bizarrium_EOS": 56, "dtCFL": 24}


class EventTimer:
    """kernel_start/kernel_end callback recording HIP events around chosen kernels (no host sync)."""

    def __init__(self, device, names, max_pairs=500):
        self.device, self.names, self.max_pairs = device, set(names), max_pairs
        self.pairs = []          # (name, slot_a, slot_b)
        self.enabled = False

    def start(self, name):
        if self.enabled and name in self.names and len(self.pairs) < self.max_pairs:
            self.device.event_record(2 * len(self.pairs))

    def end(self, name):
        if self.enabled and name in self.names and len(self.pairs) < self.max_pairs:
            n = len(self.pairs)
            self.device.event_record(2 * n + 1)
            self.pairs.append((name, 2 * n, 2 * n + 1))

    def reserve(self, names):
        """Slots for a cycle the LIBRARY enqueues (armon_hip_mgpu_cycle records events 2s / 2s+1 of the first local tile
        around sweep s): returns the first slot, or -1."""
        if not self.enabled or len(self.pairs) + len(names) > self.max_pairs:
            return -1
        base = 2 * len(self.pairs)
        for k, name in enumerate(names):
            self.pairs.append((name, base + 2 * k, base + 2 * k + 1))
        return base

    def durations_ms(self):
        out = {}
        for name, a, b in self.pairs:
            out.setdefault(name, []).append(self.device.event_elapsed_ms(a, b))
        return out


# vectors smaller than this are not placed (armon_amd's default: 256 MiB); the environment variable lets a test place small ones
PLACEMENT_MIN_BYTES = int(os.environ.get("ARMON_BENCH_PLACEMENT_MIN_BYTES", 256 << 20))
PMC_TRAFFIC_FILE = "profiles/r03_pmc_traffic_fused_fast_sod16384.json"


def pmc_traffic(args, world, N_global):
    """HBM bytes per launch of the dominant kernels. Hardware counters cannot be read from inside this process:
    the figure is REPLAYED from the committed rocprofv3 --pmc passes of this same command (PMC_TRAFFIC_FILE, made by
    tools/pmc.sh + tools/pmc_to_traffic.py) — only for the exact workload they were collected on, otherwise null.
    Returns (bytes, source)."""
    if (world != 1 or args.staged or args.exact or args.f32 or tuple(N_global) != (16384, 16384) or args.test != "Sod"
            or args.scheme != "GAD"):
        return None, None
    for rel in (PMC_TRAFFIC_FILE, "profiles/r02_pmc_traffic_fused_fast_sod16384.json"):
        try:
            ks = json.load(open(os.path.join(ROOT, rel)))["kernels"]
            return round(sum(k["traffic_bytes_per_launch"] for k in ks.values()) / len(ks)), rel + " (replayed, not measured in this run)"
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None, None


def measure_traffic(args):
    """HBM bytes per launch of the dominant kernels, MEASURED: two child runs of this same command (3 steps) under
    ``rocprofv3 --kernel-trace --pmc`` — FETCH_SIZE and WRITE_SIZE in separate passes, nothing but the kernel trace
    next to them, the program itself right after ``--`` — before this process touches the GPU. Unit and gfx950
    correction as MI355X_MICROARCH.md prescribes (KiB; FETCH_SIZE counts half of a streaming read):
    traffic = (2·FETCH_SIZE + WRITE_SIZE)·1024. Returns (bytes, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    child = [os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-measure-traffic", "--cells", str(args.n),
             "--test", args.test, "--scheme", args.scheme]
    child += ["--staged"] if args.staged else []
    child += ["--exact"] if args.exact else []
    child += ["--f32"] if args.f32 else []
    child += ["--global", args.global_grid] if args.global_grid else []
    want = ("k_euler_projection",) if args.staged else ("k_sweep_x", "k_sweep_y")
    tmp = tempfile.mkdtemp(prefix="armon_pmc_", dir="/tmp")
    per_kernel = {}
    # a pass is 3 steps + the placement search of a fresh process: seconds at 2048², ≈ 40 s at 16384² (+ the image's first
    # import on a fresh box); the limit scales with the grid instead of a flat ten minutes
    limit = 120 + 240 * min(1.0, (args.n / 16384.) ** 2 if not args.global_grid else 1.0)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                                "python3", *child], cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=limit,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace")
            if r.returncode != 0:
                tail = " | ".join((r.stderr or "").strip().splitlines()[-3:])[-400:]
                return None, f"{counter} pass exited with status {r.returncode}: {tail}"
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = next((w for w in want if w in row["Kernel_Name"]), None)
                    if name and row["Counter_Name"] == counter:
                        per_kernel.setdefault(name, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
        traffic = []
        for name in want:
            c = per_kernel.get(name, {})
            if not c.get("FETCH_SIZE") or not c.get("WRITE_SIZE"):
                return None, f"no counter rows for {name}"
            traffic.append((2 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) + sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])) * 1024)
        return round(sum(traffic) / len(traffic)), ("measured for this line: child rocprofv3 --kernel-trace --pmc FETCH_SIZE / "
                                                    "WRITE_SIZE passes of the same command with --steps 3")
    except subprocess.TimeoutExpired:
        return None, f"a counter pass did not finish within {limit:.0f} s"
    except (subprocess.SubprocessError, OSError, KeyError, ValueError) as e:
        return None, f"{type(e).__name__}: {str(e)[:200]}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def under_profiler():
    """Is a rocprofiler tool attached to THIS process? (its own child --pmc passes must not start inside one: this pool
    refuses a profiler in a profiled child). Environment first, then what is actually mapped."""
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")):
        return True
    try:
        maps = open("/proc/self/maps").read()
    except OSError:
        return False
    return "rocprofiler-sdk-tool" in maps or "librocprofv3" in maps or "librocprofiler-sdk-tool" in maps


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(cores, 64)


def cpu_baseline(test, scheme, target_seconds=12.0):
    """Time the CPU oracle (kind "port") on the host cores: bounded sample of the same workload."""
    from oracle import oracle as O
    cores = usable_cores()
    n = 4096 if cores >= 8 else 2048
    fields = O.alloc_fields(n, n, 4)
    kw = dict(test=test, N=(n, n), scheme=scheme, threads=cores, native=True, fields=fields, maxtime=1e9)
    O.solve(maxcycle=1, **kw)                                   # warm-up: page-touch + init
    run, _ = O.solve(maxcycle=2, **kw)
    per_cycle = run.solve_seconds / 2
    cycles = int(max(2, min(400, target_seconds / max(per_cycle, 1e-6))))
    run, _ = O.solve(maxcycle=cycles, **kw)
    value = n * n * 2 * run.cycles / run.solve_seconds / 1e6
    return {"value": round(value, 2), "unit": "Mcells/s per sweep", "cores": cores, "kind": "port",
            "sample": f"{test} {n}x{n} fp64 {scheme}+minmod+euler_2nd, {run.cycles} cycles "
                      f"({run.solve_seconds:.1f} s), oracle/armon_oracle.c -O3 -march=native -ffp-contract=off OpenMP (oracle/Makefile)"}


def annotate_placements(placement, sizes):
    """Every rank's chosen draw against the group's best (ms per Mcell: tiles may differ in size)."""
    rates = [placement_rate(pl, n) for pl, n in zip(placement, sizes)]
    known = [r for r in rates if r is not None]
    for pl, r in zip(placement, rates):
        if r is not None:
            pl["chosen_ms_per_Mcell"] = round(r, 5)
            pl["chosen_vs_group_best"] = round(r / min(known), 4)
    return placement


def slowest_rank(placement):
    """The rank whose placement draw is the slowest of the group (it sets the pace), or None when nothing was placed."""
    if not isinstance(placement, list):
        return None
    known = [pl for pl in placement if pl.get("chosen_vs_group_best")]
    return max(known, key=lambda pl: pl["chosen_vs_group_best"])["rank"] if known else None


def best_stream_copy(device, src, dst, nb):
    """The same-device ceiling: the 4-in / 4-out copy of the sweeps' bytes in its two useful forms — plain, and with
    non-temporal loads and stores (the faster one depends on the size and the placement, profiles/r05_ab_nt.txt) — the better
    median of 5 launches after 2 warm-up launches each. GB/s."""
    import armon_amd
    best = 0.0
    for nt in (0, 3):
        armon_amd.lib().armon_hip_set_tuning(device.ctx, b"ARMON_COPY_NT", nt)
        ms = []
        for k in range(7):
            device.event_record(1000)
            device.stream_copy4(src, dst, nb)
            device.event_record(1001)
            ms.append(device.event_elapsed_ms(1000, 1001))
        best = max(best, 8 * nb / (sorted(ms[2:])[len(ms[2:]) // 2] * 1e-3) / 1e9)
    armon_amd.lib().armon_hip_set_tuning(device.ctx, b"ARMON_COPY_NT", 0)
    return best


def lines_check(test, rho):
    """For the test cases that vary along one axis only: (every row / column of this tile's density identical and finite,
    the waves have left the initial two states on this tile)."""
    import numpy as np
    line = rho[:, 0:1] if test == "Sod_y" else rho[0:1]
    return (bool(np.isfinite(rho).all() and np.array_equal(rho, np.broadcast_to(line, rho.shape))), bool(np.unique(line).size > 2))


def placement_rate(placement, cells):
    """ms per Mcell of the chosen draw (tiles of one grid may differ in size), None when the tile was not placed."""
    if not placement or not placement.get("chosen_ms"):
        return None
    return placement["chosen_ms"] / (cells / 1e6)


def redraw_placement(args, params, grid, dist, world, rank):
    """The slowest rank's placement draw sets the pace of a multi-GPU run: after the search of init_test the ranks compare
    their chosen draws (ms per cell), and a rank more than 3 % above the group's best searches one more round — a fresh
    batch of spare vectors next to the ones it has — before anything is timed. ARMON_BENCH_SLOW_RANK=r (test hook): rank r
    reports a draw twice as slow as it was."""
    if dist is None or world == 1:
        return
    cells = params.N[0] * params.N[1]
    mine = placement_rate(grid.placement, cells)
    if mine is not None and os.environ.get("ARMON_BENCH_SLOW_RANK") == str(rank):
        mine *= 2.0
    rates = [None] * world
    dist.all_gather_object(rates, mine)
    known = [r for r in rates if r is not None]
    if mine is None or not known or mine <= 1.03 * min(known):
        return
    first = grid.placement
    grid.placement = None
    from armon_amd.solver import init_test
    again = grid.tune_placement(keep_state=False)          # starts from the assignment it has, adds a fresh batch of spares
    init_test(params, grid, tune=False)
    if again is None:
        grid.placement = dict(first, redraw="failed (no memory for a fresh batch)")
    else:
        grid.placement = dict(again, tries=first["tries"] + again["tries"], redraw={"first_chosen_ms": first["chosen_ms"],
                              "group_best_ms_per_Mcell": round(min(known), 5), "was_ms_per_Mcell": round(mine, 5)})


def run_workload(args, dist, world, rank, local_rank, P, N_global, scaling, live_traffic=None, primary=True):
    """Build the grid of one workload on this rank's GPU, choose the halo transport, warm up, time args.steps cycles
    (barrier + device synchronisation on both sides, maximum over ranks) and return what the line reports about it.
    ``primary``: the workload `value` is quoted on (the secondary one of an N > 1 run skips the copy ceiling and traffic)."""
    import armon_amd
    from armon_amd.solver import BlockGrid, init_test, solver_cycle
    params = armon_amd.ArmonParameters(
        test=args.test, N=N_global, scheme=args.scheme, riemann_limiter="minmod", projection="euler_2nd",
        axis_splitting="Sequential", nghost=4, maxtime=1e9, maxcycle=10 ** 9, silent=5,
        use_MPI=dist is not None, P=P, device_id=local_rank,
        use_fused_sweep=not args.staged, exact_arithmetic=args.exact, data_type="float32" if args.f32 else "float64",
        placement_min_bytes=PLACEMENT_MIN_BYTES)
    grid = BlockGrid(params)
    if dist is not None:
        from armon_amd.halo_exchange import setup
        setup(params, grid)
    init_test(params, grid)
    redraw_placement(args, params, grid, dist, world, rank)
    gdt = grid.global_dt
    gdt.reset()

    # N > 1 over RCCL: three ways to move the halos, fastest first — (1) the library's own multi-GPU entry points
    # (armon_hip_halo_exchange_start/finish + armon_hip_dt_allreduce: RCCL send/recv on a transfer stream, events
    # only), (2) torch.distributed's RCCL ordered on the kernels' stream, (3) torch.distributed with host
    # synchronisation (the reference's MPI protocol). (1) and (2) never wait on the host, so they are checked on THIS
    # machine against (3) before anything is timed: five cycles from the same initial state must give the same dt
    # sequence and the same global mass / energy bit for bit on every rank; the first that does is used and named.
    halo_mode = None
    # A transport downgrade must be LOUD (it would cost the multi-GPU target without failing anything): the line carries
    # halo_exchange_downgraded (true when N > 1 over RCCL and anything but the library's native exchange is timed) and
    # halo_exchange_error (every exception text and self-check miss on the way); --require-native turns it into exit 3.
    halo_downgraded, halo_errors = False, []
    if dist is not None and grid.comm is not None:
        if dist.get_backend() == "nccl" and not getattr(grid.comm, "native", False):
            halo_errors.append("native exchange not initialised: " + str(getattr(params, "native_halo_error", "unknown reason")))
        from armon_amd.halo_exchange import HaloExchanger, allreduce_min
        from armon_amd.solver import conservation_vars, drain_halo
        native_comm = grid.comm if getattr(grid.comm, "native", False) else None
        torch_comm = grid.comm if native_comm is None else HaloExchanger(params, grid)
        halo_mode = "host-synchronised (gloo rehearsal: every rank on one GPU, host staging)"
        if torch_comm.stream_ordered or native_comm is not None:

            def probe(comm, stream_ordered, cycles=5):
                try:
                    drain_halo(grid)             # whatever a previous candidate left posted (it may have failed half-way)
                except Exception:
                    pass
                grid.comm = comm
                if comm is torch_comm:
                    comm.stream_ordered = stream_ordered
                init_test(params, grid, tune=False)
                gdt.reset()
                grid.dt_inflight.clear()
                dts = []
                for _ in range(cycles):
                    solver_cycle(params, grid, last_cycle=False)
                    dts.append(float(gdt.current_dt))
                    gdt.next_cycle()
                drain_halo(grid)
                params.wait()
                return dts, conservation_vars(params, grid)

            could_order = torch_comm.stream_ordered
            ref = probe(torch_comm, False)
            halo_mode, chosen = None, None
            if primary:
                # Insurance. The candidates below have never run on this machine's links before this process: if one of
                # them HANGS (a collective that never completes), nothing after it is reached and the watchdog ends the run.
                # So the workload is timed once, now, over the transport that has just completed five cycles — the
                # host-synchronised protocol — and the watchdog prints THAT line (downgraded, loudly, exit status 4)
                # instead of nothing. A few dozen milliseconds.
                grid.comm, torch_comm.stream_ordered = torch_comm, False
                init_test(params, grid, tune=False)
                gdt.reset()
                grid.dt_inflight.clear()
                solver_cycle(params, grid, last_cycle=False)
                gdt.next_cycle()
                params.wait()
                dist.barrier()
                t_ins = time.perf_counter()
                for _ in range(args.steps):
                    solver_cycle(params, grid, last_cycle=False)
                    gdt.next_cycle()
                drain_halo(grid)
                params.wait()
                dist.barrier()
                t_ins = time.perf_counter() - t_ins
                import torch
                t = torch.tensor([t_ins], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                INSURANCE.update(elapsed=float(t.item()), N_global=tuple(N_global), P=tuple(P), tile=tuple(params.N), scaling=scaling,
                                 halo_mode="torch.distributed RCCL, host-synchronised (the reference's MPI protocol) — the INSURANCE "
                                           "measurement taken before the faster transports were tried")
            # (the library's exchange twice: whole cycles enqueued by ONE call — armon_hip_mgpu_cycle, packs and the dt chain on
            # the transfer stream — and, should that form misbehave on this machine, the same exchange driven call by call)
            native_name = ("native (armon_hip_halo_exchange over RCCL send/recv on a transfer stream, unpack and boundary "
                           "strips on that stream too, dt all-reduce by the library; device-ordered; ")
            candidates = ([(native_name + "whole cycles enqueued by armon_hip_mgpu_cycle)", native_comm, True, True),
                           (native_name + "driven call by call)", native_comm, True, False)] if native_comm else [])
            if could_order:
                candidates.append(("torch.distributed RCCL, stream-ordered", torch_comm, True, False))
            failed = []
            for name, comm, so, one_call in candidates:
                params.native_cycle = one_call
                PHASE["name"] = "self-check of the transport: " + name[:60]
                if os.environ.get("ARMON_BENCH_HANG_CANDIDATE") == "1":          # test hook of the insurance line
                    time.sleep(10 ** 6)
                try:
                    same = 1.0 if probe(comm, so) == ref else 0.0
                    if same == 0.0:
                        halo_errors.append(f"{name.split(' ')[0]}: self-check against the host-synchronised protocol FAILED on rank {rank}")
                except Exception as e:          # a transport that cannot run here must not cost the bench line
                    same = 0.0
                    failed.append(f"{name.split(' ')[0]}: {type(e).__name__}")
                    halo_errors.append(f"{name.split(' ')[0]}: {type(e).__name__}: {str(e)[:300]}")
                if allreduce_min(params, same) == 1.0:
                    halo_mode, chosen = name + " — self-check against the host-synchronised protocol passed", (comm, so, one_call)
                    break
                failed.append(name.split(" ")[0] + (" (one call per cycle)" if one_call else "") + " self-check FAILED")
            if chosen is None:
                chosen = (torch_comm, False, False)
                halo_mode = "torch.distributed RCCL, host-synchronised (" + "; ".join(failed) + ")"
            elif failed:
                halo_mode += " (" + "; ".join(failed) + ")"
            grid.comm = chosen[0]
            params.native_cycle = chosen[2]
            if chosen[0] is torch_comm:
                torch_comm.stream_ordered = chosen[1]
            init_test(params, grid, tune=False)
            gdt.reset()
            grid.dt_inflight.clear()

    if dist is not None and dist.get_backend() == "nccl":
        halo_downgraded = not getattr(grid.comm, "native", False)       # what is timed below is not the library's own exchange

    from armon_amd.solver import conservation_vars
    mass0, energy0 = conservation_vars(params, grid)         # self-check of the timed work, see below

    dominant = ("sweep_x", "sweep_y") if not args.staged else ("euler_projection",)
    timer = EventTimer(params.device, dominant)
    params.kernel_callbacks.append(timer)

    def barrier():
        params.wait()
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver_cycle(params, grid, last_cycle=False)
        gdt.next_cycle()
    barrier()
    # Dominant kernels are timed with HIP events for the first cycles_timed cycles of the timed region (the event
    # pool bounds it: a tile with two remote sides launches 3 kernels per sweep — interior + 2 boundary strips).
    cycles_timed = min(args.steps, timer.max_pairs // 6)
    t0 = time.perf_counter()
    for i in range(args.steps):
        timer.enabled = i < cycles_timed
        solver_cycle(params, grid, last_cycle=False)
        gdt.next_cycle()
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    from armon_amd.solver import drain_halo
    drain_halo(grid)                     # the exchange posted ahead for the cycle that will not run

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Self-check that the timed cycles did the work: global mass and energy against the initial state, and (for the
    # test cases that vary along one axis only) every row / column of this rank's density identical and finite.
    import numpy as np
    mass1, energy1 = conservation_vars(params, grid)
    self_check = {"mass_drift": abs(mass1 - mass0) / abs(mass0), "energy_drift": abs(energy1 - energy0) / abs(energy0)}
    if args.test in ("Sod", "Sod_y", "Bizarrium"):
        ident, moved = lines_check(args.test, grid.real_view(grid.data["rho"].to_host()))
        if dist is not None:                 # every rank's tile: identical lines everywhere, the waves moved somewhere
            import torch
            t = torch.tensor([float(ident), -float(moved)], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ident, moved = bool(t[0].item()), bool(-t[1].item())
        self_check["lines_identical"], self_check["moved"] = ident, moved

    # Practical ceiling on THIS device: the same bytes (4 arrays read + 4 written) as a plain copy, no arithmetic.
    copy_gbps = None
    if grid.alt is not None and primary:
        from armon_amd.solver import STATE_VARS
        src, dst = [grid.data[f] for f in STATE_VARS], [grid.alt[f] for f in STATE_VARS]
        nb = src[0].nbytes & ~15
        copy_gbps = best_stream_copy(params.device, src, dst, nb)

    placement = grid.placement
    if dist is not None:                      # every rank draws its own placement: report them all (the slowest sets the pace)
        gathered = [None] * world
        dist.all_gather_object(gathered, grid.placement)
        placement = [dict(rank=r, **(g or {"tries": 0})) for r, g in enumerate(gathered)]
        sizes = [None] * world
        dist.all_gather_object(sizes, params.N[0] * params.N[1])
        placement = annotate_placements(placement, sizes)

    cells_local = params.N[0] * params.N[1]
    cells_total = N_global[0] * N_global[1]
    sweeps = 2 * args.steps
    value = cells_total * sweeps / elapsed / 1e6

    durs = timer.durations_ms()
    all_ms = [d for v in durs.values() for d in v]
    # one "launch" of the roofline = one whole sweep of the tile (its interior + boundary-strip launches together);
    # the staged mode times euler_projection, once per sweep as well
    sweeps_timed = 2 * cycles_timed
    mean_ms = sum(all_ms) / max(sweeps_timed, 1)
    bpc = B_PER_CELL[dominant[0]] // (2 if args.f32 else 1)
    achieved = bpc * cells_local / (mean_ms * 1e-3) / 1e9 if all_ms else 0.0
    traffic, traffic_source = pmc_traffic(args, world, N_global) if primary else (None, None)
    if live_traffic is not None:
        if live_traffic[0] is not None:
            traffic, traffic_source = live_traffic
        else:
            traffic_source = (traffic_source or "none") + f" (live measurement failed: {live_traffic[1]})"
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": "+".join(dominant), "bytes_per_cell": bpc, "launches_timed": len(all_ms),
                "sweeps_timed": sweeps_timed, "mean_launch_ms": round(mean_ms, 4),
                "per_kernel_ms": {k: round(sum(v) / max(cycles_timed * (2 if args.staged else 1), 1), 4)
                                  for k, v in durs.items()}}     # per sweep of that axis (staged: per call)
    # each kernel against the peak on its own (the slower one is the dominant kernel of the contract; `frac` above is the mean sweep)
    roofline["per_kernel_frac"] = {k: round(bpc * cells_local / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                   for k, ms in roofline["per_kernel_ms"].items() if ms > 0 and not args.staged}
    if copy_gbps:
        roofline["stream_copy_GBps_this_device"] = round(copy_gbps, 1)     # measured right after the timed region
        roofline["frac_of_stream_copy"] = round(achieved / copy_gbps, 4)
    res = dict(value=value, elapsed=elapsed, roofline=roofline, self_check=self_check, placement=placement,
               halo_mode=halo_mode, halo_downgraded=halo_downgraded, halo_errors=halo_errors, tile=tuple(params.N),
               cells_local=cells_local, cells_total=cells_total, sweeps=sweeps, device=params.device.name)
    # release this workload's communicators and vectors before the next one is built
    if dist is not None:
        for comm in {id(c): c for c in (grid.comm, locals().get("native_comm"), locals().get("torch_comm")) if c is not None}.values():
            if hasattr(comm, "close"):
                comm.close()                   # the library's RCCL communicators, before the launcher's
    params.kernel_callbacks.remove(timer)
    del grid
    import gc
    gc.collect()
    return res



def run_peer_workload(args, n_dev, P, N_global, device_ids, primary=True, live_traffic=None):
    """The same measurement with every tile of the process grid in THIS process (--transport peer): tile r on device
    device_ids[r] through the library's in-process group (armon_hip_mgpu_init: one hipMemcpyPeerAsync per face over xGMI,
    the dt minimum gathered on tile 0's device), a whole cycle enqueued by one armon_hip_mgpu_cycle call with one host
    thread per device. No RCCL, no launcher, no torch."""
    import numpy as np
    from armon_amd.multi_tile import TileGroup
    from armon_amd import solver as S
    group = TileGroup(P, device_ids=device_ids, test=args.test, N=N_global, scheme=args.scheme, riemann_limiter="minmod",
                      projection="euler_2nd", axis_splitting="Sequential", nghost=4, maxtime=1e9, maxcycle=10 ** 9, silent=5,
                      use_fused_sweep=True, exact_arithmetic=args.exact, data_type="float32" if args.f32 else "float64",
                      placement_min_bytes=PLACEMENT_MIN_BYTES)
    try:
        group.init_test()
        sizes = [p.N[0] * p.N[1] for p in group.params]
        # the slowest tile's placement draw sets the pace: one more round for a tile > 3 % above the group's best
        rates = [placement_rate(g.placement, n) for g, n in zip(group.grids, sizes)]
        known = [r for r in rates if r is not None]
        for p, g, r in zip(group.params, group.grids, rates):
            if r is not None and r > 1.03 * min(known):
                first, g.placement = g.placement, None
                again = g.tune_placement(keep_state=False)
                S.init_test(p, g, tune=False)
                g.placement = (dict(again, tries=first["tries"] + again["tries"], redraw={"first_chosen_ms": first["chosen_ms"]})
                               if again else dict(first, redraw="failed"))
        gdt = group.global_dt
        gdt.reset()
        group.dt_inflight.clear()
        mass0, energy0 = group.conservation_vars()
        root = group.root
        timer = EventTimer(root.device, ("sweep_x", "sweep_y"))
        root.kernel_callbacks.append(timer)
        for _ in range(args.warmup):
            group.solver_cycle(last_cycle=False)
            gdt.next_cycle()
        group.wait()
        cycles_timed = min(args.steps, timer.max_pairs // 2)
        t0 = time.perf_counter()
        for i in range(args.steps):
            timer.enabled = i < cycles_timed
            group.solver_cycle(last_cycle=False)
            gdt.next_cycle()
        group.wait()
        elapsed = time.perf_counter() - t0
        timer.enabled = False
        group.drain()
        group.wait()
        mass1, energy1 = group.conservation_vars()
        self_check = {"mass_drift": abs(mass1 - mass0) / abs(mass0), "energy_drift": abs(energy1 - energy0) / abs(energy0)}
        if args.test in ("Sod", "Sod_y", "Bizarrium"):
            checks = [lines_check(args.test, g.real_view(g.data["rho"].to_host())) for g in group.grids]
            self_check["lines_identical"] = all(c[0] for c in checks)
            self_check["moved"] = any(c[1] for c in checks)
        copy_gbps = None
        g0 = group.grids[0]
        if primary and g0.alt is not None:
            src, dst = [g0.data[f] for f in S.STATE_VARS], [g0.alt[f] for f in S.STATE_VARS]
            nb = src[0].nbytes & ~15
            copy_gbps = best_stream_copy(root.device, src, dst, nb)
        placement = annotate_placements([dict(rank=r, **(g.placement or {"tries": 0})) for r, g in enumerate(group.grids)], sizes)
        cells_local, cells_total = sizes[0], N_global[0] * N_global[1]
        sweeps = 2 * args.steps
        value = cells_total * sweeps / elapsed / 1e6
        durs = timer.durations_ms()
        all_ms = [d for v in durs.values() for d in v]
        sweeps_timed = 2 * cycles_timed
        mean_ms = sum(all_ms) / max(sweeps_timed, 1)
        bpc = B_PER_CELL["sweep_x"] // (2 if args.f32 else 1)
        achieved = bpc * cells_local / (mean_ms * 1e-3) / 1e9 if all_ms else 0.0
        traffic, traffic_source = (live_traffic if live_traffic and live_traffic[0] is not None else (None, None))
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": "sweep_x+sweep_y", "bytes_per_cell": bpc, "launches_timed": len(all_ms),
                    "sweeps_timed": sweeps_timed, "mean_launch_ms": round(mean_ms, 4),
                    "timed_on": "tile 0: events around each whole sweep (interior + boundary strips) on its compute stream",
                    "per_kernel_ms": {k: round(sum(v) / max(cycles_timed, 1), 4) for k, v in durs.items()}}
        roofline["per_kernel_frac"] = {k: round(bpc * cells_local / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                       for k, ms in roofline["per_kernel_ms"].items() if ms > 0}
        if copy_gbps:
            roofline["stream_copy_GBps_this_device"] = round(copy_gbps, 1)
            roofline["frac_of_stream_copy"] = round(achieved / copy_gbps, 4)
        threads = os.environ.get("ARMON_MGPU_THREADS", "1") != "0"
        mode = ("in-process tile group (armon_hip_mgpu_init, device_ids " + str(list(device_ids)) + "): one hipMemcpyPeerAsync per "
                "face, whole cycles enqueued by armon_hip_mgpu_cycle with " + ("one host thread per tile" if threads else "the calling thread only")
                + "; no RCCL, no launcher")
        return dict(value=value, elapsed=elapsed, roofline=roofline, self_check=self_check, placement=placement,
                    halo_mode=mode, halo_downgraded=False, halo_errors=[], tile=tuple(group.params[0].N),
                    cells_local=cells_local, cells_total=cells_total, sweeps=sweeps, device=root.device.name)
    finally:
        group.close()


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def run_child(cmd, env, timeout):
    """Run a child in its own process group, relay its stderr, return (rc, stdout text); on timeout kill the GROUP we started
    (never a pattern) and return rc = -9."""
    import signal
    import subprocess
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True, cwd=ROOT)
    try:
        out, _ = child.communicate(timeout=timeout)
        return child.returncode, out
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)
            except ProcessLookupError:
                break
            try:
                child.communicate(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        out = ""
        try:
            out = child.communicate(timeout=5)[0] or ""
        except Exception:
            pass
        return -9, out


def last_json_line(text):
    for line in reversed([l for l in (text or "").splitlines() if l.strip()]):
        try:
            d = json.loads(line)
            if isinstance(d, dict) and "metric" in d:
                return d
        except ValueError:
            continue
    return None


def launch(args, argv, real_stdout):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE): this process has NOT touched a GPU and will not — it
    starts the measurement as child processes and relays the one line. (1) unless --transport peer: the N ranks under
    torch.distributed.run on 127.0.0.1 (one process per GPU over RCCL — the driver's own command shape); (2) when that
    produced no line, or with --transport peer: one child that drives the N devices itself (in-process tile group). A
    process that has initialised the GPU is never re-executed; a hung child is killed by its process group."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ARMON_BENCH_CHILD"] = "1"
    me = os.path.abspath(__file__)
    rest = [a for a in argv]
    notes = []
    if args.transport != "peer":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), me, *rest]
        rc, out = run_child(cmd, env, args.launch_timeout)
        line = last_json_line(out)
        if line is not None:
            line.setdefault("config", {})["launched_by"] = "bench.py itself: child torch.distributed.run, one rank per GPU"
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
            return rc if rc != -9 else 4
        notes.append(f"rank launch under torch.distributed.run gave no line (exit status {rc}"
                     + (f", killed after {args.launch_timeout:.0f} s" if rc == -9 else "") + ")")
        print("bench.py: " + notes[-1] + "; falling back to --transport peer", file=sys.stderr)
        if args.transport == "rccl":
            return rc if rc not in (0, -9) else 4
    cmd = [sys.executable, me, *[a for a in rest if a not in ("--transport", "rccl", "auto")], "--transport", "peer"]
    if args.transport == "peer":
        cmd = [sys.executable, me, *rest]
    env["ARMON_BENCH_PEER_CHILD"] = "1"
    if notes:
        env["ARMON_BENCH_FALLBACK_NOTE"] = "; ".join(notes)
    rc, out = run_child(cmd, env, args.launch_timeout)
    line = last_json_line(out)
    if line is None:
        print(f"bench.py: the in-process (peer) run gave no line either (exit status {rc})", file=sys.stderr)
        return rc if rc not in (0, -9) else 4
    line.setdefault("config", {})["launched_by"] = "bench.py itself: one child process driving every device (--transport peer)"
    os.write(real_stdout, (json.dumps(line) + "\n").encode())
    return rc if rc != -9 else 4


PHASE = {"name": "start"}        # what the process was doing, for the watchdog's message
INSURANCE = {}                   # run_workload: the primary workload timed over the most conservative transport, see there


def exit_now(code):
    """End a process that may be stuck in a collective: no teardown (it could block), a status the launcher sees."""
    sys.stderr.flush()
    os._exit(code)


def main():
    # The contract is ONE JSON line on stdout. RCCL and gloo print banners on fd 1 while they initialise: send
    # everything written to fd 1 during the run to stderr and keep the real stdout for that line alone.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=16384, dest="n",
                    help="cells per axis of the GLOBAL grid (BASELINE's 16384² Sod; with --weak: per GPU)")
    ap.add_argument("--test", default="Sod")
    ap.add_argument("--scheme", default="GAD")
    ap.add_argument("--staged", action="store_true", help="5 staged kernels per sweep instead of the fused one")
    ap.add_argument("--exact", action="store_true",
                    help="IEEE division/sqrt, no contraction (bit-identical to the CPU oracle) instead of the "
                         "default tuned arithmetic (shared 1-ulp reciprocals + FMA, within the reference's tolerance)")
    ap.add_argument("--fast", action="store_true", help="(default) tuned arithmetic; kept for compatibility")
    ap.add_argument("--f32", action="store_true", help="Float32 data_type (the _f32 entry points) instead of the fp64 headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="(default at --gpus 1 when rocprofv3 is on PATH) roofline.traffic from two child rocprofv3 --pmc "
                         "passes of this command (adds ≈1 min)")
    ap.add_argument("--no-measure-traffic", action="store_true",
                    help="roofline.traffic replayed from the committed passes under profiles/ instead of measured for this line")
    ap.add_argument("--global", dest="global_grid", default=None, metavar="NXxNY",
                    help="GLOBAL grid, split over the process grid (tiles = N÷P, remainder on the last tile, ref "
                         "src/parameters.jl:673-697); default: --cells²")
    ap.add_argument("--grid", default=None, metavar="PXxPY", help="process grid (default: 1x1, 2x1, 2x2, 4x2 for 1, 2, 4, 8 ranks)")
    ap.add_argument("--transport", choices=("auto", "rccl", "peer"), default="auto",
                    help="N > 1 — rccl: one process per GPU (started by this process when no launcher did), halos over RCCL "
                         "send/recv; peer: ONE process, N devices, hipMemcpyPeerAsync through the library's in-process "
                         "group; auto (default): rccl, and peer when the rank launch gives no line")
    ap.add_argument("--launch-timeout", type=float, default=1500.,
                    help="seconds a child run started by this process may take before its process group is killed")
    ap.add_argument("--timeout", type=float, default=600.,
                    help="seconds the primary workload may take inside a rank before the process gives up (exit status 4)")
    ap.add_argument("--strong", action="store_true", help="(default) the global grid stays --cells² (or --global) whatever the number of GPUs")
    ap.add_argument("--weak", action="store_true", help="weak scaling only: --cells² cells PER GPU is the (single) workload")
    ap.add_argument("--no-weak", "--no-strong", dest="no_second", action="store_true",
                    help="N > 1, default workload: skip the weak-scaling workload timed after the strong one")
    ap.add_argument("--weak-timeout", "--strong-timeout", dest="second_timeout", type=float, default=180.,
                    help="seconds the second (weak) workload may take before the line is printed without it")
    ap.add_argument("--require-native", action="store_true",
                    help="N > 1: exit with status 3 (after printing the line) when the halos do not travel through the "
                         "library's own RCCL exchange, i.e. when config.halo_exchange_downgraded is true")
    ap.add_argument("--config", type=int, choices=(2, 3, 4, 5), default=None,
                    help="BASELINE.json configs[N-1]: 2 = Sod 8192² Godunov, 3 = Sedov 16384², "
                         "4 = Sod 32768x16384 on 2x2, 5 = Bizarrium 32768² on 4x2")
    argv = sys.argv[1:]
    args = ap.parse_args(argv)
    if args.config == 2:
        args.test, args.scheme, args.n = "Sod", "Godunov", 8192
    elif args.config == 3:
        args.test, args.n = "Sedov", 16384
    elif args.config == 4:
        args.test, args.global_grid, args.grid = "Sod", "32768x16384", "2x2"
    elif args.config == 5:
        args.test, args.global_grid, args.grid = "Bizarrium", "32768x32768", "4x2"

    launched = "WORLD_SIZE" in os.environ
    if launched and os.environ.get("ARMON_BENCH_FAIL_RANKS") == "1" and os.environ.get("ARMON_BENCH_CHILD") == "1":
        sys.exit("injected failure of the rank launch (ARMON_BENCH_FAIL_RANKS)")          # test hook of the fallback
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    peer = False
    if launched:
        if args.transport == "peer" and world > 1:
            sys.exit("--transport peer is ONE process driving every device: start it without a launcher")
        args.gpus = world                 # one of the ranks of somebody's launcher
    elif args.gpus > 1:
        if args.staged:
            sys.exit("--staged is a single-GPU measurement")
        n_tiles = args.gpus
        P_ = tuple(int(v) for v in args.grid.lower().split("x")) if args.grid else None
        if P_ is not None and (len(P_) != 2 or P_[0] * P_[1] != n_tiles):
            sys.exit(f"--grid {args.grid}: {n_tiles} rank(s) cannot form that process grid")
        if os.environ.get("ARMON_BENCH_PEER_CHILD") == "1" and args.transport == "peer":
            peer = True                   # the child that drives every device itself
        else:
            sys.exit(launch(args, argv, real_stdout))          # parent: never touches a GPU
    world_tiles = args.gpus

    from armon_amd.parameters import proc_grid_for
    P = tuple(int(v) for v in args.grid.lower().split("x")) if args.grid else proc_grid_for(world_tiles)   # (px, py), e.g. 8 → (4, 2)
    if len(P) != 2 or P[0] * P[1] != world_tiles:
        sys.exit(f"--grid {args.grid}: {world_tiles} rank(s) cannot form that process grid")

    # roofline.traffic is MEASURED for the line by default (N = 1): two child passes of this command under rocprofv3 --pmc,
    # before this process touches the GPU. Not from inside a profiler (tools/profile_*.sh run this file under rocprofv3:
    # a profiler in a profiled child is refused on this pool), not in the children themselves; the committed passes are
    # then replayed, and the line says which it was (roofline.traffic_source).
    live_traffic = None
    if world_tiles == 1 and not args.no_measure_traffic and os.environ.get("ARMON_BENCH_FORCE_DIST") != "1":
        live_traffic = (None, "running under a profiler") if under_profiler() else measure_traffic(args)

    # a rank stuck in a collective (a peer died, a transport deadlocked) must end with a status the launcher sees
    import threading

    def primary_watchdog():
        print(f"bench.py rank {rank}: no result after {args.timeout:.0f} s (phase: {PHASE['name']}); giving up", file=sys.stderr)
        if rank == 0 and INSURANCE:
            # a transport hung after the insurance measurement: that measurement is the line (and the status says failure)
            i = INSURANCE
            cells, sweeps = i["N_global"][0] * i["N_global"][1], 2 * args.steps
            ach = 64 * i["tile"][0] * i["tile"][1] * sweeps / i["elapsed"] / 1e9
            line = {"metric": "Mcells/sec per sweep (fp64)", "value": round(cells * sweeps / i["elapsed"] / 1e6, 1), "unit": "Mcells/s",
                    "n_gpus": world_tiles, "steps": args.steps, "warmup": 1, "ms_per_step": round(i["elapsed"] / args.steps * 1e3, 4),
                    "higher_is_better": True, "scaling": i["scaling"], "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                    "config": {"workload": f"{args.test} {i['N_global'][0]}x{i['N_global'][1]} fp64, {args.scheme}+minmod+euler_2nd, Sequential X,Y "
                                           f"splitting, nghost 4, {i['P'][0]}x{i['P'][1]} tiles of {i['tile'][0]}x{i['tile'][1]} cells ({i['scaling']} scaling)",
                               "halo_exchange": i["halo_mode"], "halo_exchange_downgraded": True,
                               "halo_exchange_error": f"no result after {args.timeout:.0f} s in phase: {PHASE['name']}"},
                    "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                                 "traffic": None, "kernel": "whole cycle (sweeps + exchange), wall clock: the insurance measurement has no per-kernel events"}}
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        exit_now(4)

    guard = threading.Timer(args.timeout, primary_watchdog)
    guard.daemon = True
    guard.start()

    dist = None
    rehearsal = os.environ.get("ARMON_BENCH_REHEARSAL") == "1"
    if not peer and (world > 1 or os.environ.get("ARMON_BENCH_FORCE_DIST") == "1"):   # FORCE_DIST: exercise RCCL init/all-reduce with one rank
        PHASE["name"] = "process-group initialisation"
        import torch
        import torch.distributed as dist
        if world == 1:                 # ARMON_BENCH_FORCE_DIST without a launcher: a one-rank rendezvous on this host
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
        # Rehearsal knob (one-GPU box): ARMON_BENCH_REHEARSAL=1 puts every rank on cuda:0 over gloo with host
        # staging, to exercise this code path; real runs use one GPU per rank over RCCL.
        if rehearsal:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            # one GPU per rank: LOCAL_RANK-th of the devices this process can see — or the only one, when the launcher
            # narrowed every rank's view to its own GPU (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank)
            visible = torch.cuda.device_count()
            if visible < 1:
                sys.exit(f"bench.py rank {rank}: no GPU visible")
            local_rank = local_rank % visible
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # Workloads. BASELINE.json's metric is quoted on "16384² Sod, 1/2/4/8 MI355X": the same grid at every N — the STRONG
    # curve — is `value`. The weak workload (that grid per GPU) follows at N > 1 as a side field.
    if args.global_grid:
        N_primary = tuple(int(v) for v in args.global_grid.lower().split("x"))
        scaling = "strong"                                    # a named global grid: total work is fixed
        second = None
    elif args.weak:
        N_primary = (args.n * P[0], args.n * P[1])            # weak scaling: n×n cells per GPU
        scaling = "weak"
        second = None
    else:
        N_primary = (args.n, args.n)                          # the same n×n grid whatever the GPU count
        scaling = "strong"
        second = (args.n * P[0], args.n * P[1]) if world_tiles > 1 and not args.no_second else None
    if os.environ.get("ARMON_BENCH_FORCE_SECOND") == "1" and second is None and not args.no_second:
        second = N_primary               # (with ARMON_BENCH_FORCE_DIST=1: the second workload with ONE rank over RCCL — tearing
        #                                   the library's communicators down and building them again inside one process)

    device_ids = [0] * world_tiles if rehearsal else list(range(world_tiles))
    if peer and not rehearsal:
        import ctypes
        import armon_amd
        count = ctypes.c_int()
        armon_amd.lib().armon_hip_device_count(ctypes.byref(count))
        if count.value < world_tiles:
            sys.exit(f"--transport peer --gpus {world_tiles}: this process sees {count.value} GPU(s) "
                     "(ARMON_BENCH_REHEARSAL=1 puts every tile on device 0: a rehearsal, not a measurement)")

    def run(N_global, kind, primary):
        PHASE["name"] = f"{kind} workload {N_global[0]}x{N_global[1]}"
        if os.environ.get("ARMON_BENCH_FAIL_SECOND") == "1" and not primary:
            raise RuntimeError("injected failure of the second workload (ARMON_BENCH_FAIL_SECOND)")
        if os.environ.get("ARMON_BENCH_HANG_SECOND") == "1" and not primary:
            time.sleep(10 ** 6)
        if peer:
            return run_peer_workload(args, world_tiles, P, N_global, device_ids, primary=primary)
        return run_workload(args, dist, world, rank, local_rank, P, N_global, kind, live_traffic if primary else None, primary=primary)

    try:
        r = run(N_primary, scaling, True)
    except BaseException:
        import traceback
        print(f"bench.py rank {rank}: the primary workload failed:\n{traceback.format_exc()}", file=sys.stderr)
        exit_now(4)               # other ranks may sit in a collective this one will never join: no teardown
    guard.cancel()
    value, elapsed, roofline, self_check = r["value"], r["elapsed"], r["roofline"], r["self_check"]
    halo_mode, halo_downgraded, halo_errors, placement = r["halo_mode"], r["halo_downgraded"], r["halo_errors"], r["placement"]
    cells_local, cells_total, sweeps, tile = r["cells_local"], r["cells_total"], r["sweeps"], r["tile"]

    prec = "fp32" if args.f32 else "fp64"
    out = {
        "metric": f"Mcells/sec per sweep ({prec})", "value": round(value, 1), "unit": "Mcells/s",
        "n_gpus": world_tiles, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None, "dtype": "f32" if args.f32 else "f64", "data": "synthetic",
        "config": {"workload": f"{args.test} {N_primary[0]}x{N_primary[1]} {prec}, {args.scheme}+minmod+euler_2nd, "
                               f"Sequential X,Y splitting, nghost 4, {P[0]}x{P[1]} tiles of {tile[0]}x{tile[1]} cells "
                               f"({scaling} scaling)",
                   "baseline_config": args.config,
                   "path": "staged (5 kernels/sweep)" if args.staged else "fused sweep",
                   "arithmetic": "exact (IEEE div/sqrt, no contraction; bit-identical to the CPU oracle, subnormal values included)" if args.exact
                   else "tuned (shared 1-ulp reciprocals + FMA; within the reference's golden tolerance)",
                   "process_grid": list(P), "sweeps_per_step": 2, "cells_per_gpu": cells_local,
                   "transport": "peer (one process, every device)" if peer else ("rccl ranks" if world > 1 else "single block"),
                   "hbm_placement": placement, "slowest_rank": slowest_rank(placement), "device": r["device"], "halo_exchange": halo_mode,
                   "halo_exchange_downgraded": halo_downgraded,
                   "halo_exchange_error": "; ".join(halo_errors) if halo_errors else None,
                   "launch_fallback": os.environ.get("ARMON_BENCH_FALLBACK_NOTE")},
        "hbm_GBps_algorithmic_whole_job": round((32 if args.f32 else 64) * cells_total * sweeps / elapsed / 1e9, 1),
        "roofline": roofline,
        "self_check": self_check,
    }
    # The second workload of a default N > 1 run: weak scaling (--cells² per GPU on the same process grid), timed in the same
    # processes right after the strong one and reported next to it. The line is complete at this point: a second workload
    # that raises on some rank, or hangs in a collective because it did, must not cost it — the line is printed as it stands
    # (`weak.error` says why), the cause goes to stderr on every rank, and the process ends with status 4: a failure is a
    # failure for the launcher and the driver, whatever was printed.
    exit_code = 0
    if second is not None:

        def print_line():
            if rank == 0:
                os.write(real_stdout, (json.dumps(out) + "\n").encode())

        def bail():
            print(f"bench.py rank {rank}: watchdog fired: the weak workload gave no result after {args.second_timeout:.0f} s "
                  f"(phase: {PHASE['name']}; a rank failed or a collective hung)", file=sys.stderr)
            out["weak"] = {"value": None, "error": f"no result after {args.second_timeout} s (a rank failed or a collective hung)"}
            print_line()
            exit_now(3 if (args.require_native and halo_downgraded) else 4)

        watchdog = threading.Timer(args.second_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            s2 = run(second, "weak", False)
            out["weak"] = {
                "value": round(s2["value"], 1), "unit": "Mcells/s", "ms_per_step": round(s2["elapsed"] / args.steps * 1e3, 4),
                "workload": f"{args.test} {second[0]}x{second[1]} on {P[0]}x{P[1]} tiles of {s2['tile'][0]}x{s2['tile'][1]} cells (weak scaling: --cells² per GPU)",
                "strong_vs_weak": round(value / s2["value"], 4),
                "note": "strong_vs_weak = `value` ÷ this: what splitting the 16384² grid costs against every GPU working on a full tile",
                "roofline_frac": s2["roofline"]["frac"], "per_kernel_ms": s2["roofline"]["per_kernel_ms"],
                "halo_exchange": s2["halo_mode"], "halo_exchange_downgraded": s2["halo_downgraded"],
                "self_check": s2["self_check"]}
            if s2["halo_downgraded"] and not halo_downgraded:
                halo_downgraded, halo_mode = True, s2["halo_mode"]
                halo_errors = halo_errors + [e for e in s2["halo_errors"] if e not in halo_errors]
                out["config"]["halo_exchange_downgraded"] = True
                out["config"]["halo_exchange_error"] = "; ".join(halo_errors) if halo_errors else None
        except BaseException as e:
            import traceback
            watchdog.cancel()
            print(f"bench.py rank {rank}: the weak workload failed:\n{traceback.format_exc()}", file=sys.stderr)
            out["weak"] = {"value": None, "error": f"{type(e).__name__}: {str(e)[:300]}"}
            print_line()          # the other ranks may be inside a collective this one will never join: no teardown
            exit_now(3 if (args.require_native and halo_downgraded) else 4)
        watchdog.cancel()
    if rank == 0 and world_tiles == 1 and not args.no_cpu_baseline and not args.f32:
        PHASE["name"] = "cpu baseline"
        try:
            out["cpu_baseline"] = cpu_baseline(args.test, args.scheme)
        except Exception as e:   # the baseline is a reported extra: never lose the GPU line over it
            out["cpu_baseline"] = {"value": None, "unit": "Mcells/s per sweep", "cores": 0,
                                   "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    if halo_downgraded:
        print(f"bench.py: HALO EXCHANGE DOWNGRADED on rank {rank}: {halo_mode}; {'; '.join(halo_errors)}", file=sys.stderr)
        if args.require_native:
            sys.exit(3)
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
