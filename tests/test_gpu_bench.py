"""bench.py's line is a contract with the driver (one JSON line on stdout: metric, value, roofline, cpu_baseline at N = 1;
at N > 1 the STRONG workload — BASELINE's grid split over the process grid — as `value` and the weak one next to it),
whoever launched it: a launcher (the driver's torch.distributed.run command) or nobody (a bare `python bench.py --gpus N`
starts its own ranks, or drives every device from one process with --transport peer). Small grids, seconds each."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_line(cmd, env=None, timeout=600, rc=0):
    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env={**clean, **(env or {})}, cwd=ROOT)
    assert r.returncode == rc, (r.returncode, r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE line, whatever the libraries print while they start
    return json.loads(lines[0])


def test_single_gpu_line_carries_the_contract():
    d = run_line([sys.executable, BENCH, "--cells", "1024", "--steps", "3", "--warmup", "1", "--no-measure-traffic"])
    assert d["metric"].startswith("Mcells/sec per sweep") and d["unit"] == "Mcells/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "Sod 1024x1024" in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - 1024 * 1024 * 2 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and ro["bytes_per_cell"] == 64
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and set(ro["per_kernel_ms"]) == {"sweep_x", "sweep_y"}
    assert "traffic" in ro and "weak" not in d and "strong" not in d
    assert set(ro["per_kernel_frac"]) == {"sweep_x", "sweep_y"} and all(0 < v < 1 for v in ro["per_kernel_frac"].values())
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    sc = d["self_check"]
    assert sc["mass_drift"] <= 1e-12 and sc["energy_drift"] <= 1e-12 and sc["lines_identical"] is True


def test_traffic_is_measured_for_the_line_by_default():
    """Two child passes under rocprofv3 --pmc before the process touches the GPU (the children themselves replay)."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    d = run_line([sys.executable, BENCH, "--cells", "2048", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    ro = d["roofline"]
    src = ro["traffic_source"] or ""
    if "live measurement failed" in src and ("rocprofv3 not found" in src or "ounters" in src and "unavailable" in src):
        pytest.skip("counter passes not available on this box: " + src)
    # any other failure of the live measurement is a regression of the counter path (the line carries the child's status
    # and the tail of its stderr): it must fail here, not hide behind a skip
    assert ro["traffic_source"].startswith("measured for this line"), ro["traffic_source"]
    algorithmic = 64 * 2048 * 2048
    assert 0.98 * algorithmic <= ro["traffic"] <= 1.25 * algorithmic, ro["traffic"] / algorithmic


def check_two_tile_line(d):
    """N = 2: `value` is the strong workload (--cells² split 2x1), the weak one (--cells² per GPU) sits next to it."""
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "Sod 512x512" in d["config"]["workload"]
    assert "2x1 tiles of 256x512" in d["config"]["workload"] and d["config"]["cells_per_gpu"] == 256 * 512
    assert d["value"] > 0 and abs(d["value"] - 512 * 512 * 2 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    assert d["self_check"]["mass_drift"] <= 1e-12 and d["self_check"]["lines_identical"] is True
    w = d["weak"]
    assert w["value"] > 0 and "Sod 1024x512 on 2x1 tiles of 512x512" in w["workload"]
    assert abs(w["strong_vs_weak"] - d["value"] / w["value"]) < 1e-3
    assert w["self_check"]["mass_drift"] <= 1e-12 and w["self_check"]["lines_identical"] is True
    assert set(d["roofline"]["per_kernel_ms"]) == {"sweep_x", "sweep_y"} and "slowest_rank" in d["config"]


REHEARSAL = {"ARMON_BENCH_REHEARSAL": "1"}            # every rank / tile on this one GPU (gloo between ranks)
SMALL = ["--gpus", "2", "--cells", "512", "--steps", "3", "--warmup", "1"]


def test_two_ranks_under_a_launcher_report_the_strong_and_the_weak_workload():
    """The driver's command shape: torch.distributed.run starts the ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", BENCH, *SMALL]
    check_two_tile_line(run_line(cmd, env=REHEARSAL))


def test_bare_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a bare shell: the parent starts the ranks as children and relays the one line."""
    d = run_line([sys.executable, BENCH, *SMALL], env=REHEARSAL)
    check_two_tile_line(d)
    assert "child torch.distributed.run" in d["config"]["launched_by"] and d["config"]["launch_fallback"] is None


def test_peer_transport_drives_every_tile_from_one_process():
    """--transport peer: one process, the library's in-process group, whole cycles enqueued by armon_hip_mgpu_cycle."""
    d = run_line([sys.executable, BENCH, *SMALL, "--transport", "peer"], env=REHEARSAL)
    check_two_tile_line(d)
    assert d["config"]["transport"].startswith("peer") and "armon_hip_mgpu_cycle" in d["config"]["halo_exchange"]
    assert "one host thread per tile" in d["config"]["halo_exchange"]
    d = run_line([sys.executable, BENCH, "--gpus", "4", "--cells", "512", "--steps", "3", "--warmup", "1", "--transport", "peer", "--no-weak"],
                 env={**REHEARSAL, "ARMON_MGPU_THREADS": "0"})
    assert d["n_gpus"] == 4 and "2x2 tiles of 256x256" in d["config"]["workload"] and "weak" not in d
    assert "the calling thread only" in d["config"]["halo_exchange"]
    # --weak: the weak workload alone (--cells² per GPU) is the line; --global names any other grid
    d = run_line([sys.executable, BENCH, *SMALL, "--transport", "peer", "--weak"], env=REHEARSAL)
    assert d["scaling"] == "weak" and "Sod 1024x512" in d["config"]["workload"] and "2x1 tiles of 512x512" in d["config"]["workload"] and "weak" not in d
    d = run_line([sys.executable, BENCH, "--gpus", "2", "--global", "768x512", "--grid", "1x2", "--steps", "2", "--warmup", "1", "--transport", "peer"],
                 env=REHEARSAL)
    assert d["scaling"] == "strong" and "1x2 tiles of 768x256" in d["config"]["workload"] and "weak" not in d


def test_failed_rank_launch_falls_back_to_the_peer_transport_loudly():
    """The automatic fallback: the rank launch gives no line (here: made to fail), the in-process transport runs instead and
    the line says so."""
    d = run_line([sys.executable, BENCH, *SMALL, "--no-weak"], env={**REHEARSAL, "ARMON_BENCH_FAIL_RANKS": "1"})
    assert d["n_gpus"] == 2 and d["config"]["transport"].startswith("peer")
    assert "rank launch under torch.distributed.run gave no line" in d["config"]["launch_fallback"]


def test_a_failing_second_workload_keeps_the_line_and_exits_non_zero():
    """A weak workload that raises (or hangs: the watchdog) must not cost the strong line — and must not pass for success."""
    for hook, text in (("ARMON_BENCH_FAIL_SECOND", "injected failure"), ("ARMON_BENCH_HANG_SECOND", "no result after")):
        d = run_line([sys.executable, BENCH, *SMALL, "--transport", "peer", "--weak-timeout", "5"], env={**REHEARSAL, hook: "1"}, rc=4)
        assert d["value"] > 0 and d["weak"]["value"] is None and text in d["weak"]["error"]


def test_a_slow_rank_redraws_its_placement():
    """Every rank's chosen placement draw is compared with the group's best; a rank more than 3 % above it searches one more
    round before anything is timed, and the line names the slowest rank."""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--cells", "2048", "--steps", "2", "--warmup", "1", "--no-weak"]
    d = run_line(cmd, env={**REHEARSAL, "ARMON_BENCH_SLOW_RANK": "1", "ARMON_BENCH_PLACEMENT_MIN_BYTES": "1"})
    pl = {p["rank"]: p for p in d["config"]["hbm_placement"]}
    assert "redraw" in pl[1] and pl[1]["tries"] > pl[1]["max_tries"]      # (rank 0 may redraw too: two ranks share this GPU, timings are noisy)
    assert d["config"]["slowest_rank"] in (0, 1) and all("chosen_vs_group_best" in p for p in pl.values())


def test_one_rank_over_real_rccl_runs_both_workloads_through_the_native_cycle():
    """What one GPU can execute of the real N > 1 rank path: torch.distributed over RCCL (nccl backend) with ONE rank, the
    library's own RCCL group (two communicators, the all-reduce of the next CFL step on the transfer stream), whole cycles
    through armon_hip_mgpu_cycle, the transports' self-check — and the SECOND workload in the same process, which tears the
    library's communicators down and builds them again (ARMON_BENCH_FORCE_DIST / ARMON_BENCH_FORCE_SECOND)."""
    d = run_line([sys.executable, BENCH, "--cells", "1024", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                 env={"ARMON_BENCH_FORCE_DIST": "1", "ARMON_BENCH_FORCE_SECOND": "1"})
    assert d["n_gpus"] == 1 and d["config"]["halo_exchange_downgraded"] is False, d["config"]["halo_exchange_error"]
    assert d["config"]["halo_exchange"].startswith("native (armon_hip_halo_exchange over RCCL")
    assert "whole cycles enqueued by armon_hip_mgpu_cycle" in d["config"]["halo_exchange"]
    assert d["weak"]["value"] > 0 and d["weak"]["halo_exchange_downgraded"] is False
    assert d["self_check"]["mass_drift"] <= 1e-12 and d["weak"]["self_check"]["lines_identical"] is True
    assert set(d["roofline"]["per_kernel_ms"]) == {"sweep_x", "sweep_y"} and d["roofline"]["launches_timed"] == 6


def test_a_hanging_transport_candidate_leaves_the_insurance_line():
    """A transport that hangs in its self-check (a collective that never completes on a machine nobody has run before) must
    not cost the line: the workload was timed over the host-synchronised protocol first, the watchdog prints that — downgraded,
    with the phase it gave up in — and the status says failure."""
    d = run_line([sys.executable, BENCH, "--cells", "1024", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--timeout", "25"],
                 env={"ARMON_BENCH_FORCE_DIST": "1", "ARMON_BENCH_HANG_CANDIDATE": "1"}, rc=4)
    assert d["value"] > 0 and d["config"]["halo_exchange_downgraded"] is True and "INSURANCE" in d["config"]["halo_exchange"]
    assert "self-check of the transport" in d["config"]["halo_exchange_error"] and "Sod 1024x1024" in d["config"]["workload"]
