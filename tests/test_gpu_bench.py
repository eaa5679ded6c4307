"""bench.py's line is a contract with the driver (one JSON line on stdout: metric, value, roofline, cpu_baseline at N = 1;
at N > 1 the weak workload as `value` and the strong one next to it). Small grids, seconds each."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_line(cmd, env=None, timeout=600):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env={**os.environ, **(env or {})}, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE line, whatever the libraries print while they start
    return json.loads(lines[0])


def test_single_gpu_line_carries_the_contract():
    d = run_line([sys.executable, BENCH, "--cells", "1024", "--steps", "3", "--warmup", "1", "--no-measure-traffic"])
    assert d["metric"].startswith("Mcells/sec per sweep") and d["unit"] == "Mcells/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "Sod 1024x1024" in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - 1024 * 1024 * 2 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and ro["bytes_per_cell"] == 64
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and set(ro["per_kernel_ms"]) == {"sweep_x", "sweep_y"}
    assert "traffic" in ro and "strong" not in d
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
    if "live measurement failed" in (ro["traffic_source"] or ""):
        pytest.skip("counter passes not available on this box: " + ro["traffic_source"])     # the line says so itself; not a code error
    assert ro["traffic_source"].startswith("measured for this line"), ro["traffic_source"]
    algorithmic = 64 * 2048 * 2048
    assert 0.98 * algorithmic <= ro["traffic"] <= 1.25 * algorithmic, ro["traffic"] / algorithmic


def test_two_ranks_on_one_gpu_report_the_weak_and_the_strong_workload(tmp_path):
    """The N > 1 line, rehearsed with every rank on this GPU over gloo (code path only)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--standalone", "--local-addr", "127.0.0.1", BENCH, "--gpus", "2", "--cells", "512", "--steps", "3", "--warmup", "1"]
    d = run_line(cmd, env={"ARMON_BENCH_REHEARSAL": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "2x1 tiles of 512x512" in d["config"]["workload"]
    s = d["strong"]
    assert s["value"] > 0 and "split over 2x1 tiles of 256x512" in s["workload"]
    assert abs(s["efficiency_vs_ideal"] - s["value"] / d["value"]) < 1e-3
    assert s["self_check"]["mass_drift"] <= 1e-12 and s["self_check"]["lines_identical"] is True
