"""examples/native_cycle.c drives the hot path from plain C through include/armon_hip.h alone: the header must be
valid C99 and C++11 (no GPU needed), and on a GPU the C driver must reproduce the Python host's run."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("compiler,std", [("gcc", "-std=c99"), ("g++", "-std=c++11")])
def test_header_is_plain_c_and_cxx(tmp_path, compiler, std):
    src = tmp_path / ("t.c" if compiler == "gcc" else "t.cpp")
    src.write_text('#include "armon_hip.h"\nint main(void) { armon_sweep_desc d; armon_range r; armon_block_data b;'
                   ' (void)d; (void)r; (void)b; return armon_hip_flt_size() == 8 ? 0 : 1; }\n')
    subprocess.run([compiler, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    "-fsyntax-only", str(src)], check=True)


def build_example(tmp_path, name="native_cycle"):
    exe = str(tmp_path / name)
    subprocess.run(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra", "-Werror",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"), "-o", exe,
                    "-L", os.path.join(ROOT, "armon.jl_amd"), "-larmon_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "armon.jl_amd"), "-lm"], check=True)
    return exe


def test_native_example_builds(tmp_path):
    build_example(tmp_path)
    build_example(tmp_path, "native_tiles")
    build_example(tmp_path, "native_graph")


@pytest.mark.gpu
def test_native_example_matches_the_python_host(tmp_path):
    import armon_amd
    from armon_amd.solver import BlockGrid, conservation_vars, init_test, time_loop
    n, cycles = 640, 40
    out = subprocess.run([build_example(tmp_path), str(n), str(cycles)], check=True, capture_output=True, text=True).stdout
    m = re.search(r"mass (\S+) -> (\S+), energy (\S+) -> (\S+)", out)
    assert m, out
    m0, m1, e0, e1 = (float(x.rstrip(",")) for x in m.groups())
    params = armon_amd.ArmonParameters(test="Sod", N=(n, n), maxcycle=cycles, maxtime=1e9, silent=5)
    grid = BlockGrid(params)
    init_test(params, grid)
    pm0, pe0 = conservation_vars(params, grid)
    time_loop(params, grid)
    pm1, pe1 = conservation_vars(params, grid)
    assert (m0, e0) == (pm0, pe0)
    assert (m1, e1) == (pm1, pe1)          # same kernels, same dt rule: the same bits


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["calls", "cycle"])
@pytest.mark.parametrize("n,px,py", [(640, 2, 2), (517, 3, 1), (400, 1, 2)])
def test_native_tiles_example_matches_the_python_tile_group(tmp_path, n, px, py, mode):
    """examples/native_tiles.c — the multi-GPU entry points (armon_hip_mgpu_init, halo_exchange_start/finish,
    dt_allreduce) driven from plain C, every tile on device 0 — ends on the same global mass and energy, bit for bit, as
    multi_tile.TileGroup on the same tile grid (same kernels, same exchange, same dt rule, same order of the sums).
    mode "cycle": the same run with every cycle enqueued by ONE armon_hip_mgpu_cycle call."""
    from armon_amd.multi_tile import TileGroup
    cycles = 30
    out = subprocess.run([build_example(tmp_path, "native_tiles"), str(n), str(px), str(py), str(cycles), mode], check=True,
                         capture_output=True, text=True).stdout
    m = re.search(r"mass (\S+) -> (\S+), energy (\S+) -> (\S+)", out)
    assert m, out
    m0, m1, e0, e1 = (float(x.rstrip(",")) for x in m.groups())
    group = TileGroup((px, py), test="Sod", N=(n, n), maxcycle=cycles, maxtime=1e9, silent=5)
    try:
        group.init_test()
        pm0, pe0 = group.conservation_vars()
        group.time_loop()
        pm1, pe1 = group.conservation_vars()
    finally:
        group.close()
    assert (m0, e0) == (pm0, pe0)
    assert (m1, e1) == (pm1, pe1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,cycles", [(640, 40), (129, 7), (1000, 2)])
def test_native_graph_example_replays_the_host_driven_run(tmp_path, n, cycles):
    """examples/native_graph.c — armon_dt_state, auto_step and armon_hip_graph_* from plain C: the run whose time step never
    leaves the device, one captured cycle replayed per call (three replays past the end included), must end on the bits of
    the host-driven loop: time, next dt, cycle count, mass and energy."""
    res = subprocess.run([build_example(tmp_path, "native_graph"), str(n), str(cycles)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "identical: yes" in res.stdout, res.stdout
    assert re.search(r"cycle %d, done 1" % cycles, res.stdout), res.stdout
