"""Seeded random sweep over tile decompositions: a TileGroup (the library's multi-GPU entry points, every tile on device 0 —
pack, copy on the transfer stream, unpack, interior while the faces travel, boundary strips, the group's dt reduction) must
give the single block's bits whatever the process grid, the (uneven) tile sizes, the ghost width, the options, the precision
and the arithmetic. tests/test_gpu_distributed.py lists its layouts by hand; this one draws them."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ("rho", "u", "v", "E", "p")


def draw_cases(seed, count):
    rng = random.Random(seed)
    cases = []
    for _ in range(count):
        scheme = rng.choice(["GAD", "GAD", "Godunov"])
        projection = rng.choice(["euler_2nd", "euler_2nd", "euler"])
        lag = 2 + (scheme == "GAD") + (projection == "euler_2nd")
        nghost = max(lag, rng.choice([lag, 4, 5, 6]))
        P = rng.choice([(2, 1), (1, 2), (2, 2), (3, 1), (1, 3), (3, 2), (2, 3), (4, 2), (3, 3)])
        # every tile needs at least `nghost` cells along each axis (its faces are read from real cells)
        nx = rng.randint(P[0] * max(nghost, 5), P[0] * 70)
        ny = rng.randint(P[1] * max(nghost, 5), P[1] * 60)
        cases.append(dict(P=P, test=rng.choice(["Sod_circ", "Sod_circ", "Sedov", "Bizarrium", "Sod", "Sod_y"]), N=(nx, ny),
                          scheme=scheme, projection=projection, riemann_limiter=rng.choice(["minmod", "superbee", "no_limiter"]),
                          axis_splitting=rng.choice(["Sequential", "Sequential", "Godunov", "Strang"]), nghost=nghost,
                          maxcycle=rng.randint(3, 6), data_type=rng.choice(["float64", "float64", "float32"]),
                          exact_arithmetic=rng.random() < 0.5, use_fused_sweep=rng.random() < 0.8,
                          overlap_halo=rng.random() < 0.8, edge_stream=rng.random() < 0.7))
    return cases


CASES = draw_cases(int(os.environ.get("ARMON_RANDOM_SEED", "20261005")), int(os.environ.get("ARMON_RANDOM_CASES", "24")))


def case_id(c):
    return (f"{c['P'][0]}x{c['P'][1]}-{c['test']}-{c['N'][0]}x{c['N'][1]}-g{c['nghost']}-{c['scheme']}-{c['projection']}-"
            f"{c['axis_splitting']}-{c['data_type']}-{'exact' if c['exact_arithmetic'] else 'tuned'}-"
            f"{'fused' if c['use_fused_sweep'] else 'staged'}")


@pytest.mark.parametrize("case", CASES, ids=case_id)
def test_random_tile_group_equals_the_single_block(case):
    import armon_amd
    from armon_amd.multi_tile import TileGroup
    opts = dict(case)
    P = opts.pop("P")
    group_only = {k: opts.pop(k) for k in ("overlap_halo", "edge_stream")}
    if not opts["use_fused_sweep"]:
        opts["exact_arithmetic"] = True                     # the staged kernels have one arithmetic
    ref = armon_amd.armon(armon_amd.ArmonParameters(silent=5, return_data=True, **opts))
    host = ref.data.device_to_host(NAMES)
    full = {k: ref.data.real_view(v) for k, v in host.items()}
    group = TileGroup(P, silent=5, **opts, **group_only)
    try:
        stats = group.run()
        assert stats.cycles == ref.cycles and stats.last_dt == ref.last_dt and stats.final_time == ref.final_time
        got = group.gather()
        for k in NAMES:
            assert np.array_equal(got[k], full[k]), f"{k}: {int((got[k] != full[k]).sum())} cells differ"
    finally:
        group.close()
