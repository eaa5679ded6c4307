"""GPU parity of every staged HIP kernel against the CPU oracle, through the C ABI.

Bar: BIT-EXACT (np.array_equal on the fp64 bit patterns) — the kernels are built with
-ffp-contract=off and evaluate the reference formulas in the reference's order, as the oracle does.
Only the conservation sums (order of additions differs) get a tolerance: 1e-13 relative.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = 4
SHAPES = [(37, 29), (128, 3), (5, 70), (257, 64)]


@pytest.fixture(scope="module")
def dev():
    import armon_amd
    from armon_amd.device import HIPDevice
    d = HIPDevice(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def L():
    import armon_amd
    return armon_amd.lib()


def rand_state(nx, ny, seed):
    """Physically plausible random fields over the whole ghosted block."""
    rng = np.random.default_rng(seed)
    n = (nx + 2 * G) * (ny + 2 * G)
    f = {
        "rho": rng.uniform(0.1, 2.0, n), "u": rng.uniform(-1, 1, n), "v": rng.uniform(-1, 1, n),
        "E": rng.uniform(2.0, 4.0, n), "p": rng.uniform(0.1, 2.0, n), "c": rng.uniform(0.5, 2.0, n),
        "g": rng.uniform(1, 2, n), "us": rng.uniform(-1, 1, n), "ps": rng.uniform(0.1, 2.0, n),
        "work_1": rng.uniform(-1, 1, n), "work_2": rng.uniform(-1, 1, n),
        "work_3": rng.uniform(-1, 1, n), "work_4": rng.uniform(-1, 1, n),
    }
    return f


def upload(dev, f):
    return {k: dev.from_host(a) for k, a in f.items()}


def P(d, k):
    return C.c_void_p(d[k].ptr)


def ranges_for(oracle, nx, ny, axis, w=2):
    dr = oracle.domain_range
    if axis == 0:
        return dict(s=1, fluxes=dr(nx, ny, G, (-w, 0), (w + 1, 0)), cell_update=dr(nx, ny, G, (-w, 0), (w, 0)),
                    advection=dr(nx, ny, G, (0, 0), (1, 0)), real=dr(nx, ny, G))
    return dict(s=nx + 2 * G, fluxes=dr(nx, ny, G, (0, -w), (0, w + 1)), cell_update=dr(nx, ny, G, (0, -w), (0, w)),
                advection=dr(nx, ny, G, (0, 0), (0, 1)), real=dr(nx, ny, G))


def conv(r):
    """oracle.Range → armon_amd Range (same layout)."""
    from armon_amd._lib import Range
    return Range(r.col_start, r.col_step, r.col_len, r.row_start, r.row_len)


def assert_same(dev_arrays, host, names):
    for k in names:
        got = dev_arrays[k].to_host()
        assert np.array_equal(got, host[k]), f"{k}: {np.abs(got - host[k]).max()} max abs diff"


def test_sanity_exports(L):
    assert L.armon_hip_flt_size() == 8 and L.armon_hip_idx_size() == 8
    n = C.c_int()
    assert L.armon_hip_device_count(C.byref(n)) == 0 and n.value >= 1


def test_device_array_roundtrip_and_meminfo(dev):
    a = np.arange(1000, dtype=np.float64)
    d = dev.from_host(a)
    assert np.array_equal(d.to_host(), a)
    free, total = dev.memory_info()
    assert 0 < free <= total and total > 200e9
    assert "gfx950" in dev.name


def test_stream_copy4(dev):
    """The measurement aid bench.py uses as the same-device streaming ceiling really copies 4 arrays into 4."""
    rng = np.random.default_rng(3)
    src = [rng.random(4098) for _ in range(4)]
    d_src = [dev.from_host(a) for a in src]
    d_dst = [dev.zeros(4098) for _ in range(4)]
    import armon_amd
    for nt in (0, 1, 2, 3):                          # plain / non-temporal loads / stores / both (ARMON_COPY_NT)
        for d in d_dst:
            d.fill_bytes(0) if hasattr(d, "fill_bytes") else d.copy_from_host(np.zeros(4098))
        assert armon_amd.lib().armon_hip_set_tuning(dev.ctx, b"ARMON_COPY_NT", nt) == 0 and dev.get_tuning("ARMON_COPY_NT") == nt
        dev.stream_copy4(d_src, d_dst, src[0].nbytes)
        for a, d in zip(src, d_dst):
            assert np.array_equal(d.to_host(), a), nt
    with pytest.raises(Exception):
        dev.stream_copy4(d_src, d_dst, 24)          # not a multiple of 16


@pytest.mark.parametrize("shape", SHAPES)
def test_perfect_gas_EOS(dev, L, oracle, shape):
    nx, ny = shape
    f = rand_state(nx, ny, 1)
    d = upload(dev, f)
    r = oracle.domain_range(nx, ny, G)
    oracle.lib().armon_oracle_perfect_gas_EOS(r, 1.4, *(oracle.ptr(f[k]) for k in ("rho", "E", "u", "v", "p", "c", "g")))
    assert L.armon_hip_perfect_gas_EOS(dev.ctx, conv(r), 1.4, *(P(d, k) for k in ("rho", "E", "u", "v", "p", "c", "g"))) == 0
    assert_same(d, f, ("p", "c", "g"))


@pytest.mark.parametrize("shape", SHAPES[:2])
def test_bizarrium_EOS(dev, L, oracle, shape):
    nx, ny = shape
    rng = np.random.default_rng(2)
    n = (nx + 2 * G) * (ny + 2 * G)
    f = rand_state(nx, ny, 2)
    f["rho"] = rng.uniform(0.9e4, 1.5e4, n)
    f["u"] = rng.uniform(-300, 300, n)
    f["v"] = rng.uniform(-300, 300, n)
    f["E"] = rng.uniform(1e6, 5e6, n)
    d = upload(dev, f)
    r = oracle.domain_range(nx, ny, G)
    oracle.lib().armon_oracle_bizarrium_EOS(r, *(oracle.ptr(f[k]) for k in ("rho", "u", "v", "E", "p", "c", "g")))
    assert L.armon_hip_bizarrium_EOS(dev.ctx, conv(r), *(P(d, k) for k in ("rho", "u", "v", "E", "p", "c", "g"))) == 0
    assert_same(d, f, ("p", "c", "g"))


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_acoustic(dev, L, oracle, shape, axis):
    nx, ny = shape
    f = rand_state(nx, ny, 3)
    d = upload(dev, f)
    R = ranges_for(oracle, nx, ny, axis)
    ua = "u" if axis == 0 else "v"
    oracle.lib().armon_oracle_acoustic(R["fluxes"], R["s"], *(oracle.ptr(f[k]) for k in ("us", "ps", "rho", ua, "p", "c")))
    assert L.armon_hip_acoustic(dev.ctx, conv(R["fluxes"]), R["s"], *(P(d, k) for k in ("us", "ps", "rho", ua, "p", "c"))) == 0
    assert_same(d, f, ("us", "ps"))


@pytest.mark.parametrize("limiter", [0, 1, 2])
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("shape", SHAPES[:2])
def test_acoustic_GAD(dev, L, oracle, shape, axis, limiter):
    nx, ny = shape
    f = rand_state(nx, ny, 4)
    d = upload(dev, f)
    R = ranges_for(oracle, nx, ny, axis)
    ua = "u" if axis == 0 else "v"
    dt, dx = 1e-3, 1.0 / nx
    oracle.lib().armon_oracle_acoustic_GAD(R["fluxes"], R["s"], dt, dx,
                                           *(oracle.ptr(f[k]) for k in ("us", "ps", "rho", ua, "p", "c")), limiter)
    assert L.armon_hip_acoustic_GAD(dev.ctx, conv(R["fluxes"]), R["s"], dt, dx,
                                    *(P(d, k) for k in ("us", "ps", "rho", ua, "p", "c")), limiter) == 0
    assert_same(d, f, ("us", "ps"))


@pytest.mark.parametrize("limiter", [0, 1])
@pytest.mark.parametrize("shift", range(8))
@pytest.mark.parametrize("shape", [(250, 9), (1006, 5), (121, 6)])
def test_acoustic_GAD_x_strips(dev, L, oracle, shape, shift, limiter):
    """The two-cells-per-lane x form (even row pitch, fp64): several strips of 120 fluxes per row, rows that are not a
    multiple of the 4 of a workgroup, every position of the first flux within its 64-B sector, short last strips."""
    nx, ny = shape
    f = rand_state(nx, ny, 40 + shift)
    d = upload(dev, f)
    r = oracle.domain_range(nx, ny, G, (-2 + shift, 0), (3 - (shift % 3), 0))
    dt, dx = 1e-3, 1.0 / nx
    oracle.lib().armon_oracle_acoustic_GAD(r, 1, dt, dx, *(oracle.ptr(f[k]) for k in ("us", "ps", "rho", "u", "p", "c")), limiter)
    assert L.armon_hip_acoustic_GAD(dev.ctx, conv(r), 1, dt, dx, *(P(d, k) for k in ("us", "ps", "rho", "u", "p", "c")), limiter) == 0
    assert_same(d, f, ("us", "ps"))


def test_acoustic_GAD_x_unaligned_arrays(dev, L, oracle):
    """Arrays that do not start on 16 bytes (a view one cell into an allocation) take the one-cell-per-lane form: same bits."""
    nx, ny = 250, 9
    n = (nx + 2 * G) * (ny + 2 * G)
    f = rand_state(nx, ny, 51)
    r = oracle.domain_range(nx, ny, G, (-2, 0), (3, 0))
    dt, dx = 1e-3, 1.0 / nx
    big = {k: dev.from_host(np.concatenate([[0.0], f[k]])) for k in ("us", "ps", "rho", "u", "p", "c")}
    oracle.lib().armon_oracle_acoustic_GAD(r, 1, dt, dx, *(oracle.ptr(f[k]) for k in ("us", "ps", "rho", "u", "p", "c")), 1)
    assert L.armon_hip_acoustic_GAD(dev.ctx, conv(r), 1, dt, dx,
                                    *(C.c_void_p(big[k].ptr + 8) for k in ("us", "ps", "rho", "u", "p", "c")), 1) == 0
    for k in ("us", "ps"):
        assert np.array_equal(big[k].to_host()[1:], f[k])


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_cell_update(dev, L, oracle, shape, axis):
    nx, ny = shape
    f = rand_state(nx, ny, 5)
    d = upload(dev, f)
    R = ranges_for(oracle, nx, ny, axis)
    ua = "u" if axis == 0 else "v"
    dt, dx = 1e-3, 1.0 / nx
    oracle.lib().armon_oracle_cell_update(R["cell_update"], R["s"], dx, dt,
                                          *(oracle.ptr(f[k]) for k in ("us", "ps", "rho", ua, "E")))
    assert L.armon_hip_cell_update(dev.ctx, conv(R["cell_update"]), R["s"], dx, dt,
                                   *(P(d, k) for k in ("us", "ps", "rho", ua, "E"))) == 0
    assert_same(d, f, ("rho", "u", "v", "E"))


ADV = ("us", "rho", "u", "v", "E", "work_1", "work_2", "work_3", "work_4")


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_advection(dev, L, oracle, shape, axis, order):
    nx, ny = shape
    f = rand_state(nx, ny, 6)
    d = upload(dev, f)
    R = ranges_for(oracle, nx, ny, axis)
    dt, dx = 1e-3, 1.0 / nx
    if order == 1:
        oracle.lib().armon_oracle_advection_first_order(R["advection"], R["s"], dt, *(oracle.ptr(f[k]) for k in ADV))
        assert L.armon_hip_advection_first_order(dev.ctx, conv(R["advection"]), R["s"], dt, *(P(d, k) for k in ADV)) == 0
    else:
        oracle.lib().armon_oracle_advection_second_order(R["advection"], R["s"], dx, dt, *(oracle.ptr(f[k]) for k in ADV))
        assert L.armon_hip_advection_second_order(dev.ctx, conv(R["advection"]), R["s"], dx, dt, *(P(d, k) for k in ADV)) == 0
    assert_same(d, f, ("work_1", "work_2", "work_3", "work_4"))


def _random_flux_cases(seed, count):
    import random
    rng = random.Random(seed)
    cases = []
    for _ in range(count):
        nx, ny = rng.randint(1, 600), rng.randint(1, 90)
        # a range inside the block whose stencil (2 cells below, 2 above along the sweep) stays inside the ghosted arrays
        lo = rng.randint(-2, min(3, nx - 1))
        hi = rng.randint(-min(3, nx - 1 - max(lo, 0)), 2)
        cases.append((nx, ny, lo, hi, rng.randint(0, 1), rng.randint(0, 2), rng.randint(0, 10 ** 6)))
    return cases


@pytest.mark.parametrize("nx,ny,lo,hi,axis,limiter,seed", _random_flux_cases(4242, 40))
def test_flux_kernels_on_random_ranges(dev, L, oracle, nx, ny, lo, hi, axis, limiter, seed):
    """acoustic_GAD and advection_second_order on drawn shapes and ranges (first / last cell anywhere within the ghost
    stencil, along x and along y): the wave layouts of the x forms (56 results + 4 + 4 halo lanes, placed on the sectors of
    the array) and the marches of the y forms must write the oracle's bits inside the range and nothing outside it."""
    if axis == 1:
        nx, ny = ny, nx
    f = rand_state(nx, ny, seed)
    d = upload(dev, f)
    bl, tr = ((lo, 0), (hi, 0)) if axis == 0 else ((0, lo), (0, hi))
    r = oracle.domain_range(nx, ny, G, bl, tr)
    s = 1 if axis == 0 else nx + 2 * G
    ua = "u" if axis == 0 else "v"
    dt, dx = 1e-3, 1.0 / max(nx, ny)
    oracle.lib().armon_oracle_acoustic_GAD(r, s, dt, dx, *(oracle.ptr(f[k]) for k in ("us", "ps", "rho", ua, "p", "c")), limiter)
    assert L.armon_hip_acoustic_GAD(dev.ctx, conv(r), s, dt, dx, *(P(d, k) for k in ("us", "ps", "rho", ua, "p", "c")), limiter) == 0
    assert_same(d, f, ("us", "ps"))
    oracle.lib().armon_oracle_advection_second_order(r, s, dx, dt, *(oracle.ptr(f[k]) for k in ADV))
    assert L.armon_hip_advection_second_order(dev.ctx, conv(r), s, dx, dt, *(P(d, k) for k in ADV)) == 0
    assert_same(d, f, ("work_1", "work_2", "work_3", "work_4"))


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_euler_projection(dev, L, oracle, shape, axis):
    nx, ny = shape
    f = rand_state(nx, ny, 7)
    d = upload(dev, f)
    R = ranges_for(oracle, nx, ny, axis)
    dt, dx = 1e-3, 1.0 / nx
    oracle.lib().armon_oracle_euler_projection(R["real"], R["s"], dx, dt, *(oracle.ptr(f[k]) for k in ADV))
    assert L.armon_hip_euler_projection(dev.ctx, conv(R["real"]), R["s"], dx, dt, *(P(d, k) for k in ADV)) == 0
    assert_same(d, f, ("rho", "u", "v", "E"))


BCV = ("rho", "u", "v", "p", "c", "g", "E")


@pytest.mark.parametrize("side", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", SHAPES[:3])
def test_boundary_conditions(dev, L, oracle, shape, side):
    from armon_amd.blocking import BlockSize, Side, axis_of
    nx, ny = shape
    bs = BlockSize((nx + 2 * G, ny + 2 * G), G)
    sd = Side(side)
    f = rand_state(nx, ny, 8)
    d = upload(dev, f)
    r = bs.border_domain(sd).to_c()
    incr = bs.stride_along(axis_of(sd)) * (-1 if sd in (Side.Left, Side.Bottom) else 1)
    uf, vf = (-1., 1.) if sd in (Side.Left, Side.Right) else (1., -1.)
    orr = oracle.Range(r.col_start, r.col_step, r.col_len, r.row_start, r.row_len)
    oracle.lib().armon_oracle_boundary_conditions(orr, incr, G, uf, vf, *(oracle.ptr(f[k]) for k in BCV))
    assert L.armon_hip_boundary_conditions(dev.ctx, r, incr, G, uf, vf, *(P(d, k) for k in BCV)) == 0
    assert_same(d, f, BCV)


@pytest.mark.parametrize("side", [1, 2, 3, 4])
def test_pack_unpack_roundtrip_and_layout(dev, L, oracle, side):
    """ref src/halo_exchange.jl:187-216; the index-encoding idea of ref test/mpi.jl:272-360."""
    from armon_amd.blocking import BlockSize, Side
    nx, ny = 21, 13
    bs = BlockSize((nx + 2 * G, ny + 2 * G), G)
    sd = Side(side)
    n = bs.n_cells
    names = ("rho", "u", "v", "E", "p", "c", "g")
    # every var holds (var_index*1e6 + cell index): the packed buffer tells where each value came from
    f = {k: (np.arange(n, dtype=np.float64) + 1e6 * vi) for vi, k in enumerate(names)}
    d = upload(dev, f)
    face = bs.real_face_size(sd)
    send = bs.border_domain(sd, single_strip=False).to_c()
    recv = bs.ghost_domain(sd, single_strip=False).to_c()
    buf_h = np.zeros(face * G * 7)
    buf_d = dev.zeros(face * G * 7)
    vars_h = (C.c_void_p * 7)(*(f[k].ctypes.data for k in names))
    vars_d = (C.c_void_p * 7)(*(d[k].ptr for k in names))
    osend = oracle.Range(send.col_start, send.col_step, send.col_len, send.row_start, send.row_len)
    orecv = oracle.Range(recv.col_start, recv.col_step, recv.col_len, recv.row_start, recv.row_len)
    oracle.lib().armon_oracle_pack_to_array(osend, G, face, oracle.ptr(buf_h), 7, vars_h)
    assert L.armon_hip_pack_to_array(dev.ctx, send, G, face, C.c_void_p(buf_d.ptr), 7, vars_d) == 0
    assert np.array_equal(buf_d.to_host(), buf_h)
    assert len(np.unique(buf_h)) == buf_h.size          # a bijection: nothing packed twice
    oracle.lib().armon_oracle_unpack_from_array(orecv, G, face, oracle.ptr(buf_h), 7, vars_h)
    assert L.armon_hip_unpack_from_array(dev.ctx, recv, G, face, C.c_void_p(buf_d.ptr), 7, vars_d) == 0
    assert_same(d, f, names)


@pytest.mark.parametrize("shape", SHAPES + [(1000, 700)])
def test_dtCFL_and_conservation(dev, L, oracle, shape):
    nx, ny = shape
    f = rand_state(nx, ny, 9)
    d = upload(dev, f)
    r = oracle.domain_range(nx, ny, G)
    dx, dy = 1.0 / nx, 1.0 / ny
    ref = oracle.lib().armon_oracle_dtCFL(r, dx, dy, *(oracle.ptr(f[k]) for k in ("u", "v", "c")))
    out = C.c_double()
    assert L.armon_hip_dtCFL(dev.ctx, conv(r), dx, dy, *(P(d, k) for k in ("u", "v", "c")), C.byref(out)) == 0
    assert out.value == ref                               # min is order independent: exact
    cons_ref = (C.c_double * 2)()
    oracle.lib().armon_oracle_conservation_vars(r, dx * dy, oracle.ptr(f["rho"]), oracle.ptr(f["E"]), C.byref(cons_ref))
    cons = (C.c_double * 2)()
    assert L.armon_hip_conservation_vars(dev.ctx, conv(r), dx * dy, P(d, "rho"), P(d, "E"), C.byref(cons)) == 0
    for a, b in zip(cons, cons_ref):
        assert abs(a - b) <= 1e-13 * abs(b)


def test_dtCFL_reports_nonfinite(dev, L, oracle):
    nx, ny = 16, 16
    f = rand_state(nx, ny, 10)
    f["u"][:] = 0.; f["v"][:] = 0.; f["c"][:] = 0.      # dx/0 → inf everywhere
    d = upload(dev, f)
    out = C.c_double()
    r = conv(oracle.domain_range(nx, ny, G))
    assert L.armon_hip_dtCFL(dev.ctx, r, 0.1, 0.1, *(P(d, k) for k in ("u", "v", "c")), C.byref(out)) == 0
    assert np.isinf(out.value)


def test_empty_range_and_bad_args(dev, L):
    from armon_amd._lib import Range
    a = dev.zeros(16)
    p = C.c_void_p(a.ptr)
    assert L.armon_hip_perfect_gas_EOS(dev.ctx, Range(0, 4, 0, 0, 4), 1.4, p, p, p, p, p, p, p) == 0   # empty: no-op
    assert L.armon_hip_perfect_gas_EOS(dev.ctx, Range(0, 4, 1, 0, 4), 1.4, None, p, p, p, p, p, p) == 1
    assert b"NULL" in L.armon_hip_last_error()
    assert L.armon_hip_acoustic_GAD(dev.ctx, Range(0, 4, 1, 0, 4), 1, 0.1, 0.1, p, p, p, p, p, p, 7) == 1
    assert b"limiter" in L.armon_hip_last_error()


@pytest.mark.parametrize("suffix", ["", "_f32"])
@pytest.mark.parametrize("empty", [dict(col_len=0, row_len=8), dict(col_len=6, row_len=0)], ids=["no-rows", "no-columns"])
def test_every_kernel_accepts_an_empty_range(dev, L, empty, suffix):
    """Empty inputs: every range-taking entry point, in both precisions, is a no-op on a range without rows or without
    columns (the reference's step ranges can be empty, e.g. `1:0` borders of a 1-cell-wide block) — it returns 0, launches
    nothing that writes, and the reductions return their neutral element."""
    from armon_amd._lib import Range, SIGNATURES
    flt = np.float32 if suffix else np.float64
    cflt = C.c_float if suffix else C.c_double
    marker = np.arange(16 * 16, dtype=flt) + 1
    a = dev.from_host(marker)
    p = C.c_void_p(a.ptr)
    r = Range(16 * 2 + 2, 16, empty["col_len"], 0, empty["row_len"])
    names = ["perfect_gas_EOS", "bizarrium_EOS", "acoustic", "acoustic_GAD", "cell_update", "advection_first_order",
             "advection_second_order", "euler_projection", "boundary_conditions", "dtCFL_async"]
    for name in names:
        _res, argtypes = SIGNATURES["armon_hip_" + name + suffix]
        args = []
        for t in argtypes[2:]:
            if t is C.c_void_p:
                args.append(p)
            elif t in (C.c_double, C.c_float):
                args.append(t(0.1))
            elif t is C.c_int64:
                args.append(C.c_int64(1))
            else:
                args.append(C.c_int(1))            # limiter tag / nghost
        assert getattr(L, "armon_hip_" + name + suffix)(dev.ctx, r, *args) == 0, (name, L.armon_hip_last_error())
    vars_ = (C.c_void_p * 7)(*[a.ptr] * 7)
    for name in ("pack_to_array", "unpack_from_array"):
        assert getattr(L, "armon_hip_" + name + suffix)(dev.ctx, r, 4, 8, p, 7, vars_) == 0, name
    out = cflt(-1.)
    assert getattr(L, "armon_hip_dtCFL" + suffix)(dev.ctx, r, cflt(0.1), cflt(0.1), p, p, p, C.byref(out)) == 0
    assert np.isinf(out.value) and out.value > 0                       # the minimum over nothing
    cons = (cflt * 2)(-1., -1.)
    assert getattr(L, "armon_hip_conservation_vars" + suffix)(dev.ctx, r, cflt(0.5), p, p, C.byref(cons)) == 0
    assert tuple(cons) == (0., 0.)                                     # the sum over nothing
    dev.wait()
    got = a.to_host()
    # dtCFL_async writes its scalar (+inf) where it is told to: element 0 of the marker array here
    assert np.isinf(got[0]) and np.array_equal(got[1:], marker[1:])


@pytest.mark.parametrize("test", ["Sod", "Sod_circ", "Bizarrium", "Sedov", "DebugIndexes"])
def test_init_test(dev, L, oracle, test):
    import armon_amd
    from armon_amd.solver import BlockGrid, init_test
    nx, ny = 45, 31
    params = armon_amd.ArmonParameters(test=test, N=(nx, ny), use_fused_sweep=False)
    grid = BlockGrid(params)
    init_test(params, grid)
    got = grid.device_to_host(names=oracle.FIELDS)
    f = oracle.alloc_fields(nx, ny, G, fill=np.nan)
    run, f = oracle.solve(test=test, N=(nx, ny), maxcycle=0, fields=f)
    for k in oracle.FIELDS:
        assert np.array_equal(got[k], f[k]), k
