"""ctypes wrapper of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this module.
The product package (``armon.jl_amd``) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

TESTS = {"Sod": 0, "Sod_y": 1, "Sod_circ": 2, "Bizarrium": 3, "Sedov": 4, "DebugIndexes": 5}
LIMITERS = {"no_limiter": 0, "minmod": 1, "superbee": 2}
SCHEMES = {"Godunov": 0, "GAD": 1}
PROJECTIONS = {"euler": 0, "euler_2nd": 1}
SPLITTINGS = {"Sequential": 0, "Godunov": 1, "SequentialSym": 1, "Strang": 2, "X_only": 3, "Y_only": 4}

FIELDS = ("x", "y", "rho", "u", "v", "E", "p", "c", "g", "us", "ps",
          "work_1", "work_2", "work_3", "work_4", "mask")

# ref src/tests.jl:32-44
DEFAULTS = {
    "Sod": dict(domain=(1., 1.), origin=(0., 0.), cfl=0.95, maxtime=0.20),
    "Sod_y": dict(domain=(1., 1.), origin=(0., 0.), cfl=0.95, maxtime=0.20),
    "Sod_circ": dict(domain=(1., 1.), origin=(0., 0.), cfl=0.95, maxtime=0.20),
    "Bizarrium": dict(domain=(1., 1.), origin=(0., 0.), cfl=0.6, maxtime=80e-6),
    "Sedov": dict(domain=(2., 2.), origin=(-1., -1.), cfl=0.7, maxtime=1.0),
    "DebugIndexes": dict(domain=(1., 1.), origin=(0., 0.), cfl=0., maxtime=0.),
}


class Range(C.Structure):
    _fields_ = [("col_start", C.c_int64), ("col_step", C.c_int64), ("col_len", C.c_int64),
                ("row_start", C.c_int64), ("row_len", C.c_int64)]


class BlockData(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in FIELDS]


def _run_struct(real):
    class _Run(C.Structure):
        _fields_ = [("test", C.c_int32), ("scheme", C.c_int32), ("limiter", C.c_int32),
                    ("projection", C.c_int32), ("splitting", C.c_int32), ("nghost", C.c_int32),
                    ("nx", C.c_int64), ("ny", C.c_int64),
                    ("domain_size", real * 2), ("origin", real * 2),
                    ("cfl", real), ("maxtime", real), ("maxcycle", C.c_int64),
                    ("cst_dt", C.c_int32), ("Dt", real),
                    ("final_time", real), ("last_dt", real), ("cycles", C.c_int64),
                    ("solve_seconds", C.c_double),
                    ("initial_mass", real), ("initial_energy", real),
                    ("final_mass", real), ("final_energy", real),
                    ("status", C.c_int32), ("periodic", C.c_int32 * 2)]
    return _Run


Run = _run_struct(C.c_double)
Run32 = _run_struct(C.c_float)


def build(native=False, force=False, f32=False):
    """Compile the oracle with gcc (seconds). ``native`` → -march=native into its own file; ``f32`` → the
    fp32 build of the same source (libarmon_oracle_f32.so)."""
    if os.environ.get("ARMON_ORACLE_LIB") and not f32:      # e.g. the sanitizer build (oracle/Makefile)
        return os.path.abspath(os.environ["ARMON_ORACLE_LIB"])
    out = "libarmon_oracle_native.so" if native else "libarmon_oracle.so"
    if f32:
        out = "libarmon_oracle_f32.so"
    path = os.path.join(_HERE, out)
    src = os.path.join(_HERE, "armon_oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        args = ["make", "-C", _HERE, "-B", out if f32 else f"OUT={out}"]
        if f32:
            args = ["make", "-C", _HERE, "-B", "libarmon_oracle_f32.so"]
        elif native:
            args.append("ARCH=native")
        subprocess.run(args, check=True, capture_output=True)
    return path


_libs = {}


def lib(native=False, f32=False):
    key = (native, f32)
    if key not in _libs:
        L = C.CDLL(build(native, f32=f32))
        dp, i64, ci = C.c_void_p, C.c_int64, C.c_int
        dbl = C.c_float if f32 else C.c_double
        RunT = Run32 if f32 else Run
        L.armon_oracle_set_threads.argtypes = [ci]
        L.armon_oracle_get_threads.restype = ci
        L.armon_oracle_perfect_gas_EOS.argtypes = [Range, dbl] + [dp] * 7
        L.armon_oracle_bizarrium_EOS.argtypes = [Range] + [dp] * 7
        L.armon_oracle_acoustic.argtypes = [Range, i64] + [dp] * 6
        L.armon_oracle_acoustic_GAD.argtypes = [Range, i64, dbl, dbl] + [dp] * 6 + [ci]
        L.armon_oracle_cell_update.argtypes = [Range, i64, dbl, dbl] + [dp] * 5
        L.armon_oracle_advection_first_order.argtypes = [Range, i64, dbl] + [dp] * 9
        L.armon_oracle_advection_second_order.argtypes = [Range, i64, dbl, dbl] + [dp] * 9
        L.armon_oracle_euler_projection.argtypes = [Range, i64, dbl, dbl] + [dp] * 9
        L.armon_oracle_boundary_conditions.argtypes = [Range, i64, ci, dbl, dbl] + [dp] * 7
        L.armon_oracle_pack_to_array.argtypes = [Range, ci, i64, dp, ci, C.POINTER(dp)]
        L.armon_oracle_unpack_from_array.argtypes = [Range, ci, i64, dp, ci, C.POINTER(dp)]
        L.armon_oracle_dtCFL.argtypes = [Range, dbl, dbl] + [dp] * 3
        L.armon_oracle_dtCFL.restype = dbl
        L.armon_oracle_conservation_vars.argtypes = [Range, dbl, dp, dp, C.POINTER(dbl * 2)]
        L.armon_oracle_init_test.argtypes = [Range, ci, i64, i64, ci, C.POINTER(i64 * 2),
                                             C.POINTER(i64 * 2), C.POINTER(dbl * 2),
                                             C.POINTER(dbl * 2), dbl, C.POINTER(BlockData)]
        L.armon_oracle_solve.argtypes = [C.POINTER(RunT), C.POINTER(BlockData), ci]
        L.armon_oracle_solve.restype = ci
        _libs[key] = L
    return _libs[key]


def ptr(a):
    assert a.dtype in (np.float64, np.float32) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def domain_range(nx, ny, g, bl=(0, 0), tr=(0, 0)):
    """block_domain_range (ref src/blocking/blocking.jl:71-85), 0-based."""
    row = nx + 2 * g
    fx, fy = bl[0] + 1, bl[1] + 1
    lx, ly = tr[0] + nx, tr[1] + ny
    return Range((fy + g - 1) * row, row, ly - fy + 1, fx + g - 1, lx - fx + 1)


def alloc_fields(nx, ny, g, fill=0.0, dtype=np.float64):
    n = (nx + 2 * g) * (ny + 2 * g)
    return {f: np.full(n, fill, dtype=dtype) for f in FIELDS}


def block_data(fields):
    bd = BlockData()
    for f in FIELDS:
        setattr(bd, f, fields[f].ctypes.data)
    return bd


def solve(test="Sod", N=(100, 100), scheme="GAD", riemann_limiter="minmod", projection="euler_2nd",
          axis_splitting="Sequential", nghost=4, cfl=0., maxtime=0., maxcycle=500_000,
          cst_dt=False, Dt=0., domain_size=None, origin=None, threads=1, native=False,
          fields=None, skip_init=False, data_type=np.float64, periodic=(False, False)):
    """Run the oracle's armon(): returns (Run, fields dict). Option names follow ArmonParameters. ``periodic``: test aid (not
    in the reference): ghosts of that axis from the opposite border instead of the mirror."""
    f32 = np.dtype(data_type) == np.float32
    L = lib(native, f32=f32)
    L.armon_oracle_set_threads(threads)
    nx, ny = N
    d = DEFAULTS[test]
    run = Run32() if f32 else Run()
    run.test, run.scheme, run.limiter = TESTS[test], SCHEMES[scheme], LIMITERS[riemann_limiter]
    run.projection, run.splitting, run.nghost = PROJECTIONS[projection], SPLITTINGS[axis_splitting], nghost
    run.nx, run.ny = nx, ny
    ds = domain_size or d["domain"]
    og = origin or d["origin"]
    run.domain_size[0], run.domain_size[1] = ds
    run.origin[0], run.origin[1] = og
    run.cfl = cfl if cfl != 0 else d["cfl"]
    run.maxtime = maxtime if maxtime != 0 else d["maxtime"]
    run.maxcycle = maxcycle
    run.cst_dt, run.Dt = int(cst_dt), Dt
    run.periodic[0], run.periodic[1] = int(bool(periodic[0])), int(bool(periodic[1]))
    if fields is None:
        fields = alloc_fields(nx, ny, nghost, dtype=np.float32 if f32 else np.float64)
    bd = block_data(fields)
    L.armon_oracle_solve(C.byref(run), C.byref(bd), int(skip_init))
    return run, fields


def real_view(a, nx, ny, g):
    """(ny, nx) view of the real cells of a flat ghosted array."""
    return a.reshape(ny + 2 * g, nx + 2 * g)[g:g + ny, g:g + nx]
