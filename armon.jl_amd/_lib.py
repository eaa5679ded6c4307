"""ctypes binding of libarmon_hip.so — the same C ABI a Julia ``ccall`` shim binds (include/armon_hip.h).

There is no CPU fallback: if the library is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARMON_HIP_LIB") or os.path.join(_HERE, "libarmon_hip.so")   # env: A/B builds

FIELDS = ("x", "y", "rho", "u", "v", "E", "p", "c", "g", "us", "ps",
          "work_1", "work_2", "work_3", "work_4", "mask")


class SolverException(Exception):
    """ref src/utils.jl:102-117 — ``category`` is :config, :time, :cpp ... in the reference."""

    def __init__(self, category, msg):
        super().__init__(f"{category}: {msg}")
        self.category = category
        self.msg = msg


def solver_error(category, msg):
    raise SolverException(category, msg)


class Range(C.Structure):
    """armon_range — 0-based DomainRange (ref src/domain_ranges.jl:39-61)."""
    _fields_ = [("col_start", C.c_int64), ("col_step", C.c_int64), ("col_len", C.c_int64),
                ("row_start", C.c_int64), ("row_len", C.c_int64)]

    def __len__(self):
        return self.col_len * self.row_len

    def __repr__(self):
        return (f"Range(col={self.col_start}:{self.col_step}:x{self.col_len}, "
                f"row={self.row_start}+{self.row_len})")


class BlockDataPtrs(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in FIELDS]


class SweepDesc(C.Structure):
    _fields_ = [("axis", C.c_int32), ("scheme", C.c_int32), ("limiter", C.c_int32),
                ("projection", C.c_int32), ("eos", C.c_int32), ("nghost", C.c_int32),
                ("bc_low", C.c_int32), ("bc_high", C.c_int32), ("exact", C.c_int32),
                ("x_kernel", C.c_int32),
                ("nx", C.c_int64), ("ny", C.c_int64),
                ("dt", C.c_double), ("dx", C.c_double), ("gamma", C.c_double),
                ("u_factor_low", C.c_double), ("v_factor_low", C.c_double),
                ("u_factor_high", C.c_double), ("v_factor_high", C.c_double),
                ("rho_in", C.c_void_p), ("u_in", C.c_void_p), ("v_in", C.c_void_p), ("E_in", C.c_void_p),
                ("rho_out", C.c_void_p), ("u_out", C.c_void_p), ("v_out", C.c_void_p), ("E_out", C.c_void_p),
                ("p_out", C.c_void_p), ("c_out", C.c_void_p),
                ("dt_cfl_out", C.c_void_p), ("cfl_dx", C.c_double), ("cfl_dy", C.c_double),
                ("out_lo", C.c_int64), ("out_hi", C.c_int64), ("dt_accumulate", C.c_int32), ("reserved", C.c_int32),
                ("dt_state", C.c_void_p)]


class DtState(C.Structure):
    """armon_dt_state — GlobalTimeStep on the device (include/armon_hip.h)."""
    _fields_ = [("current_dt", C.c_double), ("time", C.c_double), ("L_prev", C.c_double), ("cycle", C.c_int64),
                ("done", C.c_int32), ("invalid", C.c_int32), ("emit_p", C.c_int32), ("pad", C.c_int32),
                ("invalid_cycle", C.c_int64), ("invalid_value", C.c_double),
                ("auto_step", C.c_int32), ("cst_dt", C.c_int32), ("maxcycle", C.c_int64),
                ("cfl", C.c_double), ("maxtime", C.c_double), ("Dt", C.c_double)]


class HaloDesc(C.Structure):
    """armon_halo_desc — what one local tile exchanges (include/armon_hip.h)."""
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nghost", C.c_int32), ("nvars", C.c_int32),
                ("vars", C.c_void_p * 8)]


class CyclePlan(C.Structure):
    """armon_cycle_plan — one solver cycle of every local tile (include/armon_hip.h, armon_hip_mgpu_cycle)."""
    _fields_ = [("n_sweeps", C.c_int32), ("emit_p", C.c_int32), ("emit_dt", C.c_int32), ("overlap", C.c_int32),
                ("axis", C.c_int32 * 4), ("dt", C.c_double * 4), ("next_axis", C.c_int32), ("event_slot", C.c_int32),
                ("event_ctx", C.c_void_p), ("dt_host", C.c_void_p), ("dt_event_slot", C.c_int32), ("reserved", C.c_int32)]


class TileCycle(C.Structure):
    """armon_tile_cycle(_f32) — the full X and Y sweep descriptors of one local tile."""
    _fields_ = [("x", SweepDesc), ("y", SweepDesc)]


MGPU_ID_BYTES = 256

_lib = None

# name → (restype, argtypes); every symbol include/armon_hip.h declares
_dp, _i64, _dbl, _ci, _vp = C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_void_p
SIGNATURES = {
    "armon_hip_flt_size": (_ci, []),
    "armon_hip_idx_size": (_ci, []),
    "armon_hip_version": (C.c_char_p, []),
    "armon_hip_last_error": (C.c_char_p, []),
    "armon_hip_device_count": (_ci, [C.POINTER(_ci)]),
    "armon_hip_init": (_ci, [_ci, _vp, C.POINTER(_vp)]),
    "armon_hip_destroy": (_ci, [_vp]),
    "armon_hip_sync": (_ci, [_vp]),
    "armon_hip_device_memory_info": (_ci, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "armon_hip_device_name": (_ci, [_vp, C.c_char_p, C.c_size_t]),
    "armon_hip_stream": (_vp, [_vp]),
    "armon_hip_set_tuning": (_ci, [_vp, C.c_char_p, _ci]),
    "armon_hip_get_tuning": (_ci, [_vp, C.c_char_p, C.POINTER(_ci)]),
    "armon_hip_malloc": (_ci, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "armon_hip_free": (_ci, [_vp, _vp]),
    "armon_hip_memcpy": (_ci, [_vp, _vp, _vp, C.c_size_t, _ci]),
    "armon_hip_memset": (_ci, [_vp, _vp, _ci, C.c_size_t]),
    "armon_hip_malloc_host": (_ci, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "armon_hip_free_host": (_ci, [_vp, _vp]),
    "armon_hip_memcpy_async": (_ci, [_vp, _vp, _vp, C.c_size_t, _ci]),
    "armon_hip_event_sync": (_ci, [_vp, _ci]),
    "armon_hip_stream_copy4": (_ci, [_vp, C.POINTER(_vp * 4), C.POINTER(_vp * 4), C.c_size_t]),
    "armon_hip_timer_start": (_ci, [_vp]),
    "armon_hip_timer_stop": (_ci, [_vp, C.POINTER(_dbl)]),
    "armon_hip_event_record": (_ci, [_vp, _ci]),
    "armon_hip_event_elapsed_ms": (_ci, [_vp, _ci, _ci, C.POINTER(_dbl)]),
    "armon_hip_perfect_gas_EOS": (_ci, [_vp, Range, _dbl] + [_dp] * 7),
    "armon_hip_bizarrium_EOS": (_ci, [_vp, Range] + [_dp] * 7),
    "armon_hip_acoustic": (_ci, [_vp, Range, _i64] + [_dp] * 6),
    "armon_hip_acoustic_GAD": (_ci, [_vp, Range, _i64, _dbl, _dbl] + [_dp] * 6 + [_ci]),
    "armon_hip_cell_update": (_ci, [_vp, Range, _i64, _dbl, _dbl] + [_dp] * 5),
    "armon_hip_advection_first_order": (_ci, [_vp, Range, _i64, _dbl] + [_dp] * 9),
    "armon_hip_advection_second_order": (_ci, [_vp, Range, _i64, _dbl, _dbl] + [_dp] * 9),
    "armon_hip_euler_projection": (_ci, [_vp, Range, _i64, _dbl, _dbl] + [_dp] * 9),
    "armon_hip_boundary_conditions": (_ci, [_vp, Range, _i64, _ci, _dbl, _dbl] + [_dp] * 7),
    "armon_hip_pack_to_array": (_ci, [_vp, Range, _ci, _i64, _dp, _ci, C.POINTER(_dp)]),
    "armon_hip_unpack_from_array": (_ci, [_vp, Range, _ci, _i64, _dp, _ci, C.POINTER(_dp)]),
    "armon_hip_dtCFL_async": (_ci, [_vp, Range, _dbl, _dbl] + [_dp] * 4),
    "armon_hip_dtCFL": (_ci, [_vp, Range, _dbl, _dbl] + [_dp] * 3 + [C.POINTER(_dbl)]),
    "armon_hip_conservation_vars": (_ci, [_vp, Range, _dbl, _dp, _dp, C.POINTER(_dbl * 2)]),
    "armon_hip_init_test": (_ci, [_vp, Range, _ci, _i64, _i64, _ci, C.POINTER(_i64 * 2),
                                  C.POINTER(_i64 * 2), C.POINTER(_dbl * 2), C.POINTER(_dbl * 2),
                                  _dbl, C.POINTER(BlockDataPtrs)]),
    "armon_hip_sweep": (_ci, [_vp, C.POINTER(SweepDesc)]),
    "armon_hip_dt_state_step": (_ci, [_vp, _vp, _vp, _dbl, _dbl, _i64, _ci, _dbl]),
    "armon_hip_dt_state_step_f32": (_ci, [_vp, _vp, _vp, _dbl, _dbl, _i64, _ci, _dbl]),
    "armon_hip_graph_begin": (_ci, [_vp]),
    "armon_hip_graph_end": (_ci, [_vp, C.POINTER(_vp)]),
    "armon_hip_graph_launch": (_ci, [_vp, _vp]),
    "armon_hip_graph_destroy": (_ci, [_vp]),
    "armon_hip_tune_placement": (_ci, [_vp, C.POINTER(SweepDesc), C.POINTER(SweepDesc), C.POINTER(_vp), _ci,
                                       C.c_size_t, _ci, C.POINTER(_ci * 8), C.POINTER(_dbl)]),
    "armon_hip_cycle_xy": (_ci, [_vp, C.POINTER(SweepDesc), C.POINTER(SweepDesc)]),
    "armon_hip_choose_placement": (_ci, [_vp, C.POINTER(SweepDesc), C.POINTER(SweepDesc), C.POINTER(_vp), _ci,
                                         C.c_size_t, _ci, _dbl, C.POINTER(_ci * 8), C.POINTER(_dbl), C.POINTER(_ci)]),
    "armon_hip_mgpu_init": (_ci, [_ci, _ci, C.POINTER(_ci), C.POINTER(_vp)]),
    "armon_hip_mgpu_unique_id": (_ci, [_vp]),
    "armon_hip_mgpu_init_rank": (_ci, [_ci, _ci, _ci, _ci, _vp, _vp, C.POINTER(_vp)]),
    "armon_hip_mgpu_prepare_rank": (_ci, [_ci, _ci, _ci, _ci, _vp, C.POINTER(_vp)]),
    "armon_hip_mgpu_connect": (_ci, [_vp, _vp]),
    "armon_hip_mgpu_destroy": (_ci, [_vp]),
    "armon_hip_mgpu_set_periodic": (_ci, [_vp, _ci, _ci]),
    "armon_hip_mgpu_force_peer_copy": (_ci, [_vp, _ci]),
    "armon_hip_mgpu_set_chaos": (_ci, [_vp, C.c_uint, C.c_uint64]),
    "armon_hip_mgpu_n_local": (_ci, [_vp]),
    "armon_hip_mgpu_ctx": (_vp, [_vp, _ci]),
    "armon_hip_mgpu_tile_info": (_ci, [_vp, _ci, C.POINTER(_ci), C.POINTER(_ci * 2), C.POINTER(_ci * 4)]),
    "armon_hip_halo_exchange_start": (_ci, [_vp, _ci, C.POINTER(HaloDesc)]),
    "armon_hip_halo_exchange_finish": (_ci, [_vp, _ci, C.POINTER(HaloDesc)]),
    "armon_hip_halo_exchange": (_ci, [_vp, _ci, C.POINTER(HaloDesc)]),
    "armon_hip_dt_allreduce": (_ci, [_vp, C.POINTER(_vp)]),
    "armon_hip_mgpu_edge_ctx": (_vp, [_vp, _ci]),
    "armon_hip_mgpu_edge_dt": (_vp, [_vp, _ci]),
    "armon_hip_halo_exchange_finish_edge": (_ci, [_vp, _ci, C.POINTER(HaloDesc)]),
    "armon_hip_mgpu_edge_join": (_ci, [_vp, C.POINTER(_vp)]),
    "armon_hip_mgpu_allreduce_host": (_ci, [_vp, _ci, _ci, C.POINTER(_dbl)]),
    "armon_hip_mgpu_cycle": (_ci, [_vp, C.POINTER(CyclePlan), C.POINTER(TileCycle)]),
    "armon_hip_mgpu_drain": (_ci, [_vp, C.POINTER(TileCycle)]),
    "armon_hip_mgpu_set_threads": (_ci, [_vp, _ci]),
    "armon_hip_mgpu_sync": (_ci, [_vp]),
    "armon_hip_halo_ranges": (_ci, [_i64, _i64, _ci, _ci, C.POINTER(Range), C.POINTER(Range), C.POINTER(_i64)]),
}


def _add_f32_signatures():
    """Every kernel entry point also exists with an `_f32` suffix: float arrays and scalars."""
    flt = C.c_float
    for name in ("perfect_gas_EOS", "bizarrium_EOS", "acoustic", "acoustic_GAD", "cell_update",
                 "advection_first_order", "advection_second_order", "euler_projection", "boundary_conditions",
                 "pack_to_array", "unpack_from_array", "dtCFL_async", "dtCFL", "conservation_vars", "init_test"):
        res, args = SIGNATURES["armon_hip_" + name]
        conv = []
        for a in args:
            if a is _dbl:
                conv.append(flt)
            elif a is C.POINTER(_dbl):
                conv.append(C.POINTER(flt))
            elif a is C.POINTER(_dbl * 2):
                conv.append(C.POINTER(flt * 2))
            else:
                conv.append(a)
        SIGNATURES["armon_hip_" + name + "_f32"] = (res, conv)
    for name in ("halo_exchange_start", "halo_exchange_finish", "halo_exchange", "dt_allreduce",
                 "halo_exchange_finish_edge", "mgpu_edge_join", "mgpu_cycle", "mgpu_drain"):
        SIGNATURES["armon_hip_" + name + "_f32"] = SIGNATURES["armon_hip_" + name]
    SIGNATURES["armon_hip_sweep_f32"] = SIGNATURES["armon_hip_sweep"]
    SIGNATURES["armon_hip_tune_placement_f32"] = SIGNATURES["armon_hip_tune_placement"]
    SIGNATURES["armon_hip_choose_placement_f32"] = SIGNATURES["armon_hip_choose_placement"]


_add_f32_signatures()


def load_at(path):
    """Load one build of the library and declare every entry point (used for the default build, and by
    tools/ab_sweep.py to time several builds side by side)."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python armon.jl_amd/build.py` "
            "(there is no CPU fallback for the device backend)")
    L = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)   # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if L.armon_hip_flt_size() != 8 or L.armon_hip_idx_size() != 8:   # ref ext/ArmonKokkos.jl:122-139
        raise RuntimeError("libarmon_hip.so: unexpected flt_size/idx_size")
    return L


def lib():
    """Load libarmon_hip.so (once). Raises if it is not built."""
    global _lib
    if _lib is None:
        _lib = load_at(LIB_PATH)
    return _lib


ALT_LIB_PATH = os.path.join(_HERE, "libarmon_hip_alt.so")
_alt = None


class alt_kernels:
    """Context manager for the tests and tools that exercise the measured-and-rejected kernel forms (x_kernel 2 / 3,
    armon_hip_cycle_xy): inside it every call of this package goes to libarmon_hip_alt.so — the same sources built with
    -DARMON_ALT_KERNELS. Objects that hold a context (devices, grids, tile groups) must be created AND released inside."""

    def __enter__(self):
        global _lib, _alt
        if _alt is None:
            _alt = load_at(ALT_LIB_PATH)
        self._saved, _lib = _lib, _alt
        return _alt

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def check(rc):
    """Turn a non-zero status into SolverException(:cpp, msg) (ref ext/ArmonKokkos.jl:72-76)."""
    if rc != 0:
        msg = lib().armon_hip_last_error().decode(errors="replace")
        raise SolverException("cpp" if rc != 4 else "time", f"[{rc}] {msg}")
