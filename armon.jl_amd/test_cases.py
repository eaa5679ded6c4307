"""Test-case definitions (ref src/tests.jl): defaults, boundary conditions, EOS selection, tags."""
import math
from dataclasses import dataclass

from ._lib import solver_error
from .blocking import Side

FreeFlow, Dirichlet = "FreeFlow", "Dirichlet"   # ref src/tests.jl:124


@dataclass(frozen=True)
class TestCase:
    name: str
    tag: int                      # C tag (include/armon_hip.h, ref ext/ArmonKokkos.jl:60-69)
    domain_size: tuple = (1., 1.)     # ref src/tests.jl:32-33
    origin: tuple = (0., 0.)          # ref :35-36
    cfl: float = 0.95                 # ref :38-40
    maxtime: float = 0.20             # ref :42-44
    boundaries: tuple = (Dirichlet,) * 4   # left, right, bottom, top — ref :164-211
    conservative: bool = True         # ref :48-49
    eos: str = "perfect_gas"          # ref src/kernels.jl:151-161
    r: float = 0.                     # Sedov only, ref :15-19

    @property
    def gamma(self):                  # ref src/tests.jl:46
        return 7 / 5

    def boundary_condition(self, side):
        """(u_factor, v_factor) for ``side`` — ref src/tests.jl:150-161."""
        cond = self.boundaries[int(side) - 1]
        if cond == FreeFlow:
            return (1., 1.)
        return (-1., 1.) if side in (Side.Left, Side.Right) else (1., -1.)


_CASES = {
    "Sod": dict(tag=0, boundaries=(Dirichlet, Dirichlet, FreeFlow, FreeFlow)),
    "Sod_y": dict(tag=1, boundaries=(FreeFlow, FreeFlow, Dirichlet, Dirichlet)),
    "Sod_circ": dict(tag=2, boundaries=(Dirichlet,) * 4),
    "Bizarrium": dict(tag=3, cfl=0.6, maxtime=80e-6, conservative=False, eos="bizarrium",
                      boundaries=(Dirichlet, FreeFlow, Dirichlet, Dirichlet)),
    "Sedov": dict(tag=4, domain_size=(2., 2.), origin=(-1., -1.), cfl=0.7, maxtime=1.0,
                  boundaries=(FreeFlow,) * 4),
    "DebugIndexes": dict(tag=5, cfl=0., maxtime=0., boundaries=(Dirichlet,) * 4),
}


def default_domain_size(name):
    return _known(name).get("domain_size", (1., 1.))


def default_domain_origin(name):
    return _known(name).get("origin", (0., 0.))


def _known(name):
    if name not in _CASES:
        solver_error("config", f"Unknown test case: '{name}'")   # ref src/tests.jl:27
    return _CASES[name]


def create_test(name, dX, T=float):
    """ref src/tests.jl:13-19 — Sedov's radius depends on the cell size: r::T = hypot(Δx...) / sqrt(2)."""
    import numpy as np
    kw = dict(_known(name))
    if name == "Sedov":
        # hypot in T, the division by sqrt(2)::Float64 in Float64, then the conversion to T — as Julia evaluates it. (numpy 2
        # keeps `float32 / python_float` in float32, one ulp away on non-square cells: found by tests/test_gpu_random_shapes.py)
        kw["r"] = float(T(float(np.hypot(T(dX[0]), T(dX[1]))) / math.sqrt(2)))
    return TestCase(name=name, **kw)
