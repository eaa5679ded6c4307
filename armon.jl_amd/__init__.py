"""armon.jl_amd — MI355X-native device backend for the direction-split hot path of Armon.jl.

The directory name carries a dot, so import it through the root-level shim: ``import armon_amd``.
Public surface = the reference's exports (ref src/Armon.jl:15-16): ``ArmonParameters``, ``BlockGrid``,
``SolverStats``, ``armon``, ``data_type``, ``memory_required``, ``device_to_host!``/``host_to_device!``
(methods of BlockGrid here).
"""
from ._lib import LIB_PATH, SolverException, lib          # noqa: F401
from .blocking import Axis, BlockSize, DomainRange, Side, StepRange   # noqa: F401
from .parameters import ArmonParameters, memory_required   # noqa: F401
from .solver import BlockGrid, SolverStats, armon          # noqa: F401


def data_type(params):
    return params.data_type
