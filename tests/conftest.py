import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests load armon.jl_amd/libarmon_hip.so (built in-tree, git-ignored): build it when it is missing and
    hipcc is there (≈3 min; cross-compiles without a GPU). There is no CPU stand-in to fall back to."""
    libs = [os.path.join(ROOT, "armon.jl_amd", name) for name in ("libarmon_hip.so", "libarmon_hip_alt.so")]
    if not all(os.path.exists(p) for p in libs) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        import importlib.util
        spec = importlib.util.spec_from_file_location("armon_build", os.path.join(ROOT, "armon.jl_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): compiled on demand with gcc."""
    from oracle import oracle as O
    O.lib()
    return O


def load_golden(test, bits=64):
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", f"ref_{test}_{bits}bits.npz")
    return np.load(path)
