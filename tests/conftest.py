import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
