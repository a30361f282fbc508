import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, ref):
    """Parity metric of SURVEY.md section 8(d): max|a - ref| / max|ref| per array."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    denom = np.max(np.abs(ref))
    return float(np.max(np.abs(a - ref)) / (denom if denom > 0 else 1.0))


@pytest.fixture(scope="session")
def relerr():
    return rel_err
