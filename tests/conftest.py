import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "lattice-boltzmann-method_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="session")
def oracle():
    from pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The unmodified reference on CPU libtorch; only where oracle/_ref has been built."""
    import torch  # noqa: F401  (loads libtorch for the .so)
    from pyoracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return Ref()
