import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


def cn(rng, n, dtype=np.complex64):
    """Circular complex normal CN(0,1) samples."""
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(dtype)


def qpsk(rng, n, dtype=np.complex64):
    return np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, n))).astype(dtype)
