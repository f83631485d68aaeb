import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def weights_from_fixture(fx):
    return {k[3:]: v for k, v in fx.items() if k.startswith("w::")}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def autograd_training(monkeypatch):
    """GCDenoiser.loss / BesoAgent.train_step on the torch-autograd comparator of tests/autograd_reference.py for the
    duration of a test (the product itself has no torch-op evaluation of the network): host-logic tests on CPU, and the
    reference side of the HIP-vs-autograd tests on the GPU."""
    from autograd_reference import use_autograd_training
    use_autograd_training(monkeypatch)
    return monkeypatch
