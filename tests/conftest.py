import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def t(a, dtype=None):
    """numpy -> torch (int32 index arrays become int64)."""
    x = torch.from_numpy(np.asarray(a))
    if x.dtype == torch.int32:
        x = x.long()
    if dtype is not None and x.is_floating_point():
        x = x.to(dtype)
    return x


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()
