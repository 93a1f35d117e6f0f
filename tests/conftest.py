import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)"
    )
    # the CPU oracles are ~600 tiny eager ops per unroll: on a 256-CPU host the
    # default intra-op thread count makes them 10x SLOWER than 16 threads
    # (bench.py's thread sweep: 137 ms at 16 threads, 1 995 ms at 128)
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except ImportError:
        pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max(|b|) - the 'relative to tensor scale' error used for
    the 1e-4 fp32 parity bar of BASELINE.json:north_star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.max(np.abs(b)), 1e-30)
    return float(np.max(np.abs(a - b)) / denom)
