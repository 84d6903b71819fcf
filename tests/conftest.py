import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _calm_thread_pools():
    """The GPU box shows 256 logical CPUs and grants a fraction of them (cgroup quota): default-sized OpenMP / torch
    intra-op pools spin on all 256 and get the whole process throttled for ~90 ms at a time."""
    from oracle import factored
    n = factored.host_cores()
    os.environ.setdefault('OMP_NUM_THREADS', str(n))
    try:
        import torch
        torch.set_num_threads(min(n, 16))
    except ImportError:
        pass


def pytest_configure(config):
    _calm_thread_pools()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
    return dict(np.load(path))


def has_gpu():
    import torch
    return torch.cuda.is_available()
