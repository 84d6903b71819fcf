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


@pytest.fixture(autouse=True)
def _smm_switches_follow_the_environment(monkeypatch):
    """libsmmdp reads its SMM_* tuning switches once, at first use (include/smmdp.h: smm_env_reload); tests flip them
    inside one process with monkeypatch.setenv / delenv, so every change is followed by a reload -- and every test
    starts from the environment as it is (the previous test's changes have been undone by then)."""
    from action_segmentation_amd import _lib
    _lib.reload_env()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_and_reload(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith('SMM_'):
            _lib.reload_env()

    def delenv_and_reload(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith('SMM_'):
            _lib.reload_env()
    monkeypatch.setenv, monkeypatch.delenv = setenv_and_reload, delenv_and_reload
    yield
