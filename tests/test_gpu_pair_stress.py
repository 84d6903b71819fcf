"""Randomised parity sweep of the two-CU PAIR mode (leader / follower workgroups exchanging rows through HBM with
progress counters): many shapes in one process, every video forced into a pair, bit-exact against the C twin.
A visibility or ordering bug in the exchange shows up here as a sporadic mismatch."""
import numpy as np
import pytest
import torch

from oracle import factored as F
from test_gpu_viterbi import make_problem, run_gpu, run_oracle, check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('chunk', range(6))
def test_pair_mode_random_shapes(chunk, monkeypatch):
    g = np.random.default_rng(9000 + chunk)
    for it in range(10):
        b = int(g.integers(1, 7))
        c = int(g.integers(1, 24))
        k = int(g.integers(513, 1025))
        tmax = int(g.integers(k, 3200))                       # kp = min(k, tmax) > 512: 1024-slot rings
        ends = bool(g.integers(0, 2))
        monkeypatch.setenv('SMM_PAIRS', str(int(g.integers(1, b + 1))))
        p = make_problem(int(g.integers(0, 10 ** 6)), b, tmax, c, k, ends=ends, scale=float(g.choice([0.5, 3.0])),
                         min_len=int(g.integers(1, 200)))
        p['lengths'][int(g.integers(0, b))] = tmax
        out = run_gpu(p)
        spans, v = run_oracle(p)
        check(p, out, spans, v)


def test_pair_mode_repeated_launches_are_identical(monkeypatch):
    """The same paired launch 30 times back to back (workspace and counters reused): identical bits every time."""
    monkeypatch.setenv('SMM_PAIRS', '4')
    p = make_problem(4242, 4, 2600, 19, 1024, ends=True)
    ref = run_gpu(p)
    spans, v = run_oracle(p)
    check(p, ref, spans, v)
    for _ in range(30):
        out = run_gpu(p)
        for key in ('best', 'spans', 'labels', 'n_segs'):
            np.testing.assert_array_equal(out[key], ref[key])
