"""SURVEY 8(d) baseline B1: the dense reference path (oracle/dense_ref.py: log_hsmm potentials b x N x K x C x C in
fp32 + the sequential max-DP of pytorch-struct) on the host cores, at the reduced shapes the dense tensor fits at --
(T, K) in {(2048, 256), (4096, 256), (4096, 512)}, C1 = 16, D = 200 -- with all threads and with one, and the cost of
recovering the arg-max the way the library does it (autograd through torch.max) measured separately at the sizes where
it finishes.  Extrapolation to the metric's shape (T ~ 10 000, K = 1024, C1 ~ 20) goes with T*K*C^2.

    python scripts/b1_series.py [max_seconds_per_point]        -> table on stdout (copy under profiles/)
CPU only (the oracle is the thing measured here: this is the cpu_baseline series, not a product path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense_ref as O
from oracle import factored as F

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
n_all = min(16, F.host_cores())


def problem(t, c, k, d=200, seed=0):
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(c, d, generator=g) * 0.3
    cov = torch.rand(d, generator=g) * 0.6 + 0.7
    lab = torch.randint(0, c, (t // 40 + 2,), generator=g).repeat_interleave(40)[:t]
    x = means[lab] + cov.sqrt() * torch.randn(t, d, generator=g)
    p = O.RefParams(c, torch.rand(c, generator=g) * 3 + 2.5, means, cov, torch.randn(c, c, generator=g),
                    torch.randn(c, generator=g), k, True)
    return p, x


def timed(fn):
    t0 = time.perf_counter()
    out = fn()
    return time.perf_counter() - t0, out


rows = []
for (t, k) in ((2048, 256), (4096, 256), (4096, 512)):
    c = 16
    p, x = problem(t, c, k)
    for threads in (n_all, 1):
        torch.set_num_threads(threads)
        with torch.no_grad():
            dt_s, (scores, _) = timed(lambda: O.score_features(p, x[None], torch.tensor([t]), None))
            dt_f, _ = timed(lambda: O.semimarkov_dp(scores, torch.tensor([t + 1]), O.MaxSemiring))
            dt_b, _ = timed(lambda: O.viterbi_backpointers(scores, torch.tensor([t + 1])))
        gb = scores.numel() * 4 / 1e9
        rows.append((t, k, c, threads, gb, dt_s, dt_f, dt_b))
        print("T=%d K=%d C1=%d threads=%d: potentials %.2f GB  score_features %.2f s  max-DP forward %.2f s  "
              "forward + back-pointers %.2f s  -> %.1f frames/s (score + back-pointer DP)"
              % (t, k, c, threads, gb, dt_s, dt_f, dt_b, t / (dt_s + dt_b)), flush=True)
        del scores
# the library's own arg-max: autograd through the DP (>95 % of the reference's decode time)
torch.set_num_threads(n_all)
print()
for (t, k) in ((256, 64), (512, 64), (1024, 64), (512, 256), (1024, 256), (2048, 256)):
    c = 16
    p, x = problem(t, c, k)
    with torch.no_grad():
        scores, _ = O.score_features(p, x[None], torch.tensor([t]), None)
    dt_f, _ = timed(lambda: O.semimarkov_dp(scores, torch.tensor([t + 1]), O.MaxSemiring))
    est = dt_f * 60
    if est > budget * 4:
        print("T=%d K=%d: autograd arg-max skipped (forward alone %.2f s)" % (t, k, dt_f), flush=True)
        continue
    dt_a, _ = timed(lambda: O.marginals(scores, torch.tensor([t + 1]), O.MaxSemiring))
    print("T=%d K=%d C1=%d threads=%d: max-DP forward %.2f s, forward + autograd arg-max (torch_struct's mechanism) %.2f s "
          "= %.0fx the forward  -> %.1f frames/s" % (t, k, c, n_all, dt_f, dt_a, dt_a / dt_f, t / dt_a), flush=True)
    if dt_a > budget:
        break
# extrapolation of the back-pointer form to the metric's shape
t, k, c, threads, gb, dt_s, dt_f, dt_b = [r for r in rows if r[3] == n_all][-1]
work = lambda t_, k_, c_: t_ * k_ * (c_ + 1) ** 2
for (tt, kk, cc) in ((10000, 1024, 20), (14000, 1024, 23)):
    sec = (dt_s + dt_b) * work(tt, kk, cc) / work(t, k, c)
    print("extrapolated to T=%d K=%d C1=%d (x %.1f the work of T=%d K=%d C1=%d; potentials %.1f GB: does not fit the "
          "reference's fp32 dense path on this host): %.0f s per video = %.1f frames/s on %d threads"
          % (tt, kk, cc, work(tt, kk, cc) / work(t, k, c), t, k, c, 4.0 * tt * kk * (cc + 1) ** 2 / 1e9, sec, tt / sec, threads))
