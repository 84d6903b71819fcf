"""Randomised soak of the emission scorer against the C twin (round 5, after the weight table's LDS row stride began to follow the
launch's state count): feature dimensions 3..400 (multiples of 4 and not), 1..3 class sets of 1..32 states in one launch, ragged
videos from 1 frame up, small launches (one tile per wave) and launches of >= 131 072 frames (pairs of tiles), narration constraints
on and off, fp64 and fp32 outputs.  Bars: the unit tests' (fp64 rtol 1e-12 / atol 1e-9, fp32 rtol 2e-7 / atol 1e-6).
usage: soak_emission.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import test_gpu_viterbi as tv
from oracle import factored as F

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ops = tv._ops()
dev = torch.device('cuda:0')
t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
t0, n, frames, worst, last, npair = time.time(), 0, 0, 0.0, time.time(), 0
while time.time() - t0 < budget:
    d = int(g.choice([3, 4, 17, 24, 63, 64, 100, 128, 200, 257, 300, 304, 400]))
    ng = int(g.integers(1, 4))
    cs = [int(g.integers(1, 33)) for _ in range(ng)]
    cm = max(cs) + int(g.integers(0, 3)) * (max(cs) < 30)
    big = g.random() < 0.35 and d <= 128
    b = int(g.integers(1, 41))
    if big:
        lengths = g.integers(2500, 3600, size=max(b, 132000 // 3000 + 2))
    else:
        lengths = g.integers(1, int(g.choice([20, 200, 700])) + 1, size=b)
    b = len(lengths)
    if b > 2: lengths[1] = int(g.integers(1, 16))
    group = g.integers(0, ng, size=b).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    total = int(lengths.sum())
    with_cons = bool(g.random() < 0.4)
    x = g.standard_normal((total, d)).astype(np.float32)
    var = 0.5 + g.random(d)
    lognorm = float(-0.5 * d * np.log(2 * np.pi) - 0.5 * np.log(var).sum())
    mus = [g.standard_normal((c, d)) * 0.5 for c in cs]
    w = np.zeros((ng, d, cm)); cst = np.zeros((ng, cm))
    for k, (c, mu) in enumerate(zip(cs, mus)):
        w[k, :, :c] = (mu / var).T
        cst[k, :c] = lognorm - 0.5 * (mu * mu / var).sum(1)
    cons = ((g.random((total, cm)) < 0.1) * -1e4).astype(np.float32) if with_cons else None
    batch = ops.Batch(lengths, cs, 4, c_max=cm, t_max=int(lengths.max()), frame_offset=offs, group=group, total_frames=total, d=d)
    e64, e32 = ops.emission(batch, torch.tensor(x, device=dev), t64(w), t64(cst), t64(1.0 / var),
                            torch.tensor(cons, device=dev) if with_cons else None, True, True)
    torch.cuda.synchronize()
    e64 = e64.cpu().numpy(); e32 = e32.cpu().numpy()
    for i in range(b):
        c, t, o = cs[group[i]], int(lengths[i]), int(offs[i])
        ref = F.emission(x[None, o:o + t], [t], mus[group[i]], 1.0 / var, lognorm,
                         cons[None, o:o + t, :c].astype(np.float64) if with_cons else None)[0]
        got = e64[o:o + t, :c]
        err = np.abs(got - ref) / (1e-9 + 1e-12 * np.abs(ref))
        worst = max(worst, float(err.max()))
        assert err.max() <= 1.0, (d, cs, cm, big, with_cons, i, float(err.max()))
        np.testing.assert_allclose(e32[o:o + t, :c], ref, rtol=2e-7, atol=1e-6)
    n += 1; frames += total; npair += big
    if time.time() - last > 20:
        last = time.time()
        print('  ... %d launches (%d of >= 131 072 frames), %d frames, worst fp64 error %.2f of the bar' % (n, npair, frames, worst), flush=True)
print('soak ok: %d emission launches (%d of >= 131 072 frames: pairs of tiles), %d frames, %.0f s; worst fp64 error %.3f of the bar (rtol 1e-12, atol 1e-9)'
      % (n, npair, frames, time.time() - t0, worst))
