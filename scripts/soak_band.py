"""Randomised soak of the Viterbi kernels against the C twin: random / structured / integer lattices, ragged batches, lengths from
1 frame to a few thousand.  Default: the BAND kernel (K 513..1024, 1..32 states); round 4 added flat and ramp lattices.
usage: soak_band.py [seconds] [seed] [kmin kmax cmax]     (e.g. 120 2 2 512 32: the ring kernels of every size)"""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import test_gpu_viterbi as tv
from oracle import factored as F

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KMIN, KMAX, CMAX = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (513, 1024, 32)
ops = tv._ops()
t0, n, frames = time.time(), 0, 0
modes, split = [0, 0, 0, 0], [0, 0, 0]
while time.time() - t0 < budget:
    c = int(g.integers(1, CMAX + 1))
    k = int(g.integers(KMIN, KMAX + 1))
    b = int(g.integers(1, 5))
    kind = ('random', 'structured', 'integer', 'masked', 'masked_inf', 'flat', 'ramp')[int(g.integers(0, 7))]
    tmax = int(g.choice([700, 1100, 1600, 2600, 4200, 6500]))
    lengths = [int(x) for x in g.integers(1, tmax + 1, size=b)]
    lengths[int(g.integers(0, b))] = tmax
    if b > 1 and g.random() < 0.3:
        lengths[(lengths.index(tmax) + 1) % b] = int(g.integers(1, 20))     # a tiny video beside the long ones
    seed = int(g.integers(0, 10 ** 6))
    if kind.startswith('masked'):
        # the reference's constrained path: -1e9 (or -inf) masks, -1e4 narration penalties; -inf needs a feasible chain
        if kind == 'masked_inf':
            lengths = [max(t, c + 1) for t in lengths]
        p = tv.masked_problem(seed, lengths, max(c, 2), k, neg_inf=(kind == 'masked_inf'))
    elif kind == 'structured':
        p = tv.structured_problem(seed, lengths, c, k, margin=float(g.choice([4.0, 18.0])), rate=(min(5, max(1, k // 4)), max(2, min(k - 1, int(g.choice([60, 400, 900]))))))
    elif kind == 'ramp':
        # ramp-shaped length tables (round 4: the source-dominance threshold anywhere, rising / falling h), integer ones too
        p = tv.structured_problem(seed, lengths, c, k, margin=float(g.choice([4.0, 18.0])))
        gg = np.random.default_rng(seed + 1)
        p['lens'] = float(g.choice([-30.0, -0.5, 0.0, 0.5, 30.0])) * np.arange(k)[:, None] + gg.uniform(-3.0, 3.0, size=(1, c)) + np.zeros((k, c))
        if g.random() < 0.5:
            for key in ('elp', 'lens', 'trans', 'init'):
                p[key] = np.round(p[key])
    else:
        p = tv.make_problem(seed, b, tmax, c, k, integer=(kind == 'integer'))
        p['lengths'] = np.asarray(lengths)
        if kind == 'flat':
            # every state emits the same, tables are proper log-probabilities: nothing ever falls behind (the band skip test's
            # and the source-dominance test's worst case)
            gg = np.random.default_rng(seed + 2)
            p['elp'] = p['elp'][:, :, :1] + 1e-3 * gg.standard_normal(p['elp'].shape)
            p['lens'] = -np.log(k) - 0.05 * gg.random(p['lens'].shape)
            tr = gg.standard_normal(p['trans'].shape)
            p['trans'] = tr - np.log(np.exp(tr).sum(0, keepdims=True))
            p['init'] = np.full_like(p['init'], -np.log(c))
    # round 5: the launch's shape is drawn too -- long videos cut along the time axis into the smallest units the planner makes
    # (SMM_CHUNK_P=1; with a short warm-up now and then, so that cuts fail to certify and the repair launch runs), <= 16-state
    # videos in four-wave workgroups (SMM_SMALL_WG=2), or the plain launch
    mode = int(g.integers(0, 4))
    os.environ['SMM_CHUNK'] = '1' if mode in (1, 2) else '0'
    os.environ['SMM_CHUNK_P'] = '1' if mode in (1, 2) else '0'
    os.environ['SMM_CHUNK_WC'] = '64' if mode == 2 else '512'
    os.environ['SMM_SMALL_WG'] = '2' if mode == 3 else '0'
    ops.reload_env()
    modes[mode] += 1
    out = tv.run_gpu(p)
    split[0] += int(out['_err'][4]); split[1] += int(out['_err'][5]); split[2] += int(out['_err'][7])
    try:
        spans, v = tv.run_oracle(p)
    except AssertionError as e:
        print('ORACLE REFUSED', dict(c=c, k=k, lengths=lengths, kind=kind, seed=seed), e, 'nan in inputs:',
              {kk: bool(np.isnan(vv).any()) for kk, vv in p.items() if isinstance(vv, np.ndarray) and vv.dtype == np.float64})
        sys.exit(1)
    try:
        tv.check(p, out, spans, v)
        assert out['_err'][0] == 0
    except AssertionError as e:
        print('MISMATCH', dict(c=c, k=k, lengths=lengths, kind=kind, seed=seed, mode=mode), str(e)[:300])
        sys.exit(1)
    n += 1; frames += sum(lengths)
print('soak ok: %d launches, %d frames, %.0f s; launches plain / time-split / time-split with a 64-position warm-up / small workgroups: %s; videos split %d, repaired %d, ties resolved %d'
      % (n, frames, time.time() - t0, modes, split[0], split[1], split[2]))
