"""Does the split decode (smm_api.hip: choose_split -- the launch's longest videos scored and decoded on the caller's
stream, the rest on a second stream beside them) ever LOSE against the plain two-launch decode?  Six length distributions
that are not the benchmark's: for each, smm_decode_f32 with the split allowed and with SMM_NO_SPLIT=1, same box, same
process, interleaved repetitions.  -> profiles/round4_split_sweep.txt
usage: python scripts/sweep_split.py [reps]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from scipy.special import gammaln
from action_segmentation_amd import ops, _lib
import os

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device('cuda:0')
D, C, K = 200, 15, 1024


def corpus(seed, lengths):
    g = np.random.default_rng(seed)
    tg = torch.Generator(device=dev).manual_seed(seed)
    sigma = torch.tensor(g.uniform(0.7, 1.3, size=D), dtype=torch.float32, device=dev)
    mu = g.normal(0, 0.3, size=(C, D))
    rates = g.uniform(20, 400, size=C)
    labs = []
    for t in lengths:
        out, cur, tot = [], int(g.integers(0, C)), 0
        while tot < t:
            ln = int(np.clip(g.poisson(rates[cur]), 1, K - 1))
            out.append(np.full(ln, cur)); tot += ln; cur = (cur + 1) % C
        labs.append(np.concatenate(out)[:t])
    lab = torch.from_numpy(np.concatenate(labs)).to(dev)
    x = torch.tensor(mu, dtype=torch.float32, device=dev)[lab] + sigma * torch.randn((lab.numel(), D), generator=tg, device=dev)
    var = (sigma.double() ** 2).cpu().numpy()
    w = (mu / var).T.copy()
    lognorm = -0.5 * np.log(var).sum() - 0.5 * D * np.log(2 * np.pi)
    cst = -0.5 * (mu * mu / var).sum(1) + lognorm
    trans = np.log(g.dirichlet(np.ones(C) * 0.5, size=C).T + 1e-3)
    trans -= np.log(np.exp(trans).sum(0, keepdims=True))
    init = np.log(g.dirichlet(np.ones(C)))
    kk = np.arange(K)[:, None]
    lens = kk * np.log(rates) - rates - gammaln(kk + 1)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
    ln = np.asarray(lengths, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(ln)[:-1]])
    tmax = int(ln.max())
    batch = ops.Batch(ln, [C], K, c_max=C, frame_offset=off, kp=[min(K, tmax)] * len(ln), d=D, t_max=tmax, total_frames=int(ln.sum()))
    return batch, (x, t64(w[None]), t64(cst[None]), t64(1.0 / var), t64(trans[None]), t64(init[None]), t64(lens[None]))


def timed(batch, args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = ops.decode(batch, *args, want_spans=False, want_labels=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), out


g = np.random.default_rng(0)
logn = lambda n, mu: np.clip(g.lognormal(np.log(mu), 0.5, size=n), 500, 14000).astype(int)
DISTS = [
    ("cfg3-like lognormal(6000), 360 videos", logn(360, 6000)),
    ("uniform 500..14000, 300 videos", g.integers(500, 14001, size=300)),
    ("bimodal 150 x ~2000 + 150 x ~12000", np.concatenate([g.integers(1800, 2200, size=150), g.integers(11500, 12500, size=150)])),
    ("one 14000-frame outlier over 300 x ~3000", np.concatenate([[14000], g.integers(2800, 3200, size=300)])),
    ("40 videos, lognormal(6000)", logn(40, 6000)),
    ("1000 videos, lognormal(2000)", np.clip(g.lognormal(np.log(2000), 0.5, size=1000), 500, 14000).astype(int)),
]
worst = 0.0
for name, lengths in DISTS:
    batch, args = corpus(1, [int(t) for t in lengths])
    res = {}
    labels = {}
    for mode in ('split', 'nosplit'):
        if mode == 'nosplit':
            os.environ['SMM_NO_SPLIT'] = '1'
        else:
            os.environ.pop('SMM_NO_SPLIT', None)
        _lib.reload_env()
        for _ in range(3):
            timed(batch, args)                                  # (the third call runs from a resident plan)
        res[mode] = []
    for r in range(reps):
        for mode in ('split', 'nosplit'):
            if mode == 'nosplit':
                os.environ['SMM_NO_SPLIT'] = '1'
            else:
                os.environ.pop('SMM_NO_SPLIT', None)
            _lib.reload_env()
            ops.dp_timing_read(); ops.dp_timing(True)
            ms, out = timed(batch, args)
            ops.dp_timing(False)
            tags = [t for _, t in ops.dp_timing_read(tagged=True)]
            res[mode].append(ms)
            labels[mode] = out['labels'].clone()
            res[mode + '_launches'] = len(tags)
    a, b = float(np.median(res['split'])), float(np.median(res['nosplit']))
    same = bool(torch.equal(labels['split'], labels['nosplit']))
    worst = max(worst, a / b - 1.0)
    print("%-44s %8d frames: split allowed %.3f ms (%d DP launch%s) | SMM_NO_SPLIT %.3f ms | %+5.1f %%  labels equal: %s"
          % (name, int(sum(lengths)), a, res['split_launches'], 'es' if res['split_launches'] > 1 else '', b, 100.0 * (a / b - 1.0), same))
    sys.stdout.flush()
    del batch, args
    torch.cuda.empty_cache()
print("worst case of 'split allowed' against 'never split': %+.1f %%" % (100.0 * worst))
