"""Reference-default shapes (K = 20, 5 videos, T ~ 300): where does a decode call spend its time?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops
dev = torch.device('cuda:0')
g = np.random.default_rng(0)
for (b, T, C, K, D) in ((5, 300, 12, 20, 200), (5, 300, 23, 20, 200), (5, 1000, 12, 20, 200), (64, 300, 12, 20, 200), (5, 300, 12, 64, 200)):
    lengths = np.full(b, T); lengths[1:] = g.integers(T // 2, T, size=b - 1)
    batch = ops.Batch(lengths, [C], K, c_max=C, t_max=T, total_frames=b * T, d=D)
    f64 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev).contiguous()
    x = torch.tensor(g.standard_normal((b * T, D)).astype(np.float32), device=dev)
    w, cst, iv = f64(g.standard_normal((1, D, C)) * 0.3), f64(g.standard_normal((1, C))), f64(0.5 + g.random(D))
    trans = f64(np.log(g.dirichlet(np.ones(C), size=C).T)[None]); init = f64(np.log(g.dirichlet(np.ones(C)))[None])
    lens = f64(-0.1 * (np.arange(K)[:, None] - 8.0) ** 2 + np.zeros((K, C)))[None]
    st = torch.cuda.current_stream()
    def run():
        return ops.decode(batch, x, w, cst, iv, trans, init, lens, want_spans=True, want_labels=True)
    for _ in range(3):
        out = run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 20
    t0 = time.perf_counter()
    ev[0].record(st)
    for _ in range(n):
        out = run()
    ev[1].record(st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    segs = out['n_segs'].cpu().numpy()
    print('b=%d T=%d C=%d K=%d: GPU %.1f us per decode (emission + DP + recovery/meta launches), host issue %.1f us, '
          'segments per video %.0f' % (b, T, C, K, ev[0].elapsed_time(ev[1]) * 1e3 / n, (t1 - t0) * 1e6 / n, segs.mean()), flush=True)
    elp, _ = ops.emission(batch, x, w, cst, iv)
    torch.cuda.synchronize()
    ev[0].record(st)
    for _ in range(n):
        o2 = ops.viterbi(batch, elp, trans, init, lens, want_spans=True, want_labels=True)
    ev[1].record(st)
    torch.cuda.synchronize()
    print('     viterbi alone %.1f us' % (ev[0].elapsed_time(ev[1]) * 1e3 / n), flush=True)
    import os
    os.environ['SMM_DEBUG_FLAGS'] = '1'
    ev[0].record(st)
    for _ in range(n):
        o2 = ops.viterbi(batch, elp, trans, init, lens, want_spans=True, want_labels=True)
    ev[1].record(st)
    torch.cuda.synchronize()
    os.environ.pop('SMM_DEBUG_FLAGS')
    print('     forward pass only %.1f us' % (ev[0].elapsed_time(ev[1]) * 1e3 / n), flush=True)
