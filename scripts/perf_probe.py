"""Quick timing probe of the DP kernel on random scores (not the bench; see bench.py)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from action_segmentation_amd import ops

def probe(b, T, C, K, reps=3):
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    elp = (torch.randn(b * T, C, generator=g, dtype=torch.float64) * 3 - 1).to(dev)
    trans = torch.log_softmax(torch.randn(1, C, C, generator=g, dtype=torch.float64), 1).to(dev)
    init = torch.log_softmax(torch.randn(1, C, generator=g, dtype=torch.float64), 1).to(dev)
    k = torch.arange(K, dtype=torch.float64)[:, None]
    rate = torch.rand(C, dtype=torch.float64, generator=g) * 200 + 20
    lens = (k * rate.log() - rate - torch.lgamma(k + 1))[None].contiguous().to(dev)
    batch = ops.Batch([T] * b, [C], K, t_max=T, total_frames=b * T)
    for _ in range(2):
        out = ops.viterbi(batch, elp, trans, init, lens)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = ops.viterbi(batch, elp, trans, init, lens); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = min(ts)
    cells = b * T * ((min(K, T) - 1) * C + C * C)
    print(f"b={b} T={T} C={C} K={K}: {ms:.3f} ms  {b*T/ms*1e3/1e6:.1f} Mframes/s  {cells/ms*1e3/1e12:.3f} Tcell/s  "
          f"cycles/frame/video@2.4GHz={ms*1e-3*2.4e9/T:.0f} segs={out['n_segs'].float().mean().item():.0f}", flush=True)

if __name__ == '__main__':
    probe(64, 2048, 16, 256)
    probe(256, 2048, 16, 256)
    probe(64, 2048, 16, 20)
    probe(256, 10000, 20, 1024, reps=2)
    probe(256, 10000, 16, 1024, reps=2)
    probe(512, 4000, 16, 64, reps=2)
