"""Timing probe of the log-partition kernel (not the bench)."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
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
        z = ops.logz(batch, elp, trans, init, lens)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); z = ops.logz(batch, elp, trans, init, lens); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"logz b={b} T={T} C={C} K={K}: {min(ts):.3f} ms  ns/frame/video={min(ts)*1e6/T:.0f}", flush=True)
for c, k in ((14, 1024), (15, 1024), (21, 1024), (23, 1024), (28, 1024), (16, 64), (16, 256), (12, 20), (21, 512)):
    probe(64, 2048, c, k)
