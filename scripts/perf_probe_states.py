import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops
def probe(b, T, C, K, cmax=None, reps=3):
    dev = torch.device('cuda:0')
    cm = cmax or C
    g = torch.Generator(device='cpu').manual_seed(0)
    elp = (torch.randn(b * T, cm, generator=g, dtype=torch.float64) * 3 - 1).to(dev)
    trans = torch.log_softmax(torch.randn(1, cm, cm, generator=g, dtype=torch.float64), 1).to(dev)
    init = torch.log_softmax(torch.randn(1, cm, generator=g, dtype=torch.float64), 1).to(dev)
    k = torch.arange(K, dtype=torch.float64)[:, None]
    rate = torch.rand(cm, dtype=torch.float64, generator=g) * 200 + 20
    lens = (k * rate.log() - rate - torch.lgamma(k + 1))[None].contiguous().to(dev)
    batch = ops.Batch([T] * b, [C], K, c_max=cm, t_max=T, total_frames=b * T)
    for _ in range(2):
        out = ops.viterbi(batch, elp, trans, init, lens)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = ops.viterbi(batch, elp, trans, init, lens); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = min(ts)
    print(f"b={b} T={T} C={C} K={K} pairs={os.environ.get('SMM_PAIRS','auto')}: {ms:.3f} ms  ns/frame/video={ms*1e6/T:.0f}", flush=True)
if os.environ.get('SMM_PAIRS'):
    for c in (13, 15, 16, 17, 19, 20, 21):       # forced gangs: SMM_PAIRS / SMM_TRIPLES from the environment
        probe(64, 4096, c, 1024)
    sys.exit(0)
for c in (21, 22, 23):
    for pairs in ('0', None):
        if pairs is None:
            os.environ.pop('SMM_PAIRS', None)
        else:
            os.environ['SMM_PAIRS'] = pairs
        probe(64, 4096, c, 1024)
