"""Diagnostic: per-phase cycle shares of one frame step (libsmmdp_prof.so, built by scripts/build_prof.sh)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import action_segmentation_amd as pkg
from action_segmentation_amd import _lib, ops
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libsmmdp_prof.so')

def run(b, T, C, K):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    elp = (torch.randn(b * T, C, generator=g, dtype=torch.float64) * 3 - 1).to(dev)
    trans = torch.log_softmax(torch.randn(1, C, C, generator=g, dtype=torch.float64), 1).to(dev)
    init = torch.log_softmax(torch.randn(1, C, generator=g, dtype=torch.float64), 1).to(dev)
    k = torch.arange(K, dtype=torch.float64)[:, None]
    rate = torch.rand(C, dtype=torch.float64, generator=g) * 200 + 20
    lens = (k * rate.log() - rate - torch.lgamma(k + 1))[None].contiguous().to(dev)
    batch = ops.Batch([T] * b, [C], K, t_max=T, total_frames=b * T)
    ops.viterbi(batch, elp, trans, init, lens); torch.cuda.synchronize()
    ws = list(ops._ws_cache.values())[0]
    # error block offset: mirrors make_plan() in smm_api.hip
    al = lambda x: (x + 255) // 256 * 256
    o_err = al(32 * b) + al(4 * b) + al(4)
    ch = ws[o_err + 64: o_err + 128].cpu().numpy().view(np.uint64).astype(np.float64)
    pu = ws[o_err + 192: o_err + 256].cpu().numpy().view(np.uint64).astype(np.float64)
    print(f"b={b} T={T} C={C} K={K}")
    print("  chain : gamma=%4.0f transition=%4.0f barrier=%4.0f total=%4.0f" % tuple(ch[:4] / ch[7]))
    print("  pusher: read_h=%4.0f pushes=%4.0f barrier=%4.0f total=%4.0f" % tuple(pu[:4] / pu[7]), flush=True)
    wv = ws[o_err + 320: o_err + 320 + 64].cpu().numpy().view(np.uint64).astype(np.float64)
    print("  busy cycles per frame by wave (release -> arrival):", " ".join("%4.0f" % (v / pu[7]) for v in wv))

if __name__ == '__main__':
    run(64, 2048, 16, 20)
    run(64, 4096, 20, 1024)
