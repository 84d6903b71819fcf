"""Diagnostic: busy / barrier-wait cycles per block of every wave of workgroup 0 (libsmmdp_prof<R>.so, built by
scripts/build_prof.sh with -DSMM_PROFILE)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import action_segmentation_amd as pkg
from action_segmentation_amd import _lib, ops


def run(b, T, C, K):
    r = 1
    while 64 * r < min(K, T):
        r *= 2
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get('SMM_PROF_LIB', 'libsmmdp_prof%d.so' % r))
    _lib._lib = None
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    elp = (torch.randn(b * T, C, generator=g, dtype=torch.float64) * 3 - 1).to(dev)
    trans = torch.log_softmax(torch.randn(1, C, C, generator=g, dtype=torch.float64), 1).to(dev)
    init = torch.log_softmax(torch.randn(1, C, generator=g, dtype=torch.float64), 1).to(dev)
    k = torch.arange(K, dtype=torch.float64)[:, None]
    rate = torch.rand(C, dtype=torch.float64, generator=g) * 200 + 20
    lens = (k * rate.log() - rate - torch.lgamma(k + 1))[None].contiguous().to(dev)
    batch = ops.Batch([T] * b, [C], K, t_max=T, total_frames=b * T)
    ops._ws_cache.clear()
    ops.viterbi(batch, elp, trans, init, lens); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.viterbi(batch, elp, trans, init, lens); e1.record(); torch.cuda.synchronize()
    ws = list(ops._ws_cache.values())[0]
    import ctypes
    o_err = _lib.load().smm_error_word_offset(ctypes.byref(batch.shape))
    pp = ws[o_err: o_err + 512].cpu().numpy().view(np.uint64).astype(np.float64)
    nblk = pp[7]
    print(f"b={b} T={T} C={C} K={K}: {e0.elapsed_time(e1):.3f} ms, {nblk:.0f} blocks; cycles per block (busy / in barrier) by wave:")
    for w in range(16):
        if pp[8 + w] or pp[24 + w]:
            print("   wave %2d  busy %6.0f  barrier %6.0f" % (w, pp[8 + w] / nblk, pp[24 + w] / nblk))
    if pp[43]:
        print("   back-trace of workgroup 0: %d segments; cycles per segment: phase A %.0f, phase B %.0f, labels %.0f"
              % (pp[43], pp[40] / pp[43], pp[41] / pp[43], pp[42] / pp[43]))
    sys.stdout.flush()


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'load':
        # leader of gang 0 on a full GPU: pair vs triple
        for b, t in ((250, 4096), (8, 14000)):
            for tr in (0, 8):
                os.environ['SMM_PAIRS'] = '8'; os.environ['SMM_TRIPLES'] = str(tr)
                print('b', b, 'SMM_PAIRS 8 SMM_TRIPLES', tr)
                run(b, t, 21, 1024)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'pair':
        for c in (21, 19, 17, 15, 13, 11):
            for n in (0, 64):
                os.environ['SMM_PAIRS'] = str(n)
                print('SMM_PAIRS', n)
                run(64, 4096, c, 1024)
        sys.exit(0)
    run(64, 2048, 16, 256)
    run(64, 4096, 21, 1024)
    run(64, 4096, 20, 1024)
    run(64, 4096, 16, 1024)
