"""ns per frame of 23-state videos: gangs (pairs) vs the 12-wave single-CU configuration (SMM_PAIRS=0)."""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from action_segmentation_amd import ops
import test_gpu_viterbi as TV
dev = torch.device('cuda:0')
t = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
for c in (23, 22, 21):
    for b, T in ((64, 4096),):
        p = TV.make_problem(7, b, T, c, 1024)
        p['lengths'][:] = T
        batch = ops.Batch(p['lengths'], [c], 1024, c_max=c, t_max=T, total_frames=b * T)
        args = (t(p['elp'].reshape(b * T, c)), t(p['trans'][None]), t(p['init'][None]), t(p['lens'][None]))
        for mode in ('gangs', 'wide'):
            if mode == 'wide':
                os.environ['SMM_PAIRS'] = '0'
            else:
                os.environ.pop('SMM_PAIRS', None)
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); out = ops.viterbi(batch, *args, want_spans=False); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            segs = float(out['n_segs'].float().mean())
            print('C=%d b=%d T=%d %s: %.3f ms -> %.0f ns/frame (%.0f segments per video)' % (c, b, T, mode, min(ts[1:]), min(ts[1:]) * 1e6 / T, segs), flush=True)
