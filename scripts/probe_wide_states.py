"""ns per frame of 24..28-state videos at K = 1024: triples against the spilling 16-wave configuration (SMM_PAIRS=0)."""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
os.environ['SMM_DEBUG_FLAGS'] = '1'
from action_segmentation_amd import ops
import test_gpu_viterbi as TV
dev = torch.device('cuda:0')
t = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
for c in (24, 28, 30, 32, 23):
    b, T = 32, 4096
    p = TV.make_problem(7, b, T, c, 1024)
    p['lengths'][:] = T
    batch = ops.Batch(p['lengths'], [c], 1024, c_max=c, t_max=T, total_frames=b * T)
    args = (t(p['elp'].reshape(b * T, c)), t(p['trans'][None]), t(p['init'][None]), t(p['lens'][None]))
    for mode in ('gangs', 'no gangs'):
        if mode == 'no gangs':
            os.environ['SMM_PAIRS'] = '0'
        else:
            os.environ.pop('SMM_PAIRS', None)
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = ops.viterbi(batch, *args, want_spans=False); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print('C=%d, %d videos of %d frames, %s: %.3f ms -> %.0f ns/frame (forward pass)' % (c, b, T, mode, min(ts[1:]), min(ts[1:]) * 1e6 / T), flush=True)
