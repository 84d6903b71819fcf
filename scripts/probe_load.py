"""Does a one-CU video run slower when every CU of the chip is busy with one?  (forward pass only, no gangs)"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
os.environ['SMM_PAIRS'] = '0'; os.environ['SMM_DEBUG_FLAGS'] = '1'
from action_segmentation_amd import ops
import test_gpu_viterbi as TV
dev = torch.device('cuda:0')
t = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
for c in (21, 15):
    for b in (8, 64, 128, 256, 512):
        T = 4096
        p = TV.make_problem(7, b, T, c, 1024)
        p['lengths'][:] = T
        batch = ops.Batch(p['lengths'], [c], 1024, c_max=c, t_max=T, total_frames=b * T)
        args = (t(p['elp'].reshape(b * T, c)), t(p['trans'][None]), t(p['init'][None]), t(p['lens'][None]))
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = ops.viterbi(batch, *args, want_spans=False); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        rounds = (b + 255) // 256
        print('C=%d, %3d videos of %d frames: %.3f ms -> %.0f ns/frame per round of 256' % (c, b, T, min(ts[1:]), min(ts[1:]) * 1e6 / T / rounds), flush=True)
        del args, p
