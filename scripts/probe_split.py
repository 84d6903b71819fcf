"""Development aid: the DP launches of smm_decode_f32 on the bench's cfg3 corpus under SMM_SPLIT_DEBUG (1: the second part's DP
is not launched, 2: its emission is not; results incomplete) -- how long does the critical videos' launch take with and
without the second stream's work beside it?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from action_segmentation_amd import ops, synth

a = bench.parse(['--workload', 'cfg3'])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS[a.workload]
data = synth.SynthDatasplit(a.workload, seed=a.seed, device=dev)
_, model = bench.fit_model(a, cfg, data, dev, None, 1)
pc = model.prepare(data)
t = pc.tables
def step():
    return ops.decode(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], t['trans'], t['init'], t['len'], cons=pc.cons,
                      endpen=pc.endpen, class_map=t['class_map'], want_spans=False, want_labels=True, labels_on_host=True)
for _ in range(3):
    step()
torch.cuda.synchronize()
ops.dp_timing_read(); ops.dp_timing(True)
import time
t0 = time.perf_counter()
host = 0.0
for _ in range(10):
    h0 = time.perf_counter(); step(); host += time.perf_counter() - h0
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print('host time of the decode call (returns with the work queued): %.3f ms' % (host / 10 * 1e3))
ops.dp_timing(False)
ms = ops.dp_timing_read()
n = len(ms) // 10
print('SMM_SPLIT_DEBUG=%s: step %.3f ms; DP launches per step %d; mean duration by launch order: %s'
      % (os.environ.get('SMM_SPLIT_DEBUG', '0'), dt * 1e3, n, ' '.join('%.3f' % np.mean(ms[i::n]) for i in range(n))))
