import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import test_gpu_model as TG
from action_segmentation_amd import ops
orig = ops.logz
def spy(batch, elp, *a, **k):
    z = orig(batch, elp, *a, **k)
    torch.cuda.synchronize()
    print('logz C', batch.c_max, 'b', batch.b, 'T', batch.lengths[:6], 'z', z.cpu().numpy()[:6], 'elp nan', int(torch.isnan(elp).sum()), 'elp inf', int(torch.isinf(elp).sum()), flush=True)
    return z
ops.logz = spy
try:
    TG.test_packed_log_likelihood_equals_per_batch(False)
    print('PASS')
except AssertionError as e:
    print('FAIL', str(e)[:300])
