"""Emission kernel time on the cfg3 corpus: v2 (LDS-staged lines) vs v1 (SMM_EMISSION_V1=1), HIP events."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
cfg = synth.CONFIGS[wl]
dev = torch.device('cuda:0')
data = synth.SynthDatasplit(wl, seed=2, device=dev)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
model = SemiMarkovModel.from_args(args, data)
pc = model.prepare(data)
t = pc.tables
ref = None
for mode in ('v1', 'v2', 'v1'):
    if mode == 'v2':
        os.environ['SMM_EMISSION_V2'] = '1'
    else:
        os.environ.pop('SMM_EMISSION_V2', None)
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        elp, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    if ref is None:
        ref = elp.clone()
    same = bool(torch.equal(elp, ref))
    gb = pc.n_frames * (4 * cfg['d'] + 8 * pc.c_max) / 1e9
    print('%s %s: %.3f ms (min of %s) = %.2f TB/s algorithmic   bit-identical to first: %s' % (
        wl, mode, min(ts[1:]), ['%.3f' % x for x in ts], gb / min(ts[1:]), same), flush=True)
