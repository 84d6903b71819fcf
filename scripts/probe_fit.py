"""smm_fit_stats_f64 on the cfg3 corpus: time per launch (HIP events), optionally under SMM_FIT_GRID."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
cfg = synth.CONFIGS['cfg3']
dev = torch.device('cuda:0')
data = synth.SynthDatasplit('cfg3', seed=2, device=dev)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
model = SemiMarkovModel.from_args(args, data)
pc = model.prepare(data)
gt = torch.empty(pc.batch.total_frames, dtype=torch.int64, device=dev)
for nm, tk, o, n in zip(pc.video_names, pc.task_names, pc.frame_offset, pc.lengths):
    gt[o:o + n] = data._videos[(tk, nm)]['gt_single'].to(dev)
nbytes = pc.n_frames * (4 * cfg['d'] + 16)
for g in sys.argv[1:] or ['768']:
    os.environ['SMM_FIT_GRID'] = g
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.fit_stats(pc.x, gt, pc.lengths, pc.frame_offset, data.corpus.n_classes, cfg['max_k'])
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print('grid %s: %.3f ms (min of %s) = %.2f TB/s' % (g, min(ts[1:]), ['%.3f' % t for t in ts], nbytes / min(ts[1:]) / 1e9), flush=True)
