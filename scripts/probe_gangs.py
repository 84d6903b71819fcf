"""DP kernel time of the cfg3 corpus under forced gang configurations (SMM_PAIRS / SMM_TRIPLES; -1 = the host's choice)."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = synth.CONFIGS['cfg3']
dev = torch.device('cuda:0')
data = synth.SynthDatasplit('cfg3', seed=seed, device=dev)
fit_args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
fitted = SemiMarkovModel.from_args(fit_args, data)
fitted.fit(data.subset(6), use_labels=True)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
model = SemiMarkovModel.from_args(args, data)
model.model.load_state_dict(fitted.model.state_dict(), strict=False)
model.model.to(dev)
pc = model.prepare(data)
t = pc.tables
elp, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
torch.cuda.synchronize()
ref = None
for conf in sys.argv[2:]:
    p, tr = conf.split(',')
    for k, v in (('SMM_PAIRS', p), ('SMM_TRIPLES', tr)):
        if v == '-1':
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.viterbi(pc.batch, elp, t['trans'], t['init'], t['len'], endpen=pc.endpen, class_map=t['class_map'], want_spans=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ops.check_decoded(pc.batch, out)
    lab = out['labels'].cpu()
    if ref is None:
        ref = lab
    print('gangs %s triples %s: DP %.3f ms (min of %s)  same labels: %s' % (p, tr, min(ts[1:]), ['%.2f' % x for x in ts], bool(torch.equal(lab, ref))), flush=True)
