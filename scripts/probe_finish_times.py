"""Diagnostic: which videos finish last in the DP kernel of the cfg3 corpus?  (libsmmdp_end.so, scripts/build_prof_end.sh)"""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
import action_segmentation_amd as pkg
from action_segmentation_amd import _lib, ops, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libsmmdp_end.so')
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
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
for conf in sys.argv[2:]:
    p, tr = conf.split(',')
    os.environ['SMM_PAIRS'] = p; os.environ['SMM_TRIPLES'] = tr
    for _ in range(2):
        elp, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
        torch.cuda.synchronize()
        out = ops.viterbi(pc.batch, elp, t['trans'], t['init'], t['len'], endpen=pc.endpen, class_map=t['class_map'], want_spans=False)
        torch.cuda.synchronize()
    end = out['best'].cpu().numpy() / 100.0     # us
    end -= end.min()
    if 'prev' in globals():
        d = end - prev
        j = np.argsort(-d)[:10]
        print('   later than in the previous configuration (us, end us, T, states):',
              [(int(d[i]), int(end[i]), int(pc.lengths[i]), int(pc.n_states[pc.group[i]])) for i in j])
        cost = np.array([pc.lengths[i] * pc.n_states[pc.group[i]] for i in range(len(end))])
        r = np.argsort(-cost)
        print('   the 6 most expensive videos: (end us now, end us before, T, states):',
              [(int(end[i]), int(prev[i]), int(pc.lengths[i]), int(pc.n_states[pc.group[i]])) for i in r[:6]])
    prev = end.copy()
    idx = np.argsort(-end)[:8]
    print('gangs', p, 'triples', tr, ': last finishers (us after the first, T, states, segments):',
          [(int(end[i]), int(pc.lengths[i]), int(pc.n_states[pc.group[i]]), int(out['n_segs'][i])) for i in idx], flush=True)
