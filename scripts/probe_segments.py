import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
seed = int(sys.argv[1])
cfg = synth.CONFIGS['cfg3']
dev = torch.device('cuda:0')
data = synth.SynthDatasplit('cfg3', seed=seed, device=dev)
fit_args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
fitted = SemiMarkovModel.from_args(fit_args, data)
fitted.fit(data.subset(2), use_labels=True)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
model = SemiMarkovModel.from_args(args, data)
model.model.load_state_dict(fitted.model.state_dict(), strict=False)
model.model.to(dev)
pc = model.prepare(data)
out = model.model.decode_packed(pc, want_spans=False, want_labels=True)
torch.cuda.synchronize()
ns = out['n_segs'].cpu().numpy()
idx = np.argsort(-ns)[:8]
print('seed', seed, 'top n_segs:', [(int(ns[i]), int(pc.lengths[i]), int(pc.n_states[pc.group[i]])) for i in idx], 'mean', ns.mean())
