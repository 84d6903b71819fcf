"""Where does SemiMarkovModel.predict spend its wall time on cfg3?"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
cfg = synth.CONFIGS['cfg3']
dev = torch.device('cuda:0')
data = synth.SynthDatasplit('cfg3', seed=2, device=dev)
fit_args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
fitted = SemiMarkovModel.from_args(fit_args, data)
fitted.fit(data.subset(6), use_labels=True)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
model = SemiMarkovModel.from_args(args, data)
model.model.load_state_dict(fitted.model.state_dict(), strict=False)
model.model.to(dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = model.predict(data)
    torch.cuda.synchronize(); print('predict call %d: %.2f ms' % (i, (time.perf_counter() - t0) * 1e3), flush=True)
pr = cProfile.Profile(); pr.enable()
p = model.predict(data); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
