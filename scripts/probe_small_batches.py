"""Latency of the reference-style per-batch API on the reference's default shapes (batch of 5 videos, T ~ 300, K = 20)."""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
from action_segmentation_amd.batching import make_data_loader
cfg = dict(n_tasks=4, videos_per_task=10, steps=(4, 8), t_lognormal=(300, 0.3, 100, 600), max_k=20, d=200, chain=True, rate=(3, 12), batch_size=5)
dev = torch.device('cuda:0')
data = synth.SynthDatasplit(cfg, seed=5, device=dev)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=5)
model = SemiMarkovModel.from_args(args, data)
model.fit(data.subset(6), use_labels=True)
model.model.to(dev)
batches = list(make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=5))
m = model.model
def per_batch():
    for b in batches:
        m.viterbi(b['features'].to(dev), b['lengths'], b['task_indices'], add_eos=True)
per_batch(); per_batch()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): per_batch()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
frames = sum(int(b['lengths'].sum()) for b in batches)
print('reference-style batches: %d batches, %d frames: %.3f ms per batch, %.2f M frames/s' % (len(batches), frames, dt / len(batches) * 1e3, frames / dt / 1e6))
def packed():
    return model.predict(data)
packed(); packed()
t0 = time.perf_counter()
for _ in range(5): packed()
dt = (time.perf_counter() - t0) / 5
print('predict (one packed launch incl. collation): %.3f ms, %.2f M frames/s' % (dt * 1e3, frames / dt / 1e6))
pc = model.prepare(data)
model.predict_packed(pc); 
t0 = time.perf_counter()
for _ in range(20): model.predict_packed(pc)
dt = (time.perf_counter() - t0) / 20
print('predict_packed (resident corpus): %.3f ms, %.2f M frames/s' % (dt * 1e3, frames / dt / 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): per_batch()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(22)
