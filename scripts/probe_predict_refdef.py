"""Where the wall time of SemiMarkovModel.predict(fused=False) -- the reference's call pattern, one viterbi() per batch of five
videos -- goes at the reference's default shapes (synth 'refdef': 18 batches, K = 20, T ~ 300)."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
import bench
torch.set_num_threads(8)      # as bench.py and cli.py do: the box shows 256 CPUs, and a 256-thread pool only adds latency to small host ops
from action_segmentation_amd import synth
a = bench.parse(['--workload', 'refdef'])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS['refdef']
data = synth.SynthDatasplit('refdef', seed=2, device=dev)
a.fit_videos = 5
args, model = bench.fit_model(a, cfg, data, dev, None, 1)
for _ in range(3):
    model.predict(data, fused=False)
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); model.predict(data, fused=False); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print('predict(fused=False): min %.3f ms, median %.3f ms for %d batches' % (min(ts), float(np.median(ts)), cfg['n_tasks']))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    model.predict(data, fused=False)
pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(45)
pstats.Stats(pr).sort_stats('tottime').print_stats(30)
