"""predict(fused=False) -- the reference's call pattern -- on cfg3 (72 batches of five long videos) by pipeline depth."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
torch.set_num_threads(8)      # as bench.py and cli.py do: the box shows 256 CPUs, and a 256-thread pool only adds latency to small host ops
from action_segmentation_amd import synth
a = bench.parse(['--workload', 'cfg3'])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS['cfg3']
data = synth.SynthDatasplit('cfg3', seed=2, device=dev)
args, model = bench.fit_model(a, cfg, data, dev, None, 1)
ref = model.predict(data)
for depth in (1, 2, 4, 8, 16):
    model.args.decode_depth = depth
    model.predict(data, fused=False)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p = model.predict(data, fused=False); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    same = all(np.array_equal(p[k], ref[k]) for k in ref)
    print('depth %2d: %.1f ms (min of 3) for %d batches; equal to the fused decode: %s' % (depth, min(ts), len(data._videos) // 5, same))
# where the host's time per batch goes at the shipped depth (cProfile, one pass)
import cProfile, pstats, io
model.args.decode_depth = 8
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
model.predict(data, fused=False)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:6000])
