"""Where the wall time of SemiMarkovModel.predict(data) (fused: one resident launch) goes beyond the decode itself, on cfg3."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
import bench
torch.set_num_threads(8)      # as bench.py and cli.py do: the box shows 256 CPUs, and a 256-thread pool only adds latency to small host ops
from action_segmentation_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
a = bench.parse(['--workload', wl])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS[wl]
data = synth.SynthDatasplit(wl, seed=2, device=dev)
args, model = bench.fit_model(a, cfg, data, dev, None, 1)
for _ in range(3):
    model.predict(data)
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); model.predict(data); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print('predict(fused): min %.3f ms, median %.3f ms' % (min(ts), float(np.median(ts))))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    model.predict(data)
pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(35)
pstats.Stats(pr).sort_stats('tottime').print_stats(20)
# the pieces of predict_packed, timed one by one
pc = model.prepare(data)
def t(f, n=10):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
ms, _ = t(lambda: model.prepare(data)); print('prepare (cache hit): %.3f ms' % ms)
ms, out = t(lambda: model.model.decode_packed(pc, want_spans=False, want_labels=True, labels_on_host=True)); print('decode_packed + sync: %.3f ms' % ms)
ms, lab = t(lambda: out['labels'].clone()); print('labels.clone(): %.3f ms (%d bytes)' % (ms, out['labels'].numel() * 8))
ms, _ = t(lambda: int(lab.max())); print('labels.max(): %.3f ms' % ms)
labels = lab.numpy()
ms, _ = t(lambda: {name: labels[off:off + tt] for name, off, tt in zip(pc.video_names, pc.frame_offset, pc.lengths)}); print('dict of views: %.3f ms' % ms)
# as bench.py's predict_end_to_end times it (gc.collect() in front, the previous result still alive) and the variations
import gc
for hold in (True, False):
    for collect in (True, False):
        ts = []
        preds = None
        for _ in range(8):
            if collect:
                gc.collect()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if hold:
                preds = model.predict(data)
            else:
                model.predict(data)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print('predict(fused), previous result held: %s, gc.collect() in front: %s: %s ms' % (hold, collect, ' '.join('%.2f' % x for x in ts)))
