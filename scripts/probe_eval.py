"""Where do evaluate_labels' ~3 ms on cfg3 go (0.24 ms of kernels)?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(8)
from action_segmentation_amd import evaluation, synth

dev = torch.device('cuda:0')
data = synth.SynthDatasplit('cfg3', seed=2, keep=set())          # structure only: labels and lengths
space = evaluation.LabelSpace.from_corpus(data.corpus, list(data._videos_by_task))
lengths, offsets, tasks, keys, gts = [], [], [], [], []
off = 0
for t, names in data._videos_by_task.items():
    for i, n in enumerate(names):
        g = data._videos[(t, n)]['gt_single']
        lengths.append(int(g.numel())); offsets.append(off); tasks.append(t); keys.append(i); gts.append(g); off += lengths[-1]
gt = torch.cat(gts).to(dev)
g = torch.Generator().manual_seed(0)
preds = []
for t, n_, gv in zip(tasks, lengths, gts):
    ids = torch.tensor(data.corpus._indices_by_task[t])
    p = gv.clone()
    flip = torch.rand(n_, generator=g) < 0.02
    p[flip] = ids[torch.randint(0, len(ids), (int(flip.sum()),), generator=g)]
    preds.append(p)
pred = torch.cat(preds).to(dev)
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = evaluation.evaluate_labels(pred, gt, lengths, offsets, tasks, space, optimal_assignment=False, seed=0, video_key=keys)
    torch.cuda.synchronize(); print('evaluate_labels %.3f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    evaluation.evaluate_labels(pred, gt, lengths, offsets, tasks, space, optimal_assignment=False, seed=0, video_key=keys)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
