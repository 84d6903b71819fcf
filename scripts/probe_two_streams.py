"""Experiment: split the packed corpus into [expensive videos | rest] and decode the two parts on two streams, so that
the second part's emission overlaps the first part's DP."""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
cfg = synth.CONFIGS['cfg3']
dev = torch.device('cuda:0')
data = synth.SynthDatasplit('cfg3', seed=1000, device=dev)
fit_args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
fitted = SemiMarkovModel.from_args(fit_args, data)
fitted.fit(data.subset(6), use_labels=True)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
model = SemiMarkovModel.from_args(args, data)
model.model.load_state_dict(fitted.model.state_dict(), strict=False)
model.model.to(dev)
pc = model.prepare(data)
t = pc.tables
B = pc.batch
cost = np.array([l * pc.n_states[g] for l, g in zip(B.lengths, B.group)])
order = np.argsort(-cost)

def sub(idx):
    idx = np.sort(idx)
    return ops.Batch(B.lengths[idx], B.n_states, B.k_rows, c_max=B.c_max, frame_offset=B.frame_offset[idx], group=B.group[idx],
                     kp=B.kp[idx], d=B.d, t_max=B.t_max, total_frames=B.total_frames), idx

elp = torch.empty((B.total_frames, B.c_max), dtype=torch.float64, device=dev)
labels = ops._labels_on_host(B, dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def step_one():
    ops.emission(B, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons, out64=elp)
    ops.viterbi(B, elp, t['trans'], t['init'], t['len'], endpen=pc.endpen, class_map=t['class_map'], want_spans=False, labels_out=labels)
    torch.cuda.synchronize()

def step_two(bA, iA, bB, iB, pairsB='0'):
    epA = None if pc.endpen is None else pc.endpen[iA].contiguous()
    epB = None if pc.endpen is None else pc.endpen[iB].contiguous()
    with torch.cuda.stream(s1):
        os.environ.pop('SMM_PAIRS', None)
        ops.emission(bA, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons, out64=elp)
        ops.viterbi(bA, elp, t['trans'], t['init'], t['len'], endpen=epA, class_map=t['class_map'], want_spans=False, labels_out=labels)
    with torch.cuda.stream(s2):
        os.environ['SMM_PAIRS'] = pairsB
        ops.emission(bB, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons, out64=elp)
        ops.viterbi(bB, elp, t['trans'], t['init'], t['len'], endpen=epB, class_map=t['class_map'], want_spans=False, labels_out=labels)
        os.environ.pop('SMM_PAIRS', None)
    torch.cuda.synchronize()

def timeit(fn, n=5):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3

step_one(); ref = labels.clone()
print('one launch pair: %.3f ms' % timeit(step_one))
for na in (24, 40, 64, 96):
    (bA, iA), (bB, iB) = sub(order[:na]), sub(order[na:])
    ms = timeit(lambda: step_two(bA, iA, bB, iB))
    step_two(bA, iA, bB, iB)
    print('split %3d | rest: %.3f ms  labels equal: %s' % (na, ms, bool((labels == ref).all())), flush=True)
