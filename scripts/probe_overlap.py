"""Experiment: cfg3 decode step as TWO sub-batches on two streams (expensive videos first; the rest's emission kernel
runs beside the first DP launch) against the single-launch step of bench.py."""
import sys, os, time
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
B = pc.batch
main = torch.cuda.current_stream()
labels_host = ops._labels_on_host(B, dev)
elp = torch.empty((B.total_frames, B.c_max), dtype=torch.float64, device=dev)

def single():
    ops.emission(B, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons, out64=elp)
    out = ops.viterbi(B, elp, t['trans'], t['init'], t['len'], endpen=pc.endpen, class_map=t['class_map'],
                      want_spans=False, want_labels=True, labels_out=labels_host)
    main.synchronize()
    return out

def sub(idx):
    idx = np.asarray(idx)
    b = ops.Batch(B.lengths[idx], B.n_states, B.k_rows, c_max=B.c_max, frame_offset=B.frame_offset[idx],
                  group=B.group[idx], kp=None if B.kp is None else B.kp[idx], d=B.d, t_max=B.t_max,
                  total_frames=B.total_frames)
    ep = None if pc.endpen is None else pc.endpen[torch.as_tensor(idx, device=dev)].contiguous()
    return b, ep

cost = B.lengths * B.n_states[B.group]
order = np.argsort(-cost, kind='stable')
big = B.n_states[B.group] > 21
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ref = single()['labels'].clone()
for n_first in (40, 72, 104, 136):
    first = [i for i in order if big[i]] + [i for i in order if not big[i]][:max(0, n_first - int(big.sum()))]
    rest = [i for i in range(B.b) if i not in set(first)]
    (bA, eA), (bB, eB) = sub(first), sub(rest)
    def overlapped():
        e = torch.cuda.Event(); e.record(main)
        s1.wait_event(e); s2.wait_event(e)
        with torch.cuda.stream(s1):
            ops.emission(bA, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons, out64=elp)
            oA = ops.viterbi(bA, elp, t['trans'], t['init'], t['len'], endpen=eA, class_map=t['class_map'],
                             want_spans=False, want_labels=True, labels_out=labels_host)
        with torch.cuda.stream(s2):
            ops.emission(bB, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons, out64=elp)
            oB = ops.viterbi(bB, elp, t['trans'], t['init'], t['len'], endpen=eB, class_map=t['class_map'],
                             want_spans=False, want_labels=True, labels_out=labels_host)
        s1.synchronize(); s2.synchronize()
        return oA, oB
    for fn, name in ((single, 'single launch'), (overlapped, 'two streams, %d videos first' % len(first))):
        ts = []
        for _ in range(8):
            labels_host.fill_(-7)
            torch.cuda.synchronize(); t0 = time.perf_counter(); o = fn(); ts.append((time.perf_counter() - t0) * 1e3)
        same = bool((labels_host == ref).all())
        errs = [ops.error_words(None, x) for x in (o if isinstance(o, tuple) else (o,))]
        print('seed %d  %-34s step %.3f ms (min), %.3f (median)  labels identical: %s  err %s' % (
            seed, name, min(ts[2:]), float(np.median(ts[2:])), same, errs), flush=True)
