"""Does a decode (emission + DP + recovery launches + metadata uploads) capture into a hipGraph and replay bit-exactly?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
cfg = synth.CONFIGS[wl]
dev = torch.device('cuda:0')
data = synth.SynthDatasplit(wl, seed=2, device=dev)
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
labels_host = ops._labels_on_host(B, dev)

def step():
    elp, _ = ops.emission(B, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
    return ops.viterbi(B, elp, t['trans'], t['init'], t['len'], endpen=pc.endpen, class_map=t['class_map'],
                       want_spans=False, want_labels=True, labels_out=labels_host)

for _ in range(3):
    out = step()
torch.cuda.synchronize()
ref = labels_host.clone()
ts = []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('%s eager step: min %.3f ms median %.3f ms' % (wl, min(ts), float(np.median(ts))), flush=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gout = step()
torch.cuda.synchronize()
print('captured', flush=True)
labels_host.fill_(-5)
g.replay(); torch.cuda.synchronize()
print('replay identical to eager:', bool((labels_host == ref).all()), 'err', gout['_err'].tolist(), flush=True)
ts = []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('%s graph replay step: min %.3f ms median %.3f ms' % (wl, min(ts), float(np.median(ts))), flush=True)
# new features in the same buffers: shuffle the frames of the corpus in place, decode eagerly and by replay
perm = torch.randperm(pc.x.size(0), device=dev)
pc.x.copy_(pc.x[perm])
step(); torch.cuda.synchronize(); ref2 = labels_host.clone()
labels_host.fill_(-5)
g.replay(); torch.cuda.synchronize()
print('after changing the features in place: replay identical to eager:', bool((labels_host == ref2).all()),
      ' differs from the first decode:', bool((ref2 != ref).any()), flush=True)
