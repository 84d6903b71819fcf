"""Host / device time split of one decode step on cfg3 (emission + DP), metadata through kernel arguments vs hipMemcpyAsync."""
import sys, os, time
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
pc = model.prepare(data)
t = pc.tables
st = torch.cuda.current_stream()
def step(times):
    t0 = time.perf_counter()
    elp, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
    t1 = time.perf_counter()
    out = ops.viterbi(pc.batch, elp, t['trans'], t['init'], t['len'], endpen=pc.endpen, class_map=t['class_map'],
                      want_spans=False, want_labels=True, labels_on_host=True)
    t2 = time.perf_counter()
    st.synchronize()
    t3 = time.perf_counter()
    times.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
for mode in ('kernarg', 'memcpy', 'kernarg'):
    if mode == 'memcpy':
        os.environ['SMM_UPLOAD_MEMCPY'] = '1'
    else:
        os.environ.pop('SMM_UPLOAD_MEMCPY', None)
    times = []
    for _ in range(12):
        step(times)
    a = np.array(times[2:]) * 1e3
    print('%s: host emission call %.3f ms, host viterbi call %.3f ms, wait %.3f ms, step %.3f ms (min %.3f)' % (
        mode, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), a[:, 3].min()), flush=True)
