"""Host time of one decode call on a resident corpus: ops.decode (python marshalling + the library's host side: plan look-up,
kernel launches) against the library call alone with the ctypes arguments built once."""
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
torch.set_num_threads(8)
from action_segmentation_amd import synth, ops, _lib
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
a = bench.parse(['--workload', wl])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS[wl]
data = synth.SynthDatasplit(wl, seed=2, device=dev)
args, model = bench.fit_model(a, cfg, data, dev, None, 1)
pc = model.prepare(data)
t = pc.tables
kw = dict(cons=pc.cons, endpen=pc.endpen, class_map=t['class_map'], want_spans=False, want_labels=True, labels_on_host=True)
call = lambda: ops.decode(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], t['trans'], t['init'], t['len'], **kw)
for _ in range(3):
    call(); torch.cuda.synchronize()
host, wall = [], []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); call(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e6); wall.append((t2 - t0) * 1e3)
print('%s ops.decode: host %.1f us (median %.1f), wall with sync %.3f ms' % (wl, min(host), float(np.median(host)), min(wall)))
# the same call through a recording shim: what does the library call alone cost?
lib = _lib.load()
rec = {}
orig = lib.smm_decode_f32
class Shim:
    def __call__(self, *args):
        rec['args'] = args
        return orig(*args)
lib.smm_decode_f32 = Shim()
out = call(); torch.cuda.synchronize()
lib.smm_decode_f32 = orig
args = rec['args']
host2 = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); rc = orig(*args); t1 = time.perf_counter(); torch.cuda.synchronize()
    host2.append((t1 - t0) * 1e6)
print('%s smm_decode_f32 with prebuilt arguments: host %.1f us (median %.1f)' % (wl, min(host2), float(np.median(host2))))
