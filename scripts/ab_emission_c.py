"""Development aid: smm_emission_f64 of library variants on a one-class-set corpus (512 videos x 4800 frames, D = 200) by the class
set's size: where the 17..24-state bodies of smm_emission_stream_kernel lose their time.  usage: ab_emission_c.py tag... [--c 11,19,23]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from action_segmentation_amd import _lib, ops
cs = [11, 15, 19, 23]
tags = [t for t in sys.argv[1:] if not t.startswith('--c')]
for t in sys.argv[1:]:
    if t.startswith('--c='):
        cs = [int(v) for v in t[4:].split(',')]
dev = torch.device('cuda:0')
b, T, d = 512, 4800, 200
g = torch.Generator(device='cpu').manual_seed(3)
x = torch.randn(b * T, d, generator=g, dtype=torch.float32).to(dev)
for c in cs:
    lengths = np.full(b, T, dtype=np.int64)
    offs = np.arange(b, dtype=np.int64) * T
    batch = ops.Batch(lengths, [c], 1024, t_max=T, frame_offset=offs, total_frames=b * T, d=d)
    w = torch.randn(1, d, c, dtype=torch.float64, generator=g).to(dev)
    cst = torch.randn(1, c, dtype=torch.float64, generator=g).to(dev)
    iv = (0.5 + torch.rand(d, dtype=torch.float64, generator=g)).to(dev)
    ref = None
    for tag in tags:
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libsmmdp_%s.so' % tag if tag != 'shipped' else 'libsmmdp.so')
        _lib._lib = None
        ops._ws_cache.clear()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); elp, _ = ops.emission(batch, x, w, cst, iv); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        if ref is None:
            ref = elp.clone()
        gb = b * T * (4 * d + 8 * c) / 1e9
        print('C=%2d %-8s %.3f ms (median %.3f) = %.2f TB/s algorithmic; equal to the first: %s' % (
            c, tag, min(ts[2:]), float(np.median(ts[2:])), gb / min(ts[2:]), bool(torch.equal(elp, ref))), flush=True)
