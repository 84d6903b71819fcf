"""smm_emission_f64 by feature dimension and class-set size (what's weak 3: only D = 200 was ever timed; at D = 300 -- the
reference's i3d + resnet + audio setting, main.py:283-286 -- a > 16-state class set's weight table no longer leaves room for two
workgroups per CU).  One-class-set corpora of ~0.5 G floats of x; prints ms and algorithmic TB/s (4 D + 8 C bytes per frame).
usage: probe_emission_d.py [--d=200,257,300,400] [--c=11,19,23,28]"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from action_segmentation_amd import ops
ds, cs = [200, 256, 300, 400], [11, 19, 23, 28]
for t in sys.argv[1:]:
    if t.startswith('--d='): ds = [int(v) for v in t[4:].split(',')]
    if t.startswith('--c='): cs = [int(v) for v in t[4:].split(',')]
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(3)
for d in ds:
    b, T = 512, int(4800 * 200 / d) // 16 * 16
    x = torch.randn(b * T, d, generator=g, dtype=torch.float32).to(dev)
    for c in cs:
        lengths = np.full(b, T, dtype=np.int64)
        offs = np.arange(b, dtype=np.int64) * T
        batch = ops.Batch(lengths, [c], 1024, t_max=T, frame_offset=offs, total_frames=b * T, d=d)
        w = torch.randn(1, d, c, dtype=torch.float64, generator=g).to(dev)
        cst = torch.randn(1, c, dtype=torch.float64, generator=g).to(dev)
        iv = (0.5 + torch.rand(d, dtype=torch.float64, generator=g)).to(dev)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); elp, _ = ops.emission(batch, x, w, cst, iv); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        # spot check against torch on 4096 frames: -0.5 sum_d iv (x - mu)^2-shaped scores are w, cst, iv here: elp = x w - 0.5 x^2 iv + cst
        xs = x[:4096].double()
        ref = xs @ w[0] - 0.5 * (xs * xs) @ iv[:, None] + cst[0][None]
        err = float(((elp[:4096, :c] - ref).abs() / ref.abs().clamp(min=1.0)).max())
        gb = b * T * (4 * d + 8 * c) / 1e9
        print('D=%3d C=%2d  %d x %d frames  %.3f ms (median %.3f) = %.2f TB/s algorithmic; max rel err vs torch fp64 on 4096 frames %.1e' % (
            d, c, b, T, min(ts[2:]), float(np.median(ts[2:])), gb / min(ts[2:]), err), flush=True)
    del x
