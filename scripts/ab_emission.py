"""Development aid: smm_emission_f64 of libsmmdp_<tag>.so variants (scripts/build_variants.sh with SMM_VARIANT_SRC=smm_emission)
on a bench corpus: kernel time (HIP events) and whether the output equals the first variant's.  usage: ab_emission.py [workload] tag..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from action_segmentation_amd import _lib, ops, synth
wl = sys.argv[1] if sys.argv[1] in synth.CONFIGS else 'cfg3'
tags = [t for t in sys.argv[1:] if t not in synth.CONFIGS]
a = bench.parse(['--workload', wl])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS[wl]
data = synth.SynthDatasplit(wl, seed=a.seed, device=dev, scale=float(os.environ.get('SMM_SCALE', '1')))
_, model = bench.fit_model(a, cfg, data, dev, None, 1)
pc = model.prepare(data)
t = pc.tables
ref = None
for rep in range(2):
    for tag in tags:
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libsmmdp_%s.so' % tag if tag != 'shipped' else 'libsmmdp.so')
        _lib._lib = None
        ops._ws_cache.clear()
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); elp, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        if ref is None:
            ref = elp.clone()
        gb = pc.n_frames * (4 * cfg['d'] + 8 * pc.c_max) / 1e9
        print('%s %-10s %.3f ms (median %.3f) = %.2f TB/s algorithmic; equal to the first: %s' % (
            wl, tag, min(ts[2:]), float(np.median(ts[2:])), gb / min(ts[2:]), bool(torch.equal(elp, ref))), flush=True)
