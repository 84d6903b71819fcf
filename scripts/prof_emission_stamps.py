"""Development aid: s_memtime stamps of smm_emission_stream_kernel's first workgroups (a library variant built with -DSMM_EM_STAMP,
scripts/build_variants.sh with SMM_VARIANT_SRC=smm_emission): per wave and item, cycles from the item's start to: body entry
(metadata found), weights in LDS, loop entry, first data arrived, last step consumed, loads drained.  usage: prof_emission_stamps.py tag [workload]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from action_segmentation_amd import _lib, ops, synth
tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
a = bench.parse(['--workload', wl])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS[wl]
data = synth.SynthDatasplit(wl, seed=a.seed, device=dev)
_, model = bench.fit_model(a, cfg, data, dev, None, 1)
pc = model.prepare(data)
t = pc.tables
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libsmmdp_%s.so' % tag)
_lib._lib = None
ops._ws_cache.clear()
for _ in range(3):
    ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
WGS, WAVES, ITEMS = 8, 8, 12
buf = np.zeros((WGS, WAVES, ITEMS, 8), dtype=np.uint64)
rc = lib.smm_dev_em_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
names = ['found', 'filled', 'loop', 'first data', 'last step', 'drained']
for wg in range(WGS):                                             # (workgroups 0, 73, .. 511 of the grid)
    for wave in (0, 7):
        s = buf[wg, wave].astype(np.int64)
        t0 = s[0, 0]
        print('wg %d wave %d' % (wg, wave))
        for it in range(ITEMS):
            if s[it, 0] == 0:
                continue
            d = [(int(s[it, k]) - int(s[it, 0])) if s[it, k] else -1 for k in range(1, 7)]
            d[3] = -1                                                 # (slot 4 holds the segment's start on the 100 MHz clock)
            ghz = (int(s[it, 6]) - int(s[it, 0])) / max(1, int(s[it, 7]) - int(s[it, 4])) * 0.1
            print('   segment %2d starts at %8d: ' % (it, int(s[it, 0]) - int(t0)) + '  '.join('%s %6d' % (n, v) for n, v in zip(names, d)) +
                  '  shader clock %.2f GHz' % ghz)
