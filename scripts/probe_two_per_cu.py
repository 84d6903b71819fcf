"""Development probe (round 5): a FOUR-wave BAND workgroup (chain, mover, two pushers with up to 8 states each; <= 16 states) fits a
CU twice.  libsmmdp_w4.so / libsmmdp_w8.so = scripts/build_variants.sh with -DSMM_DEV_BAND_ONE ... (see profiles/round5_two_per_cu.txt).
Bit-exactness against the C twin first, then the DP kernel's time on CrossTask-like lattices by batch size."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts'); sys.path.insert(0, 'tests')
import numpy as np, torch
import prof_band
import test_gpu_viterbi as tv
from action_segmentation_amd import _lib, ops

os.environ['SMM_CHUNK'] = '0'
for tag in sys.argv[1:]:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libsmmdp_%s.so' % tag)
    _lib._lib = None
    _lib.reload_env()
    ops._ws_cache.clear()
    for lengths, c, k in (([2300, 1029], 16, 1024), ([3000, 700, 64], 11, 1024), ([1500, 1400], 5, 700)):
        p = tv.structured_problem(7, lengths, c, k)
        out = tv.run_gpu(p)
        spans, v = tv.run_oracle(p)
        tv.check(p, out, spans, v)
    p = tv.make_problem(9, 2, 1300, 13, 1024, integer=True)
    out = tv.run_gpu(p); spans, v = tv.run_oracle(p); tv.check(p, out, spans, v)
    print('==', tag, 'bit-exact against the twin on 4 lattices'); sys.stdout.flush()
    for rep in range(2):
        for b, t, c in ((64, 4096, 16), (64, 4096, 11), (256, 4096, 11), (512, 4096, 11), (512, 4096, 16), (1024, 2048, 13)):
            prof_band.run(b, t, c, 1024, 'libsmmdp_%s.so' % tag)
