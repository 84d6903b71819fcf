"""CPU experiment: which fraction of the long segment-length range of the Viterbi DP survives an
exact bound test?  oracle/prune_probe.c does the counting; this script feeds it videos of the cfg3 seed-2 corpus
(CPU draw of the same generator: same shapes and margins, other random numbers than the GPU draw) decoded with
closed-form-fitted parameters, like bench.py does.

    python scripts/probe_prune.py [videos_per_task] [workload]      -> table on stdout (copy under profiles/)
"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from action_segmentation_amd import synth                      # noqa: E402
from action_segmentation_amd.semimarkov import SemiMarkovModel  # noqa: E402

so = os.path.join(ROOT, 'oracle', '_build', 'libprune_probe.so')
src = os.path.join(ROOT, 'oracle', 'prune_probe.c')
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-ffp-contract=off', '-shared', '-o', so, src, '-lm'])
lib = ctypes.CDLL(so)
P = ctypes.POINTER(ctypes.c_double)


def probe(elp, trans, init, len_scores, kl, tb, kb):
    t, c = elp.shape
    kp = min(len_scores.shape[0], t)
    out = np.zeros(8)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (elp, trans, init, len_scores[:kp])]
    rc = lib.smm_prune_probe(arrs[0].ctypes.data_as(P), t, c, arrs[1].ctypes.data_as(P), arrs[2].ctypes.data_as(P),
                             arrs[3].ctypes.data_as(P), kp, kl, tb, kb, out.ctypes.data_as(P))
    assert rc == 0
    return out


def band_probe(elp, trans, init, len_scores, bw):
    t, c = elp.shape
    kp = min(len_scores.shape[0], t)
    out = np.zeros(8)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (elp, trans, init, len_scores[:kp])]
    rc = lib.smm_band_probe(arrs[0].ctypes.data_as(P), t, c, arrs[1].ctypes.data_as(P), arrs[2].ctypes.data_as(P),
                            arrs[3].ctypes.data_as(P), kp, bw, out.ctypes.data_as(P))
    assert rc == 0
    return out


def band_probe2(elp, trans, init, len_scores, grp):
    t, c = elp.shape
    kp = min(len_scores.shape[0], t)
    out = np.zeros(8)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (elp, trans, init, len_scores[:kp])]
    rc = lib.smm_band_probe2(arrs[0].ctypes.data_as(P), t, c, arrs[1].ctypes.data_as(P), arrs[2].ctypes.data_as(P),
                             arrs[3].ctypes.data_as(P), kp, grp, out.ctypes.data_as(P))
    assert rc == 0
    return out


def main():
    per_task = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    wl = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
    cfg = synth.CONFIGS[wl]
    dry = synth.SynthDatasplit(wl, seed=2, keep=set())
    keep = {n for names in dry._videos_by_task.values() for n in names[:6]}
    data = synth.SynthDatasplit(wl, seed=2, keep=keep)
    args = synth.make_args(cfg['max_k'], cuda=False, batch_size=cfg['batch_size'])
    model = SemiMarkovModel.from_args(args, data)
    model.fit(data.subset(6), use_labels=True)
    m = model.model
    designs = [(128, 64, 64), (256, 128, 64), (256, 128, 128), (256, 256, 64), (128, 128, 64), (64, 64, 64), (9, 64, 64)]
    tot = {d: np.zeros(8) for d in designs}
    bands = [64, 128, 256]
    btot = {b: np.zeros(8) for b in bands}
    grps = [16, 64]
    b2tot = {g: np.zeros(8) for g in grps}
    frames = 0
    t0 = time.time()
    for task, names in sorted(data._videos_by_task.items()):
        vc = torch.tensor(data.corpus._indices_by_task[task])
        with torch.no_grad():
            tab = m.factor_tables(vc, torch.device('cpu'))
        for name in names[:per_task]:
            x = data._videos[(task, name)]['features'].double()
            elp = (tab['cst'] + x @ tab['w'] - 0.5 * (x * x) @ tab['inv_var'].unsqueeze(1)).numpy()
            for d in designs:
                tot[d] += probe(elp, tab['trans'].numpy(), tab['init'].numpy(), tab['len'].numpy(), *d)
            for bw in bands:
                btot[bw] += band_probe(elp, tab['trans'].numpy(), tab['init'].numpy(), tab['len'].numpy(), bw)
            for gp in grps:
                b2tot[gp] += band_probe2(elp, tab['trans'].numpy(), tab['init'].numpy(), tab['len'].numpy(), gp)
            frames += elp.shape[0]
            print('# %s %s T=%d C=%d  (%.0f s)' % (task, name, elp.shape[0], elp.shape[1], time.time() - t0), flush=True)
    print("workload %s seed 2 (CPU draw), %d videos per task, K = %d" % (wl, per_task, cfg['max_k']))
    print("%-28s %10s %10s %10s %12s %12s" % ("long range from / target block / length block", "blocks", "survive",
                                                "static-bnd", "long cells", "whole lattice"))
    for d in designs:
        o = tot[d]
        print("kl=%-4d tb=%-4d kb=%-4d          %10d %9.1f%% %9.1f%% %11.1f%% %11.1f%%" % (
            d[0], d[1], d[2], o[0], 100 * o[1] / o[0], 100 * o[5] / o[0], 100 * o[3] / o[2],
            100 * (o[4] - o[2] + o[3]) / o[4]))
    print()
    print("banded push (band 0 always evaluated; bands m >= 1 skipped per group of 64 sources by the bound test)")
    print("%-12s %12s %10s %12s %14s" % ("band width", "band-groups", "evaluated", "of which m=0", "lattice cells"))
    for bw in bands:
        o = btot[bw]
        print("bw=%-9d %12d %9.1f%% %11.1f%% %13.1f%%" % (bw, o[0], 100 * o[1] / o[0], 100 * o[2] / o[0], 100 * o[4] / o[3]))
    print()
    print("banded push as built: band 0 = lengths 9..127 (always), bands 1..8 = 16+112m..127+112m, sources delayed by 112m")
    print("%-12s %14s %10s %18s %14s" % ("test group", "band-groups>=1", "evaluated", "activations/kframe", "lattice cells"))
    for gp in grps:
        o = b2tot[gp]
        print("grp=%-8d %14d %9.2f%% %18.2f %13.1f%%" % (gp, o[0], 100 * o[1] / o[0], 1e3 * o[2] / frames, 100 * o[4] / o[3]))


if __name__ == '__main__':
    main()
