"""CPU experiment (round 4): how many of the 8 sources of a hand-over block do the band-0 pushers of the Viterbi BAND
kernel have to push when a source that its successor dominates at every target is left out?  oracle/prune_probe.c
(smm_dom_probe) counts; videos of the cfg3 seed-2 corpus (CPU draw), closed-form-fitted parameters, like bench.py.

    python scripts/probe_dominance.py [videos_per_task] [workload]      -> table on stdout (copy under profiles/)
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


def probe(elp, trans, init, len_scores):
    t, c = elp.shape
    kp = min(len_scores.shape[0], t)
    out = np.zeros(8)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (elp, trans, init, len_scores[:kp])]
    rc = lib.smm_dom_probe(arrs[0].ctypes.data_as(P), t, c, arrs[1].ctypes.data_as(P), arrs[2].ctypes.data_as(P),
                           arrs[3].ctypes.data_as(P), kp, out.ctypes.data_as(P))
    assert rc == 0
    return out


def probe3(elp, trans, init, len_scores):
    t, c = elp.shape
    kp = min(len_scores.shape[0], t)
    out = np.zeros(8)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (elp, trans, init, len_scores[:kp])]
    rc = lib.smm_band_probe3(arrs[0].ctypes.data_as(P), t, c, arrs[1].ctypes.data_as(P), arrs[2].ctypes.data_as(P),
                             arrs[3].ctypes.data_as(P), kp, out.ctypes.data_as(P))
    assert rc == 0
    return out


def main():
    per_task = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    wl = sys.argv[2] if len(sys.argv) > 2 else 'cfg3'
    cfg = synth.CONFIGS[wl]
    dry = synth.SynthDatasplit(wl, seed=2, keep=set())
    keep = {n for names in dry._videos_by_task.values() for n in names[:6]}
    data = synth.SynthDatasplit(wl, seed=2, keep=keep)
    args = synth.make_args(cfg['max_k'], cuda=False, batch_size=cfg['batch_size'])
    model = SemiMarkovModel.from_args(args, data)
    model.fit(data.subset(6), use_labels=True)
    m = model.model
    tot = np.zeros(8)
    tot3 = np.zeros(8)
    t0 = time.time()
    for task, names in sorted(data._videos_by_task.items()):
        vc = torch.tensor(data.corpus._indices_by_task[task])
        with torch.no_grad():
            tab = m.factor_tables(vc, torch.device('cpu'))
        for name in names[:per_task]:
            x = data._videos[(task, name)]['features'].double()
            elp = (tab['cst'] + x @ tab['w'] - 0.5 * (x * x) @ tab['inv_var'].unsqueeze(1)).numpy()
            o = probe(elp, tab['trans'].numpy(), tab['init'].numpy(), tab['len'].numpy())
            tot += o
            o3 = probe3(elp, tab['trans'].numpy(), tab['init'].numpy(), tab['len'].numpy())
            tot3 += o3
            print('#    delayed band-groups: %d with sources, %d evaluated with the one witness of round 3 (%.2f %%), %d with every complete group as a witness (%.2f %%)' % (
                o3[0], o3[1], 100 * o3[1] / o3[0], o3[2], 100 * o3[2] / o3[0]), flush=True)
            print('# %s %s T=%d C=%d: %.2f pushes per (state, block), leader %.2f  (%.0f s)' % (
                task, name, elp.shape[0], elp.shape[1], o[1] / o[0], o[5] / o[4], time.time() - t0), flush=True)
    print("workload %s seed 2 (CPU draw), %d videos per task, K = %d" % (wl, per_task, cfg['max_k']))
    print("sources per (state, block of 8): %.3f of %.3f pushed with the successor test (%.1f %%); every-later-source test: %.3f" % (
        tot[1] / tot[0], tot[2] / tot[0], 100 * tot[1] / tot[2], tot[6] / tot[0]))
    print("delayed band-groups (16 sources x one band of one state): %d; evaluated with round 3's witness (group g - 2): %.3f %%; with every complete group g - 2 .. g - 55 as a witness, strict: %.3f %% (%.2f x fewer)" % (
        tot3[0], 100 * tot3[1] / tot3[0], 100 * tot3[2] / tot3[0], tot3[1] / max(1.0, tot3[2])))
    print("(state, block) pairs that push the last source only: %.1f %%; largest push count of a block's states, mean: %.2f" % (
        100 * tot[3] / tot[0], tot[5] / tot[4]))


if __name__ == '__main__':
    main()
