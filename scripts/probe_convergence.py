"""CPU experiment (round 5): how quickly does the semi-Markov forward recursion forget its start?
oracle/prune_probe.c (smm_conv_probe) restarts the (max,+) forward pass at position a from a FLAT ring (beta = 0 at every
ring position and state) and reports the first n0 from which every later value differs from the true one by ONE constant.
cfg1's video and the longest videos of the cfg3 seed-2 corpus (CPU draw), closed-form-fitted parameters like bench.py;
a in {T/8, ..., 7T/8}.

    python scripts/probe_convergence.py [n_longest] > profiles/round5_rank_convergence.txt
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
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-ffp-contract=off', '-fopenmp', '-shared', '-o', so, src, '-lm'])
lib = ctypes.CDLL(so)
P = ctypes.POINTER(ctypes.c_double)
PI = ctypes.POINTER(ctypes.c_int32)


def probe(elp, trans, init, len_scores, starts, tol=1e-5, mode=0):
    t, c = elp.shape
    kp = min(len_scores.shape[0], t)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (elp, trans, init, len_scores[:kp])]
    st = np.ascontiguousarray(starts, dtype=np.int32)
    out = np.zeros(len(st), dtype=np.int32)
    rc = lib.smm_conv_probe_mode(arrs[0].ctypes.data_as(P), t, c, arrs[1].ctypes.data_as(P), arrs[2].ctypes.data_as(P),
                                 arrs[3].ctypes.data_as(P), kp, st.ctypes.data_as(PI), len(st), ctypes.c_double(tol), mode,
                                 out.ctypes.data_as(PI))
    assert rc == 0
    return out


def corpus(wl, seed, longest):
    cfg = synth.CONFIGS[wl]
    dry = synth.SynthDatasplit(wl, seed=seed, keep=set())
    lens = sorted(((int(dry[(t, n)]['features'].shape[0]), t, n) for t, ns in dry._videos_by_task.items() for n in ns), reverse=True)
    want = {n for _, _, n in lens[:longest]}
    keep = want | {n for names in dry._videos_by_task.values() for n in names[:6]}
    data = synth.SynthDatasplit(wl, seed=seed, keep=keep)
    args = synth.make_args(cfg['max_k'], cuda=False, batch_size=cfg['batch_size'])
    model = SemiMarkovModel.from_args(args, data)
    model.fit(data.subset(6), use_labels=True)
    return data, model.model, [(t, n) for _, t, n in lens[:longest]]


def main():
    longest = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for mode, what in ((0, 'a flat ring (beta = 0 at every ring position)'), (1, 'a boundary forced at a, uniform start (h[a][c] = 0, nothing older): what a chunk of a time-split decode would start from')):
        print('==== restart from ' + what)
        run(longest, mode)


def run(longest, mode):
    rows = []
    t0 = time.time()
    for wl, seed, nl in (('cfg1', 1, 1), ('cfg3', 2, longest)):
        data, m, vids = corpus(wl, seed, nl)
        for task, name in vids:
            vc = torch.tensor(data.corpus._indices_by_task[task])
            with torch.no_grad():
                tab = m.factor_tables(vc, torch.device('cpu'))
            smp = data._videos[(task, name)]
            x = smp['features'].double()
            elp = (tab['cst'] + x @ tab['w'] - 0.5 * (x * x) @ tab['inv_var'].unsqueeze(1)).numpy()
            t = elp.shape[0]
            starts = [t * i // 8 for i in range(1, 8)] if mode == 0 else [t * i // 29 for i in range(1, 29)]
            o = probe(elp, tab['trans'].numpy(), tab['init'].numpy(), tab['len'].numpy(), starts, mode=mode)
            # the true segmentation around each start, for reading the numbers: frames from a to the next two boundaries
            gt = smp['gt_single'].numpy()
            cuts = np.flatnonzero(np.diff(gt)) + 1
            nxt = [[int(cc - a) for cc in cuts[cuts > a][:2]] for a in starts]
            rows += [(wl, name, t, elp.shape[1], a, int(d), nb) for a, d, nb in zip(starts, o, nxt)]
            print('# %s %s T=%d C=%d: n0 - a = %s   (frames to the next two true boundaries: %s)  [%.0f s]' % (
                wl, name, t, elp.shape[1], ' '.join(str(int(v)) for v in o), nxt, time.time() - t0), flush=True)
    d = np.array([r[5] for r in rows])
    print("rank convergence of the (max,+) forward recursion, tolerance 1e-5 on values of ~1e6:")
    print("%d restarts (%d videos x %d positions): n0 - a  min %d  median %d  mean %.0f  90 %% %d  99 %% %d  max %d" % (
        len(d), len(rows) // len(starts), len(starts), d.min(), np.median(d), d.mean(), np.percentile(d, 90), np.percentile(d, 99), d.max()))
    for lim in (256, 512, 1024, 1500, 2048, 3000):
        print("   converged within %4d positions: %5.1f %%" % (lim, 100.0 * (d <= lim).mean()))


if __name__ == '__main__':
    main()
