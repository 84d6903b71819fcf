"""Randomised soak of the two device-side passes either side of the decode (round 5): the closed-form fit's sufficient statistics
(csrc/smm_fit.hip) against oracle/dense_ref.py, and the evaluation counters (csrc/smm_eval.hip) against oracle/eval_ref.py -- the
unit tests' bodies (tests/test_gpu_fit.py, tests/test_gpu_eval.py) on random shapes: D 1..300, 2..40 classes (one never occurring),
span limits 2..64 or none, videos of 1..2500 frames; 1..5 tasks of 2..25 labels, multi-label ground truth, optimal assignment on
and off.  Integer statistics exact, sums rel 1e-12.    usage: soak_fit_eval.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import dense_ref as O, eval_ref
from test_gpu_fit import make_videos
from test_gpu_eval import random_task, assert_stats
from action_segmentation_amd.semimarkov_utils import semimarkov_sufficient_stats_device
from action_segmentation_amd.evaluation import LabelSpace, evaluate_labels

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, nf, ne, last = time.time(), 0, 0, time.time()
while time.time() - t0 < budget:
    # ---- fit statistics
    d = int(rng.choice([1, 3, 7, 16, 64, 200, 257, 300])); nc = int(rng.integers(2, 41))
    max_k = None if rng.random() < 0.2 else int(rng.integers(2, 65))
    t_hi = int(rng.choice([20, 90, 400, 2500]))
    absent = int(rng.integers(0, nc))
    classes = [c for c in range(nc) if c != absent] or [0]
    feats, labels = make_videos(rng, int(rng.integers(1, 9)), 1, t_hi, nc, d, mean_len=float(rng.choice([2, 11, 40])), classes=classes)
    em, st = semimarkov_sufficient_stats_device([torch.from_numpy(f) for f in feats], [torch.from_numpy(l) for l in labels],
                                                'tied_diag', nc, max_k)
    want = O.sufficient_stats(feats, labels, nc, max_k)
    for key in ('span_counts', 'span_lengths', 'span_start_counts', 'span_transition_counts'):
        np.testing.assert_array_equal(st[key], want[key], err_msg=str((key, d, nc, max_k, t_hi)))
    assert st['instance_count'] == want['instance_count']
    np.testing.assert_allclose(em.means_, want['means'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(em.covariances_[0], want['var'], rtol=1e-11, atol=0)
    nf += 1
    # ---- evaluation counters
    optimal = bool(rng.random() < 0.4); width = int(rng.integers(1, 3))
    by_task, bkg, data, nxt = {}, [], {}, 0
    for ti in range(int(rng.integers(1, 6))):
        c = int(rng.integers(2, 26))
        ids = list(range(nxt, nxt + c)); nxt += c
        by_task['t%d' % ti] = ids
        bkg += ids[0::2]
        data['t%d' % ti] = random_task(rng, ids, n_videos=int(rng.integers(1, 6)), t_lo=int(rng.choice([1, 40])), t_hi=int(rng.choice([60, 400, 3000])),
                                       width=width, mean_len=float(rng.choice([1.5, 3, 25])), flip=float(rng.choice([0.0, 0.3, 0.9])))
    if optimal:
        for t, (gt, pred) in data.items():
            perm = dict(zip(by_task[t], rng.permutation(by_task[t]).tolist()))
            data[t] = (gt, [[perm[x] for x in v] for v in pred])
    space = LabelSpace(by_task, bkg)
    order = [(t, i) for t in by_task for i in range(len(data[t][0]))]
    rng.shuffle(order)
    lengths, offsets, tasks, keys, pr, g, off = [], [], [], [], [], [], 0
    for t, i in order:
        frames, pred = data[t][0][i], data[t][1][i]
        lengths.append(len(pred)); offsets.append(off); off += len(pred); tasks.append(t); keys.append(i)
        pr.append(np.asarray(pred, dtype=np.int64))
        gm = np.full((len(pred), width), -1, dtype=np.int64)
        for r, f in enumerate(frames):
            gm[r, :len(f)] = f
        g.append(gm)
    seed = int(rng.integers(0, 1000))
    got = evaluate_labels(torch.from_numpy(np.concatenate(pr)).cuda(), torch.from_numpy(np.concatenate(g)).cuda(), lengths, offsets, tasks,
                          space, optimal, seed=seed, video_key=keys)
    for t in by_task:
        want, _ = eval_ref.task_counters(data[t][0], data[t][1], bkg, by_task[t], optimal, seed=seed)
        assert_stats(got[t], want)
    ne += 1
    if time.time() - last > 20:
        last = time.time()
        print('  ... %d fits, %d evaluations' % (nf, ne), flush=True)
print('soak ok: %d fit-statistics passes and %d evaluation passes on random shapes equal to the oracle (integers exact, sums rel 1e-12), %.0f s' % (nf, ne, time.time() - t0))
