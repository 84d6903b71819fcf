"""Golden vectors for the evaluation counters, made by IMPORTING the reference's ``Accuracy`` in the build container.

    python tests/golden/make_golden_eval.py        ->  tests/golden/eval_vectors.json   (committed)

Each case holds the inputs (per-video multi-label ground truth, per-video predictions, background ids, the task's
label set, assignment mode) and the dict the reference returns from ``Accuracy.stat()`` after the call sequence of
``Datasplit.accuracy_corpus`` (src/data/corpus.py:486-565): ``mof(...)``, ``mof_classes()``, ``iou_classes()``,
``levenshtein()``, ``single_step_recall()``.

Not the reference's own arithmetic:
* ``editdistance`` is absent from this image; the stand-in below is the textbook Levenshtein distance over label
  sequences (what ``editdistance.eval`` computes).
* ``single_step_recall`` draws with an unseeded ``np.random.choice`` (accuracy.py:449): its two numerators are stored
  under ``random_keys`` and compared only where every draw gives the same answer.
* ``pred_label in [[...], []]`` (accuracy.py:524, 555) compares a numpy integer with an EMPTY list whenever a
  corpus-wide background id has no cluster in this task -- always, with more than one task.  The numpy the reference
  was written for evaluates that to False (with a DeprecationWarning); numpy 2.2 raises.  ``_NoCluster`` below is put
  in as the default of ``Accuracy._gt2cluster`` so the reference runs here with its original meaning (no match).
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference/src')


def _lev(a, b):
    a, b = list(a), list(b)
    row = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        new = [i]
        for j, y in enumerate(b, 1):
            new.append(min(row[j] + 1, new[j - 1] + 1, row[j - 1] + (x != y)))
        row = new
    return row[-1]


ed = types.ModuleType('editdistance')
ed.eval = _lev
sys.modules['editdistance'] = ed

import logging  # noqa: E402
from evaluation.accuracy import Accuracy  # noqa: E402

logging.getLogger('basic').setLevel(logging.ERROR)


class _NoCluster:
    """An empty cluster list that compares unequal to everything (what ``bool(np.int64(x) == [])`` used to give)."""
    __array_ufunc__ = None

    def __len__(self):
        return 0

    def __iter__(self):
        return iter(())

    def __getitem__(self, i):                  # (corpus.py:553 reads val[0] inside try / except IndexError)
        raise IndexError(i)

    def __eq__(self, other):
        return False

    __hash__ = None


class Corpus:
    def __init__(self, background, n):
        self._background_indices = list(background)
        self.index2label = {i: 'c%d' % i for i in range(n)}


def spans(rng, t, labels, mean_len):
    out = []
    while len(out) < t:
        out += [int(rng.choice(labels))] * int(max(1, rng.poisson(mean_len)))
    return out[:t]


def noisy(rng, seq, labels, flip, shift):
    seq = list(np.roll(seq, int(rng.integers(-shift, shift + 1))))
    i = 0
    while i < len(seq):
        ln = int(max(1, rng.poisson(6)))
        if rng.random() < flip:
            seq[i:i + ln] = [int(rng.choice(labels))] * len(seq[i:i + ln])
        i += ln
    return [int(x) for x in seq]


def make_cases():
    rng = np.random.default_rng(11)
    cases = {}
    # 1. supervised, identity assignment: task classes 10..16, background = even ids
    labels = list(range(10, 17))
    bkg = [10, 12, 14, 16, 30, 32]                          # corpus-wide list: other tasks' ids appear too
    gt = [spans(rng, t, labels, 9) for t in (120, 77, 150)]
    cases['identity'] = dict(gt=[[[x] for x in v] for v in gt], pred=[noisy(rng, v, labels, 0.3, 4) for v in gt],
                             background=bkg, possible=labels, optimal=False)
    # 2. multi-label ground truth (second label only where the first is a step; never background)
    gtm = []
    for v in gt:
        vm = []
        for x in v:
            if x % 2 == 1 and rng.random() < 0.15:
                other = int(rng.choice([y for y in (11, 13, 15) if y != x]))
                vm.append([x, other])
            else:
                vm.append([x])
        gtm.append(vm)
    cases['multi_label'] = dict(gt=gtm, pred=cases['identity']['pred'], background=bkg, possible=labels, optimal=False)
    # 3. predictions use a subset of the classes; one class of the task never occurs in the ground truth
    sub = [10, 11, 12, 13, 14]
    gt3 = [spans(rng, t, sub, 12) for t in (90, 64)]
    cases['pred_subset'] = dict(gt=[[[x] for x in v] for v in gt3],
                                pred=[[x if x in (10, 11, 12) else 10 for x in noisy(rng, v, sub, 0.2, 3)] for v in gt3],
                                background=bkg, possible=labels, optimal=False)
    # 4. perfect predictions (every random draw of single_step_recall agrees)
    cases['perfect'] = dict(gt=[[[x] for x in v] for v in gt], pred=[list(v) for v in gt], background=bkg,
                            possible=labels, optimal=False)
    # 5. Hungarian assignment: the predictor's state ids are a permutation of the task's ids
    perm = dict(zip(labels, [13, 10, 16, 11, 15, 12, 14]))
    cases['hungarian'] = dict(gt=[[[x] for x in v] for v in gt],
                              pred=[[perm[x] for x in p] for p in cases['identity']['pred']],
                              background=bkg, possible=labels, optimal=True)
    # 6. Hungarian with fewer predicted labels than ground-truth labels (the voting table is padded)
    few = {10: 10, 11: 11, 12: 10, 13: 13, 14: 10, 15: 11, 16: 13}
    cases['hungarian_padded'] = dict(gt=[[[x] for x in v] for v in gt],
                                     pred=[[few[x] for x in p] for p in cases['identity']['pred']],
                                     background=bkg, possible=labels, optimal=True)
    # 7. one short video, a single segment each side
    cases['single_segment'] = dict(gt=[[[11]] * 7], pred=[[11] * 7], background=bkg, possible=labels, optimal=False)
    # 8. no background at all in the ground truth or predictions
    steps = [21, 23, 25]
    gt8 = [spans(rng, t, steps, 8) for t in (60, 45)]
    cases['no_background'] = dict(gt=[[[x] for x in v] for v in gt8], pred=[noisy(rng, v, steps, 0.4, 5) for v in gt8],
                                  background=[20, 22, 24, 26], possible=list(range(20, 27)), optimal=False)
    return cases


RANDOM_KEYS = ('single_step_recall', 'step_recall_non_bg')


def run(case):
    np.random.seed(0)
    acc = Accuracy(verbose=False, corpus=Corpus(case['background'], 64))
    from collections import defaultdict
    acc._gt2cluster = defaultdict(_NoCluster)
    for g, p in zip(case['gt'], case['pred']):
        acc.add_gt_labels([list(x) for x in g])
        acc.add_predicted_labels(list(p))
    frames = acc.mof(case['optimal'], possible_gt_labels=case['possible'])
    mof_val = acc.mof_val()
    acc.mof_classes()
    acc.iou_classes()
    acc.levenshtein()
    acc.single_step_recall()
    stat = {k: [float(v[0]), float(v[1])] for k, v in acc.stat().items()}
    gt2cluster = {int(k): [int(x) for x in v] for k, v in acc._gt2cluster.items() if len(v)}
    classes_mof = {int(k): [float(v[0]), float(v[1])] for k, v in acc._classes_MoF.items()}
    classes_iou = {int(k): [float(v[0]), float(v[1])] for k, v in acc._classes_IoU.items()}
    return dict(stat=stat, frames=int(frames), mof_val=float(mof_val), gt2cluster=gt2cluster,
                classes_mof=classes_mof, classes_iou=classes_iou)


# ------------------------------------------------------------------------------------------------ Datasplit level
# ``Datasplit.accuracy_corpus`` itself (src/data/corpus.py:405-604) on a stand-in datasplit: what it does ON TOP of
# ``Accuracy`` is (a) multi-label ground truth from ``video.gt()``, (b) the re-expansion of predictions made on every
# ``subsample``-th frame (:466-472), (c) the canonicalisation of background labels under
# ``--annotate_background_with_previous`` (:474-480: EVERY background id of the corpus becomes the corpus' first one).
def make_datasplit_cases():
    rng = np.random.default_rng(23)
    tasks = {'t0': list(range(0, 7)), 't1': list(range(7, 16))}           # chains BKG, step, BKG, ...: even positions = background
    bkg = [ids[i] for ids in tasks.values() for i in range(0, len(ids), 2)]
    lengths = {'t0': (95, 61, 130), 't1': (88, 143)}
    base = {}
    for task, ids in tasks.items():
        vids = {}
        for vi, t in enumerate(lengths[task]):
            gt = []
            cur = 0
            while len(gt) < t:
                gt += [ids[cur % len(ids)]] * int(max(1, rng.poisson(7)))
                cur += 1
            gt = gt[:t]
            steps = [i for i in ids if i not in bkg]
            gtm = [[x, int(rng.choice([y for y in steps if y != x]))] if (x not in bkg and rng.random() < 0.12) else [x]
                   for x in gt]
            vids['%s_v%d' % (task, vi)] = dict(gt=gtm, pred_full=noisy(rng, gt, ids, 0.3, 4))
        base[task] = vids
    cases = {}
    for name, sub, canon, optimal in (('ds_multi_label', 1, False, False), ('ds_subsample3', 3, False, False),
                                      ('ds_canonical_background', 1, True, False), ('ds_subsample2_canonical_optimal', 2, True, True)):
        videos = {}
        for task, vids in base.items():
            videos[task] = {n: dict(gt=v['gt'], pred=v['pred_full'][:(len(v['pred_full']) // sub) * sub:sub] if sub > 1 else v['pred_full'])
                            for n, v in vids.items()}
        cases[name] = dict(tasks=tasks, background=bkg, subsample=sub, annotate_background_with_previous=canon,
                           optimal=optimal, videos=videos)
    return cases


def run_datasplit(case):
    import data.corpus as DC

    class _F1:                                # (F1Score samples with np.random.random_integers, gone from numpy 2; its keys
        def __init__(self, **kw): pass        #  are not among the statistics this build reproduces)
        def set_gt(self, *a): pass
        def set_pr(self, *a): pass
        def set_gt2pr(self, *a): pass
        def f1(self): pass
        def stat(self): return {}
    DC.F1Score = _F1

    class _Video:
        def __init__(self, gt, pred):
            self._gt, self._pred, self.segmentation, self.iter = gt, pred, {}, 0

        def gt(self):
            return [list(x) for x in self._gt]

    class _Corpus(Corpus):
        def __init__(self, background, tasks, canon):
            Corpus.__init__(self, background, 64)
            self._tasks, self.annotate_background_with_previous = tasks, canon

        def indices_by_task(self, task):
            return list(self._tasks[task])

    ds = object.__new__(DC.Datasplit)
    ds._corpus = _Corpus(case['background'], case['tasks'], case['annotate_background_with_previous'])
    ds._videos_by_task = {t: {n: _Video(v['gt'], v['pred']) for n, v in vids.items()} for t, vids in case['videos'].items()}
    ds._K_by_task = {t: len(ids) for t, ids in case['tasks'].items()}
    ds.subsample = case['subsample']
    ds._gt2label, ds._label2gt = None, {}
    # (same stand-in for the empty cluster lists as in run(): Accuracy is constructed inside accuracy_corpus)
    from collections import defaultdict
    orig_init = Accuracy.__init__

    def patched(self, *a, **kw):
        orig_init(self, *a, **kw)
        self._gt2cluster = defaultdict(_NoCluster)
    DC.Accuracy.__init__ = patched
    try:
        np.random.seed(0)
        stats = ds.accuracy_corpus(case['optimal'], lambda video: video._pred, verbose=False)
    finally:
        DC.Accuracy.__init__ = orig_init
    return {t: {k: [float(v[0]), float(v[1])] for k, v in st.items()} for t, st in stats.items()}


def main():
    out = {'random_keys': list(RANDOM_KEYS), 'cases': {}}
    for name, case in make_cases().items():
        out['cases'][name] = dict(inputs=case, expected=run(case))
        print(name, {k: v for k, v in out['cases'][name]['expected']['stat'].items() if k in ('mof', 'f1', 'mean_normed_levenshtein')})
    out['datasplit_cases'] = {}
    for name, case in make_datasplit_cases().items():
        out['datasplit_cases'][name] = dict(inputs=case, expected=run_datasplit(case))
        print(name, {t: st['mof'] for t, st in out['datasplit_cases'][name]['expected'].items()})
    with open(os.path.join(HERE, 'eval_vectors.json'), 'w') as f:
        json.dump(out, f)


if __name__ == '__main__':
    main()
