"""Device evaluation counters (csrc/smm_eval.hip through evaluation.py) against the reference's Accuracy outputs
(tests/golden/eval_vectors.json) and against the CPU restatement (oracle/eval_ref.py).  Integer work: exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import eval_ref

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'eval_vectors.json')) as f:
    GOLD = json.load(f)
CASES = sorted(GOLD['cases'])


class Corpus:
    def __init__(self, background):
        self._background_indices = list(background)


def assert_stats(got, want, skip=()):
    assert set(got) == set(want), set(got) ^ set(want)
    for key, pair in want.items():
        if key in skip:
            assert float(got[key][1]) == float(pair[1]), key
            continue
        np.testing.assert_allclose(np.asarray(got[key], dtype=np.float64), np.asarray(pair, dtype=np.float64),
                                   rtol=1e-12, atol=0, err_msg=key)


@pytest.mark.parametrize('name', CASES)
def test_accuracy_class_matches_reference_and_oracle(name):
    from action_segmentation_amd.evaluation import Accuracy
    inp, exp = GOLD['cases'][name]['inputs'], GOLD['cases'][name]['expected']
    acc = Accuracy(verbose=False, corpus=Corpus(inp['background']), seed=5)
    for g, p in zip(inp['gt'], inp['pred']):
        acc.add_gt_labels([list(x) for x in g])
        acc.add_predicted_labels(list(p))
    frames = acc.mof(inp['optimal'], possible_gt_labels=inp['possible'])
    acc.mof_classes()
    acc.iou_classes()
    acc.levenshtein()
    acc.single_step_recall()
    assert frames == exp['frames']
    assert acc.mof_val() == pytest.approx(exp['mof_val'], rel=1e-15)
    deterministic = name in ('perfect', 'single_segment')
    assert_stats(acc.stat(), exp['stat'], skip=() if deterministic else GOLD['random_keys'])
    assert {int(k): v for k, v in exp['gt2cluster'].items()} == {k: v for k, v in acc._gt2cluster.items() if v}
    assert {int(k): v for k, v in exp['classes_mof'].items()} == acc._classes_MoF
    assert {int(k): v for k, v in exp['classes_iou'].items()} == acc._classes_IoU
    want, _ = eval_ref.task_counters(inp['gt'], inp['pred'], inp['background'], inp['possible'], inp['optimal'], seed=5)
    assert_stats(acc.stat(), want)                      # same hash draw on both sides: every key, exactly


def random_task(rng, ids, n_videos, t_lo, t_hi, width, mean_len, flip):
    gt, pred = [], []
    steps = [i for i in ids if i % 2 == 1] or ids
    for _ in range(n_videos):
        t = int(rng.integers(t_lo, t_hi + 1))
        seq = []
        while len(seq) < t:
            seq += [int(rng.choice(ids))] * int(max(1, rng.poisson(mean_len)))
        seq = seq[:t]
        frames = []
        for x in seq:
            if width > 1 and x in steps and len(steps) > 1 and rng.random() < 0.1:
                frames.append([x, int(rng.choice([s for s in steps if s != x]))])
            else:
                frames.append([x])
        p = list(np.roll(seq, int(rng.integers(-5, 6))))
        i = 0
        while i < t:
            ln = int(max(1, rng.poisson(mean_len)))
            if rng.random() < flip:
                p[i:i + ln] = [int(rng.choice(ids))] * len(p[i:i + ln])
            i += ln
        gt.append(frames)
        pred.append([int(x) for x in p])
    return gt, pred


@pytest.mark.parametrize('optimal,width', [(False, 1), (False, 2), (True, 1)])
def test_multi_task_packed_matches_oracle(optimal, width):
    from action_segmentation_amd.evaluation import LabelSpace, evaluate_labels
    rng = np.random.default_rng(3 + width + 10 * optimal)
    by_task, bkg, data = {}, [], {}
    nxt = 0
    for ti, c in enumerate((5, 11, 23, 7)):
        ids = list(range(nxt, nxt + c))
        nxt += c
        by_task['t%d' % ti] = ids
        bkg += ids[0::2]
        data['t%d' % ti] = random_task(rng, ids, n_videos=3 + ti, t_lo=40, t_hi=3000 if ti == 2 else 400, width=width,
                                       mean_len=3 if ti == 1 else 25, flip=0.3)
    if optimal:      # predictions in a permuted id space of the same task
        for t, (gt, pred) in data.items():
            perm = dict(zip(by_task[t], rng.permutation(by_task[t]).tolist()))
            data[t] = (gt, [[perm[x] for x in v] for v in pred])
    space = LabelSpace(by_task, bkg)
    # interleave the tasks' videos on the packed axis
    order = [(t, i) for t in by_task for i in range(len(data[t][0]))]
    rng.shuffle(order)
    lengths, offsets, tasks, keys, pr, g = [], [], [], [], [], []
    off = 0
    for t, i in order:
        frames, pred = data[t][0][i], data[t][1][i]
        lengths.append(len(pred))
        offsets.append(off)
        off += len(pred)
        tasks.append(t)
        keys.append(i)
        pr.append(np.asarray(pred, dtype=np.int64))
        gm = np.full((len(pred), width), -1, dtype=np.int64)
        for r, f in enumerate(frames):
            gm[r, :len(f)] = f
        g.append(gm)
    pred_d = torch.from_numpy(np.concatenate(pr)).cuda()
    gt_d = torch.from_numpy(np.concatenate(g)).cuda()
    got = evaluate_labels(pred_d, gt_d, lengths, offsets, tasks, space, optimal, seed=9, video_key=keys)
    for t in by_task:
        want, _ = eval_ref.task_counters(data[t][0], data[t][1], bkg, by_task[t], optimal, seed=9)
        assert_stats(got[t], want)


def test_edit_distance_long_sequences():
    """Every frame its own segment: sequences longer than one wave, both orders of (n, m)."""
    from action_segmentation_amd.evaluation import LabelSpace, evaluate_labels
    rng = np.random.default_rng(0)
    ids = list(range(6))
    space = LabelSpace({'a': ids}, [0])
    for tg, tp in ((333, 333), (200, 200), (65, 65), (64, 64), (1, 1)):
        gt = rng.integers(0, 6, size=tg)
        pred = np.repeat(rng.integers(0, 6, size=(tp + 2) // 3), 3)[:tp] if tp > 3 else rng.integers(0, 6, size=tp)
        got, ex = evaluate_labels(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), [tg], [0], ['a'], space,
                                  False, return_extras=True)
        want, wex = eval_ref.task_counters([[[int(x)] for x in gt]], [pred.tolist()], [0], ids, False)
        assert_stats(got['a'], want)
        assert got['a']['total_levenshtein'][0] == wex['levenshtein'][0]


def test_labels_outside_the_task_are_reported():
    from action_segmentation_amd.evaluation import LabelSpace, evaluate_labels
    space = LabelSpace({'a': [0, 1, 2], 'b': [3, 4]}, [0, 3])
    pred = torch.tensor([0, 1, 4, 2], dtype=torch.int64).cuda()          # 4 belongs to task b
    gt = torch.tensor([0, 1, 1, 2], dtype=torch.int64).cuda()
    with pytest.raises(ValueError, match="outside"):
        evaluate_labels(pred, gt, [4], [0], ['a'], space, False)


def test_accuracy_corpus_on_decoded_synthetic_split():
    """End to end: fit, decode, evaluate on the device; MoF equals the plain numpy count."""
    from action_segmentation_amd import synth
    from action_segmentation_amd.evaluation import accuracy_corpus, summarise, STAT_KEYS
    from action_segmentation_amd.semimarkov import SemiMarkovModel
    data = synth.SynthDatasplit('tiny', seed=0)
    args = synth.make_args(data.max_k)
    model = SemiMarkovModel.from_args(args, data)
    model.fit(data, use_labels=True)
    preds = model.predict(data)
    stats = accuracy_corpus(data, preds, optimal_assignment=False)
    flat = summarise(stats, STAT_KEYS, prefix='train_')
    hit = sum(int((np.asarray(preds[n]) == smp['gt_single'].numpy()).sum()) for (_, n), smp in data._videos.items())
    tot = sum(int(smp['gt_single'].numel()) for smp in data._videos.values())
    assert flat['train_mof'] == pytest.approx(hit / tot, rel=1e-15)
    gts = {t: [[[int(x)] for x in data._videos[(t, n)]['gt_single'].tolist()] for n in data._videos_by_task[t]]
           for t in data._videos_by_task}
    for t in data._videos_by_task:
        want, _ = eval_ref.task_counters(gts[t], [preds[n].tolist() for n in data._videos_by_task[t]],
                                         data.corpus._background_indices, data.corpus._indices_by_task[t], False)
        assert_stats(stats[t], want)


def test_sharded_evaluation_reduces_to_the_single_process_answer():
    """Two 'ranks' (threads sharing one GPU) each hold half of every task's videos; ``reduce`` sums their tensors the
    way an all-reduce would.  Hungarian assignment, so the second phase depends on the reduced confusion table."""
    import threading
    from action_segmentation_amd.evaluation import LabelSpace, evaluate_labels
    rng = np.random.default_rng(21)
    by_task = {'a': list(range(0, 7)), 'b': list(range(7, 16))}
    bkg = [0, 2, 4, 6, 7, 9, 11, 13, 15]
    data = {t: random_task(rng, ids, n_videos=6, t_lo=50, t_hi=500, width=1, mean_len=12, flip=0.35)
            for t, ids in by_task.items()}
    for t, (gt, pred) in data.items():
        perm = dict(zip(by_task[t], rng.permutation(by_task[t]).tolist()))
        data[t] = (gt, [[perm[x] for x in v] for v in pred])
    space = LabelSpace(by_task, bkg)

    def shard(rank):
        lengths, offsets, tasks, keys, pr, g = [], [], [], [], [], []
        off = 0
        for t in by_task:
            for i in range(rank, 6, 2):
                pred = data[t][1][i]
                lengths.append(len(pred)); offsets.append(off); tasks.append(t); keys.append(i)
                off += len(pred)
                pr.append(np.asarray(pred, dtype=np.int64))
                g.append(np.asarray([f[0] for f in data[t][0][i]], dtype=np.int64))
        return (torch.from_numpy(np.concatenate(pr)).cuda(), torch.from_numpy(np.concatenate(g)).cuda(),
                lengths, offsets, tasks, keys)

    barrier = threading.Barrier(2)
    slots, results, errors = [None, None], [None, None], []

    def run(rank):
        try:
            torch.cuda.set_device(0)

            def reduce(tn):
                torch.cuda.synchronize()
                slots[rank] = tn.clone()
                barrier.wait()
                total = slots[0] + slots[1]
                barrier.wait()
                return total
            pred, gt, lengths, offsets, tasks, keys = shard(rank)
            with torch.cuda.stream(torch.cuda.Stream()):
                results[rank] = evaluate_labels(pred, gt, lengths, offsets, tasks, space, True, seed=4,
                                                video_key=keys, reduce=reduce)
        except Exception as e:           # surface in the main thread
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in by_task:
        want, _ = eval_ref.task_counters(data[t][0], data[t][1], bkg, by_task[t], True, seed=4)
        assert_stats(results[0][t], want)
        assert_stats(results[1][t], want)


class _DsCorpus:
    def __init__(self, inp):
        self._background_indices = list(inp['background'])
        self._tasks = {t: list(v) for t, v in inp['tasks'].items()}
        self.annotate_background_with_previous = inp['annotate_background_with_previous']
        self.n_classes = max(max(v) for v in self._tasks.values()) + 1

    def indices_by_task(self, task):
        return self._tasks[task]


class _Ds:
    def __init__(self, inp):
        self.corpus = _DsCorpus(inp)
        self.subsample = inp['subsample']
        self._videos_by_task = {t: list(v) for t, v in inp['videos'].items()}
        self._videos = {(t, n): dict(gt=v['gt']) for t, vids in inp['videos'].items() for n, v in vids.items()}


@pytest.mark.parametrize('name', sorted(GOLD['datasplit_cases']))
def test_accuracy_corpus_datasplit_semantics_match_the_reference(name):
    """``evaluation.accuracy_corpus`` against the reference's own ``Datasplit.accuracy_corpus`` (goldens made by running
    it on a stand-in datasplit): multi-label ground truth, --frame_subsample re-expansion, background canonicalisation
    under --annotate_background_with_previous, identity and Hungarian assignment -- and against the CPU restatement with
    the same seeded draw (every key, exactly)."""
    from action_segmentation_amd.evaluation import accuracy_corpus
    inp, exp = GOLD['datasplit_cases'][name]['inputs'], GOLD['datasplit_cases'][name]['expected']
    preds = {n: np.asarray(v['pred'], dtype=np.int64) for vids in inp['videos'].values() for n, v in vids.items()}
    got = accuracy_corpus(_Ds(inp), preds, inp['optimal'], seed=5)
    assert set(got) == set(exp)
    for task in exp:
        assert_stats(got[task], exp[task], skip=GOLD['random_keys'])
    want = eval_ref.datasplit_counters(inp['tasks'], inp['background'], inp['videos'], inp['subsample'],
                                       inp['annotate_background_with_previous'], inp['optimal'], seed=5)
    for task in want:
        assert_stats(got[task], want[task])
