"""Evaluation of decoded labels: the reference's ``Accuracy`` statistics from device-side counters.

Mirrors ``src/evaluation/accuracy.py`` (class ``Accuracy``) and the loop of ``Datasplit.accuracy_corpus``
(``src/data/corpus.py:405-600``).  Everything proportional to the number of frames runs in two HIP kernels
(``csrc/smm_eval.hip``: confusion table per task; per-video frame counters, run-length sequences, edit distance, step
recall); the host keeps what is O(#classes): the label assignment (identity, or Hungarian on the confusion table,
accuracy.py:232-318) and the final ``[numerator, denominator]`` pairs.

    stats_by_task = evaluate_labels(pred, gt, meta, label_space, optimal_assignment)     # tensors stay on the device
    acc = Accuracy(corpus=corpus); acc.add_gt_labels(..); acc.add_predicted_labels(..); acc.mof(False); ...; acc.stat()

Differences from the reference, all deliberate:
* ``single_step_recall`` draws its random frame with a seeded counter hash instead of the unseeded global numpy RNG
  (accuracy.py:449), so the statistic is reproducible and shardable;
* an empty cluster never matches a prediction (the reference relies on ``bool(np.int64(x) == [])`` being False, which
  numpy >= 2.2 turns into an exception);
* ``F1Score`` (src/evaluation/f1.py: 50 rounds of unseeded sampling through ``np.random.random_integers``, an API
  numpy 2 no longer has) is not reproduced; the ``precision`` / ``recall`` keys hold ``Accuracy``'s own counters.
Multi-process: pass ``reduce`` (a function summing an int64/fp64 tensor over ranks) and every statistic is finalised
from reduced sums, so a task's videos may live on different ranks.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import _lib, ops

CN = {name: i for i, name in enumerate(_lib.EVAL_COUNTER_NAMES)}


class LabelSpace:
    """Tasks of a corpus: ``task -> sorted global class ids``, the background ids and the size of the id space."""

    def __init__(self, indices_by_task, background, n_labels=None):
        self.tasks = list(indices_by_task)
        self.ids = {t: sorted(int(i) for i in indices_by_task[t]) for t in self.tasks}
        self.background = set(int(b) for b in background)
        top = max([max(v) for v in self.ids.values() if v] + [0])
        self.n_labels = int(n_labels if n_labels is not None else top + 1)
        self.c_max = max(len(v) for v in self.ids.values())
        self.group_of = {t: g for g, t in enumerate(self.tasks)}

    @classmethod
    def from_corpus(cls, corpus, tasks):
        by_task = {t: (corpus.indices_by_task(t) if hasattr(corpus, 'indices_by_task') else corpus._indices_by_task[t])
                   for t in tasks}
        return cls(by_task, corpus._background_indices, getattr(corpus, 'n_classes', None))

    def local_table(self, device):
        """global id -> local id per task (int32 [tasks, n_labels], -1 = not of the task), built once per device"""
        cache = self.__dict__.setdefault('_local_tables', {})
        tab = cache.get(str(device))
        if tab is None:
            host = np.full((len(self.tasks), self.n_labels), -1, dtype=np.int32)
            for g, t in enumerate(self.tasks):
                host[g, self.ids[t]] = np.arange(len(self.ids[t]), dtype=np.int32)
            tab = cache[str(device)] = torch.from_numpy(host).to(device)
        return tab

    def device_background(self, device):
        cache = self.__dict__.setdefault('_gbg_dev', {})
        t = cache.get(str(device))
        if t is None:
            t = cache[str(device)] = torch.from_numpy(self.static_tables()[1]).to(device)
        return t

    def static_tables(self):
        """host tables that do not depend on the predictions: per task the number of labels and which local ids are
        background (uint8 [tasks, c_max + 1])"""
        st = self.__dict__.get('_static')
        if st is None:
            w = self.c_max + 1
            n = np.array([len(self.ids[t]) for t in self.tasks], dtype=np.int64)
            gbg = np.zeros((len(self.tasks), w), dtype=np.uint8)
            for g, t in enumerate(self.tasks):
                gbg[g, :n[g]] = [i in self.background for i in self.ids[t]]
            st = self.__dict__['_static'] = (n, gbg)
        return st


def _pad_labels(present, size):
    """accuracy.py:243-261: invented labels fill the voting table up to a square (smallest free integer >= index)."""
    out = list(present)
    for idx in range(len(out), size):
        cand = idx
        while cand in out:
            cand += 1
        out.append(cand)
    return out


def assign_from_confusion(conf, ids, optimal):
    """``gt label -> [predicted label]`` from one task's confusion table (rows gt, columns predictions, local ids)."""
    n = len(ids)
    rows = [i for i in range(n) if conf[i, :].sum() > 0]
    cols = [j for j in range(n) if conf[:, j].sum() > 0]
    if not optimal:
        return {ids[i]: [ids[i]] for i in rows}
    size = max(len(rows), len(cols))
    gt_l = _pad_labels([ids[i] for i in rows], size)
    pr_l = _pad_labels([ids[j] for j in cols], size)
    table = np.zeros((size, size))
    if rows and cols:
        table[:len(rows), :len(cols)] = conf[np.ix_(rows, cols)]
    ri, ci = linear_sum_assignment(-table)
    return {gt_l[i]: [pr_l[j]] for i, j in zip(ri, ci)}


def _kernel_tables(space, g2c_by_task, device):
    """cluster_of / gt_is_bg / pred_is_bg in the kernels' (extended) local ids, plus the invented-label bookkeeping."""
    ng, w = len(space.tasks), space.c_max + 1
    cluster = np.full((ng, w), -1, dtype=np.int32)
    gbg = np.zeros((ng, w), dtype=np.uint8)
    pbg = np.zeros((ng, 2 * w), dtype=np.uint8)
    for g, t in enumerate(space.tasks):
        ids, g2c = space.ids[t], g2c_by_task[t]
        local = {i: l for l, i in enumerate(ids)}
        bkg_clusters = set(c for b in space.background for c in g2c.get(b, []))
        invented = {}
        for l, i in enumerate(ids):
            gbg[g, l] = i in space.background
            pbg[g, l] = i in bkg_clusters
            owned = g2c.get(i, [])
            if not owned:
                continue
            assert len(owned) == 1, "one cluster per ground-truth label (accuracy.py:14-19)"
            c = owned[0]
            if c in local:
                cluster[g, l] = local[c]
            else:
                if c not in invented:
                    invented[c] = w + len(invented)
                    assert invented[c] < 2 * w
                    pbg[g, invented[c]] = c in bkg_clusters
                cluster[g, l] = invented[c]
    to = lambda a: torch.from_numpy(a).to(device)
    return to(cluster), to(gbg), to(pbg)


def _finalise_identity(conf, ids, gbg, S, lev_sum, normed_sum, maxseg_sum, n_videos, want_extras):
    """``_finalise`` for the identity assignment (every ground-truth label that occurs is its own cluster), vectorised:
    the supervised evaluation of a whole corpus is 18 of these per call."""
    row_l, col_l, hit_l = conf.sum(1).tolist(), conf.sum(0).tolist(), np.diagonal(conf).tolist()
    bg_l = gbg.tolist()
    idx = [l for l, r in enumerate(row_l) if r > 0]                     # ground-truth labels that occur
    Sl = S.tolist()
    f = lambda key: float(Sl[CN[key]])
    frames = f('frames')
    stat = {}
    hit_sum = float(sum(hit_l[l] for l in idx))
    stat['mof'] = [hit_sum, frames]
    stat['mof_bg'] = [hit_sum, float(sum(row_l[l] for l in idx))]
    stat['mof_non_bg'] = [float(sum(hit_l[l] for l in idx if not bg_l[l])), float(sum(row_l[l] for l in idx if not bg_l[l]))]
    stat['precision'] = [f('tp'), frames]
    stat['recall'] = [f('tp'), f('gt_labels')]
    ratio = lambda p: p[0] / p[1] if p[1] else 0.0
    p_, r_ = ratio(stat['precision']), ratio(stat['recall'])
    stat['f1'] = [2 * p_ * r_ / (p_ + r_) if p_ + r_ > 0 else 0.0, 1.0]
    stat['precision_non_bg'] = [f('tp_non_bg'), f('frames_non_bg')]
    stat['recall_non_bg'] = [f('tp_non_bg'), f('gt_labels_non_bg')]
    pn, rn = ratio(stat['precision_non_bg']), ratio(stat['recall_non_bg'])
    stat['f1_non_bg'] = [2 * pn * rn / (pn + rn) if pn + rn > 0 else 0.0, 1.0]
    stat['true_background'] = [f('true_bg'), frames]
    stat['pred_background'] = [f('pred_bg'), frames]
    stat['iou_multi_non_bg'] = [f('iou_num'), f('iou_den')]
    stat['multiple_gt_labels'] = [f('multi'), frames]
    per_class = float(sum(hit_l[l] / (row_l[l] + col_l[l] - hit_l[l]) for l in idx))    # (the reference's summation order)
    stat['iou'] = [per_class, len(idx)]
    stat['iou_bg'] = [per_class, len(idx)]
    nv = float(n_videos)
    stat['mean_levenshtein'] = [lev_sum / nv, 1.0]
    stat['mean_max_segments'] = [maxseg_sum / nv, 1.0]
    stat['total_levenshtein'] = [float(lev_sum), 1.0]
    stat['num_videos'] = [nv, 1.0]
    stat['mean_normed_levenshtein'] = [normed_sum / nv, 1.0]
    stat['predicted_segments_per_video'] = [f('segs_pred'), nv]
    stat['predicted_segments_non_bg_per_video'] = [f('segs_pred_non_bg'), nv]
    stat['single_step_recall'] = [f('draw_hit'), f('steps')]
    stat['step_recall_non_bg'] = [f('draw_hit_non_bg'), f('steps_non_bg')]
    stat['center_step_recall'] = [f('mid_hit'), f('steps')]
    stat['center_step_recall_non_bg'] = [f('mid_hit_non_bg'), f('steps_non_bg')]
    stat['predicted_label_types_per_video'] = [f('types'), nv]
    stat['predicted_label_types_non_bg_per_video'] = [f('types_non_bg'), nv]
    if not want_extras:
        return stat, None
    return stat, dict(classes_mof={ids[l]: [float(hit_l[l]), int(row_l[l])] for l in idx},
                      classes_iou={ids[l]: [float(hit_l[l]), int(row_l[l] + col_l[l] - hit_l[l])] for l in idx},
                      gt2cluster={ids[l]: [ids[l]] for l in idx})


def _finalise(conf, ids, g2c, background, S, lev_sum, normed_sum, maxseg_sum, n_videos):
    """One task's ``Accuracy.stat()`` from its confusion table and summed counters (accuracy.py:500-521, 581-692)."""
    n = len(ids)
    local = {i: l for l, i in enumerate(ids)}
    row, col = conf.sum(1), conf.sum(0)
    cls_mof, cls_iou = {}, {}
    for l in range(n):
        if row[l] == 0:
            continue
        hit, union = 0.0, 0
        for c in g2c.get(ids[l], []):
            lc = local.get(c)
            h = float(conf[l, lc]) if lc is not None else 0.0
            hit += h
            union += int(row[l] + (col[lc] if lc is not None else 0) - h)
        cls_mof[ids[l]] = [hit, int(row[l])]
        cls_iou[ids[l]] = [hit, union]
    f = lambda key: float(S[CN[key]])
    frames = f('frames')
    stat = {}
    stat['mof'] = [sum(v[0] for v in cls_mof.values()), frames]
    stat['mof_bg'] = [sum(v[0] for v in cls_mof.values()), float(sum(v[1] for v in cls_mof.values()))]
    nb = [v for k, v in cls_mof.items() if k not in background]
    stat['mof_non_bg'] = [sum(v[0] for v in nb), float(sum(v[1] for v in nb))]
    stat['precision'] = [f('tp'), frames]
    stat['recall'] = [f('tp'), f('gt_labels')]
    ratio = lambda p: p[0] / p[1] if p[1] else 0.0
    p_, r_ = ratio(stat['precision']), ratio(stat['recall'])
    stat['f1'] = [2 * p_ * r_ / (p_ + r_) if p_ + r_ > 0 else 0.0, 1.0]
    stat['precision_non_bg'] = [f('tp_non_bg'), f('frames_non_bg')]
    stat['recall_non_bg'] = [f('tp_non_bg'), f('gt_labels_non_bg')]
    pn, rn = ratio(stat['precision_non_bg']), ratio(stat['recall_non_bg'])
    stat['f1_non_bg'] = [2 * pn * rn / (pn + rn) if pn + rn > 0 else 0.0, 1.0]
    stat['true_background'] = [f('true_bg'), frames]
    stat['pred_background'] = [f('pred_bg'), frames]
    stat['iou_multi_non_bg'] = [f('iou_num'), f('iou_den')]
    stat['multiple_gt_labels'] = [f('multi'), frames]
    per_class = sum(v[0] / v[1] for v in cls_iou.values())
    stat['iou'] = [per_class, len(cls_iou)]
    stat['iou_bg'] = [per_class, len(cls_iou)]
    nv = float(n_videos)
    stat['mean_levenshtein'] = [lev_sum / nv, 1.0]
    stat['mean_max_segments'] = [maxseg_sum / nv, 1.0]
    stat['total_levenshtein'] = [float(lev_sum), 1.0]
    stat['num_videos'] = [nv, 1.0]
    stat['mean_normed_levenshtein'] = [normed_sum / nv, 1.0]
    stat['predicted_segments_per_video'] = [f('segs_pred'), nv]
    stat['predicted_segments_non_bg_per_video'] = [f('segs_pred_non_bg'), nv]
    stat['single_step_recall'] = [f('draw_hit'), f('steps')]
    stat['step_recall_non_bg'] = [f('draw_hit_non_bg'), f('steps_non_bg')]
    stat['center_step_recall'] = [f('mid_hit'), f('steps')]
    stat['center_step_recall_non_bg'] = [f('mid_hit_non_bg'), f('steps_non_bg')]
    stat['predicted_label_types_per_video'] = [f('types'), nv]
    stat['predicted_label_types_non_bg_per_video'] = [f('types_non_bg'), nv]
    return stat, dict(classes_mof=cls_mof, classes_iou=cls_iou, gt2cluster=g2c)


def evaluate_labels(pred, gt, lengths, frame_offset, tasks, space, optimal_assignment=False, seed=0, video_key=None,
                    reduce=None, return_extras=False):
    """Statistics per task for labels on the packed frame axis.

    pred  cuda int64 [F];  gt  cuda int64 [F] or [F, W] (-1 padded);  lengths / frame_offset / tasks: per video (host);
    ``video_key``: index of each video inside its task (default: order of appearance);  ``reduce``: sums a tensor over
    the ranks of a distributed job (default: single process).
    """
    if not pred.is_cuda:
        raise _lib.SmmError("libsmmdp: evaluation counters run on the device (there is no CPU path)")
    gt2 = gt if gt.dim() == 2 else gt.unsqueeze(1)
    gt2 = gt2.contiguous()
    group = [space.group_of[t] for t in tasks]
    if video_key is None:
        seen, video_key = {}, []
        for g in group:
            video_key.append(seen.get(g, 0))
            seen[g] = video_key[-1] + 1
    empty = len(group) == 0                  # a rank of a sharded job that holds no video: zeros into the reductions
    local_of = space.local_table(pred.device)
    if empty:
        eb = None
        conf_d = torch.zeros((len(space.tasks), space.c_max + 1, space.c_max + 1), dtype=torch.int64, device=pred.device)
    else:
        eb = ops.EvalBatch(lengths, frame_offset, group, len(space.tasks), space.c_max, space.n_labels, gt2.size(1),
                           video_key=video_key, total_frames=pred.numel())
        conf_d = ops.eval_confusion(eb, pred, gt2, local_of)
    if reduce is not None:
        conf_d = reduce(conf_d)
    ng = len(space.tasks)
    n_ids, gbg_h = space.static_tables()
    per_video_d = None
    if not optimal_assignment and not empty:
        # identity assignment: every ground-truth label that occurs is its own cluster.  The cluster tables come from
        # the confusion table ON THE DEVICE, so the second kernel follows the first without a trip to the host.
        w = space.c_max + 1
        occ_d = conf_d[:, :w, :w].sum(2) > 0
        occ_d[:, -1] = False
        gbg_d = space.device_background(pred.device)
        cluster_d = torch.where(occ_d, torch.arange(w, dtype=torch.int32, device=pred.device).unsqueeze(0),
                                torch.full((), -1, dtype=torch.int32, device=pred.device)).contiguous()
        pbg_d = torch.zeros((ng, 2 * w), dtype=torch.uint8, device=pred.device)
        pbg_d[:, :w] = gbg_d * occ_d.to(torch.uint8)
        per_video_d = ops.eval_videos(eb, pred, gt2, local_of, cluster_d, gbg_d, pbg_d, seed=seed)
    conf = conf_d.cpu().numpy()
    if conf[:, :, -1].sum() or conf[:, -1, :].sum():
        raise ValueError("labels outside their task's class set: %d predicted, %d ground-truth frames"
                         % (conf[:, :, -1].sum(), conf[:, -1, :].sum()))
    if optimal_assignment:
        g2c = {}
        for g, t in enumerate(space.tasks):
            n = len(space.ids[t])
            g2c[t] = assign_from_confusion(conf[g, :n, :n], space.ids[t], optimal_assignment)
            n_pred = int((conf[g, :n, :n].sum(0) > 0).sum())
            assert n_pred <= n, "more predicted labels than the task has classes (accuracy.py:349-352)"
        cluster, gbg, pbg = _kernel_tables(space, g2c, pred.device)
    if empty:
        per_video = np.zeros((0, _lib.EVAL_COUNTERS), dtype=np.int64)
    elif per_video_d is not None:
        per_video = per_video_d.cpu().numpy()
    else:
        per_video = ops.eval_videos(eb, pred, gt2, local_of, cluster, gbg, pbg, seed=seed).cpu().numpy()

    # additive per-task sums (one small matrix product): integer counters, then the three per-video quantities of levenshtein()
    sums = np.zeros((ng, _lib.EVAL_COUNTERS + 4), dtype=np.float64)
    grp = np.asarray(group, dtype=np.int64)
    if per_video.shape[0]:
        member = np.zeros((ng, per_video.shape[0]), dtype=np.float64)
        member[grp, np.arange(per_video.shape[0])] = 1.0
        lev = per_video[:, CN['levenshtein']].astype(np.float64)
        longest = np.maximum(per_video[:, CN['segs_gt']], per_video[:, CN['segs_pred']]).astype(np.float64)
        sums[:, :_lib.EVAL_COUNTERS] = member @ per_video.astype(np.float64)          # (counts < 2^53: exact)
        sums[:, _lib.EVAL_COUNTERS:] = member @ np.stack([lev, lev / longest, longest, np.ones_like(lev)], axis=1)
    if reduce is not None:
        sums = reduce(torch.from_numpy(sums).to(pred.device)).cpu().numpy()
    out, extras = {}, {}
    for g, t in enumerate(space.tasks):
        if sums[g, -1] == 0:
            continue
        n = int(n_ids[g])
        lev_sum, normed_sum, maxseg_sum, nv = sums[g, _lib.EVAL_COUNTERS:]
        if optimal_assignment:
            out[t], extras[t] = _finalise(conf[g, :n, :n], space.ids[t], g2c[t], space.background,
                                          sums[g, :_lib.EVAL_COUNTERS], lev_sum, normed_sum, maxseg_sum, nv)
        else:
            out[t], extras[t] = _finalise_identity(conf[g, :n, :n], space.ids[t], gbg_h[g, :n], sums[g, :_lib.EVAL_COUNTERS],
                                                   lev_sum, normed_sum, maxseg_sum, nv, return_extras)
        if return_extras:
            extras[t]['per_video'] = per_video[grp == g]
    return (out, extras) if return_extras else out


def _ground_truth(sample):
    """Full-rate ground truth of one video as int64 [T, W], -1 padded: ``gt`` (per frame a list of labels, the
    reference's ``Video.gt()``; or a [T, W] tensor) when the sample has it, else the single-label sequence."""
    gt = sample.get('gt')
    if gt is not None:
        if torch.is_tensor(gt):
            return gt.to(torch.int64).view(gt.size(0), -1)
        width = max(len(f) for f in gt)
        arr = np.full((len(gt), width), -1, dtype=np.int64)
        for i, f in enumerate(gt):
            arr[i, :len(f)] = f
        return torch.from_numpy(arr)
    single = sample.get('gt_single_unsampled')
    if single is None:
        single = sample['gt_single']
    return torch.as_tensor(single).to(torch.int64).view(-1, 1)


def accuracy_corpus(data, predictions, optimal_assignment, seed=0, reduce=None, device=None):
    """``Datasplit.accuracy_corpus`` (corpus.py:405-604): ``{task: stat dict}``.

    ``data``: a datasplit with ``_videos_by_task`` / ``_videos`` / ``corpus``; ``predictions``: ``{video_name:
    int64[T]}`` as ``SemiMarkovModel.predict`` returns (numpy or tensors), uploaded once and counted on the device.
    What the reference does on top of ``Accuracy`` is done here too, on the device:
      * multi-label ground truth (``video.gt()``: the sample's ``gt`` entry; the first label is "the" label);
      * ``data.subsample != 1`` (--frame_subsample): predictions made on every subsample-th frame are repeated back to
        the frame rate, ``np.array(pred + [pred[-1]]).repeat(subsample)[:len(gt)]`` (:466-472);
      * ``corpus.annotate_background_with_previous``: every background id of the corpus, in ground truth and predictions,
        becomes the corpus' first background id (:474-480, ``canonicalize_background`` :399-403).
    With ``reduce`` (multi-process: ``distributed.all_reduce_tensor``) ``predictions`` may hold only this rank's videos;
    the counters are summed over ranks before anything is finalised, so every rank returns the corpus statistics
    (``main.py:486-532`` sums the same pairs over tasks).
    """
    device = device or torch.device('cuda', torch.cuda.current_device())
    tasks = list(data._videos_by_task)
    corpus = data.corpus
    subsample = int(getattr(data, 'subsample', 1) or 1)
    canonical = bool(getattr(corpus, 'annotate_background_with_previous', False)) and len(corpus._background_indices) > 0
    by_task = {t: (corpus.indices_by_task(t) if hasattr(corpus, 'indices_by_task') else corpus._indices_by_task[t])
               for t in tasks}
    if canonical:
        bkg0 = int(corpus._background_indices[0])
        by_task = {t: sorted(set(int(i) for i in ids) | {bkg0}) for t, ids in by_task.items()}
    space = LabelSpace(by_task, corpus._background_indices, getattr(corpus, 'n_classes', None))
    lengths, offsets, task_of, preds, gts, keys = [], [], [], [], [], []
    off = 0
    for task in tasks:
        for key, name in enumerate(data._videos_by_task[task]):
            if reduce is not None and name not in predictions:
                continue                                   # another rank's video
            keys.append(key)                               # index inside the task: seeds the step-recall draw
            gt = _ground_truth(data._videos[(task, name)])
            pr = torch.as_tensor(predictions[name]).to(torch.int64).view(-1).to(device, non_blocking=True)
            if subsample != 1:
                pr = torch.cat([pr, pr[-1:]]).repeat_interleave(subsample)[:gt.size(0)]
            assert gt.size(0) == pr.numel(), "%s: %d ground-truth vs %d predicted frames" % (name, gt.size(0), pr.numel())
            lengths.append(int(gt.size(0)))
            offsets.append(off)
            task_of.append(task)
            preds.append(pr)
            gts.append(gt.to(device, non_blocking=True))
            off += lengths[-1]
    if not preds:                                          # a rank without videos still takes part in the reductions
        preds, gts = [torch.zeros(1, dtype=torch.int64, device=device)], [torch.zeros((1, 1), dtype=torch.int64, device=device)]
        lengths, offsets, task_of, keys = [], [], [], []
    width = max(g.size(1) for g in gts)
    gts = [g if g.size(1) == width else torch.nn.functional.pad(g, (0, width - g.size(1)), value=-1) for g in gts]
    pred, gt = torch.cat(preds), torch.cat(gts)
    if canonical:
        table = torch.arange(max(space.n_labels, int(pred.max()) + 1 if pred.numel() else 1), dtype=torch.int64, device=device)
        table[torch.as_tensor([int(b) for b in corpus._background_indices], device=device)] = bkg0
        pred = table[pred]
        gt = torch.where(gt >= 0, table[gt.clamp(min=0)], gt)
    return evaluate_labels(pred, gt, lengths, offsets, task_of, space, optimal_assignment, seed=seed, video_key=keys,
                           reduce=reduce)


def summarise(stats_by_task, keys, prefix=''):
    """Corpus-level ratios the way ``main.py:189-194`` forms them: sum numerators and denominators over tasks."""
    out = {}
    for key in keys:
        tot = np.sum([np.asarray(s[key], dtype=np.float64) for s in stats_by_task.values()], axis=0)
        out[prefix + key] = float(tot[0]) / float(tot[1])
    return out


STAT_KEYS = ['mof', 'mof_non_bg', 'step_recall_non_bg', 'mean_normed_levenshtein', 'center_step_recall_non_bg', 'f1',
             'f1_non_bg', 'pred_background', 'iou_multi_non_bg', 'predicted_label_types_per_video',
             'predicted_label_types_non_bg_per_video', 'predicted_segments_per_video',
             'predicted_segments_non_bg_per_video', 'multiple_gt_labels']          # main.py:20-26


class Accuracy:
    """The reference's per-task accumulator with the same call sequence; counted on the device at ``mof()`` time.

    ``corpus`` needs ``_background_indices``.  Labels are python / numpy sequences as in the reference:
    ground truth = per frame a list of labels, predictions = per frame one label.
    """

    def __init__(self, n_frames=1, verbose=True, corpus=None, seed=0, device=None):
        self._corpus = corpus
        self._seed = seed
        self._device = device
        self._gt, self._pred = [], []
        self._return = {}
        self._all = None
        self._gt2cluster = {}
        self._classes_MoF, self._classes_IoU = {}, {}
        self._frames_true_pr = self._frames_overall = 0

    def add_gt_labels(self, labels):
        assert isinstance(labels, list) and isinstance(labels[0], (list, tuple))
        self._gt.append(labels)

    def add_predicted_labels(self, labels):
        self._pred.append([int(x) for x in labels])

    def mof(self, optimal_assignment, with_segments=False, optimization='max', possible_gt_labels=None):
        assert not with_segments and optimization == 'max', "frame sub-sampling is not part of this path"
        assert len(self._gt) == len(self._pred)
        device = self._device or torch.device('cuda', torch.cuda.current_device())
        width = max(len(f) for v in self._gt for f in v)
        lengths = [len(v) for v in self._gt]
        gt = np.full((sum(lengths), width), -1, dtype=np.int64)
        row = 0
        for v in self._gt:
            for f in v:
                gt[row, :len(f)] = f
                row += 1
        pred = np.concatenate([np.asarray(v, dtype=np.int64) for v in self._pred])
        assert pred.shape[0] == gt.shape[0]
        present = set(int(x) for x in np.unique(gt[gt >= 0])) | set(int(x) for x in np.unique(pred))
        ids = sorted(present | set(int(x) for x in (possible_gt_labels if possible_gt_labels is not None else [])))
        space = LabelSpace({'task': ids}, self._corpus._background_indices)
        offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]])
        stats, extras = evaluate_labels(torch.from_numpy(pred).to(device), torch.from_numpy(gt).to(device), lengths,
                                        offsets, ['task'] * len(lengths), space, optimal_assignment, seed=self._seed,
                                        return_extras=True)
        self._all, ex = stats['task'], extras['task']
        self._gt2cluster = ex['gt2cluster']
        self._classes_MoF, self._classes_IoU = ex['classes_mof'], ex['classes_iou']
        self._frames_true_pr, self._frames_overall = self._all['mof'][0], int(self._all['mof'][1])
        return self._frames_overall

    def _take(self, keys):
        for k in keys:
            self._return[k] = np.asarray(self._all[k], dtype=np.float64)

    def mof_classes(self):
        self._take(['mof', 'mof_bg', 'mof_non_bg', 'precision', 'recall', 'f1', 'precision_non_bg', 'recall_non_bg',
                    'f1_non_bg', 'true_background', 'pred_background', 'iou_multi_non_bg', 'multiple_gt_labels'])

    def iou_classes(self):
        self._take(['iou', 'iou_bg'])

    def levenshtein(self, gt2cluster=None):
        self._take(['mean_levenshtein', 'mean_max_segments', 'total_levenshtein', 'num_videos',
                    'mean_normed_levenshtein', 'predicted_segments_per_video', 'predicted_segments_non_bg_per_video'])

    def single_step_recall(self, gt2cluster=None):
        self._take(['single_step_recall', 'step_recall_non_bg', 'center_step_recall', 'center_step_recall_non_bg',
                    'predicted_label_types_per_video', 'predicted_label_types_non_bg_per_video'])

    def mof_val(self):
        return float(self._frames_true_pr) / self._frames_overall

    def frames(self):
        return self._frames_true_pr

    def stat(self):
        return self._return
