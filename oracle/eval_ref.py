"""CPU restatement of the reference's per-task evaluation counters -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module; the product
(``action_segmentation_amd.evaluation``) computes the same numbers from device-side counters and never calls it.

Follows ``src/evaluation/accuracy.py`` of the reference as driven by ``Datasplit.accuracy_corpus``
(``src/data/corpus.py:486-565``):  label assignment (:232-318, :334-362), frame counters of ``mof`` (:475-579),
``mof_classes`` (:581-660), ``iou_classes`` (:662-692), ``levenshtein`` (:364-408), ``single_step_recall`` (:410-472).
Pinned against ``tests/golden/eval_vectors.json`` (outputs of the reference's own ``Accuracy``).

Deliberate departures (the first two are stated in the golden script as well; ``f1`` is 0 where the reference's
``2pr/(p+r)`` divides by zero, :627):
* an empty cluster list never matches a prediction (the reference's ``x in [[..], []]`` under the numpy of its day);
* the random frame of ``single_step_recall`` (``np.random.choice``, unseeded in the reference, :449) is the frame with
  the smallest ``frame_hash(seed, video, t)`` among the candidates -- a uniform draw that the device can repeat.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

M32 = 0xFFFFFFFF


def frame_hash(seed, video, t):
    """32-bit avalanche of (seed, video index within the task, frame); identical arithmetic in csrc/smm_eval.hip."""
    x = (seed ^ (video * 0x9E3779B1) ^ (t * 0x85EBCA77)) & M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def edit_distance(a, b):
    """Levenshtein distance between two label sequences (what the reference gets from ``editdistance.eval``)."""
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (0 if a[i - 1] == b[j - 1] else 1))
        prev = cur
    return prev[-1]


def run_lengths(seq):
    """[(label, length), ...] of maximal constant runs (accuracy.py:21-37)."""
    out = []
    for x in seq:
        if out and out[-1][0] == x:
            out[-1][1] += 1
        else:
            out.append([x, 1])
    return [(a, n) for a, n in out]


def assign(gt_first_all, pred_all, optimal):
    """gt label -> list of predicted labels it owns ({} entries only where the list is non-empty).

    identity (:316-318): every ground-truth label present maps to itself.
    Hungarian (:232-307): square voting table over the sorted distinct labels of both sides, padded with invented
    labels (the smallest integers >= the row index that are still free), maximum-weight perfect matching.
    """
    gt_u = sorted(set(int(x) for x in gt_first_all))
    pr_u = sorted(set(int(x) for x in pred_all))
    if not optimal:
        return {g: [g] for g in gt_u}, None

    def padded(labels, size):
        labels = list(labels)
        for idx in range(len(labels), size):
            cand = idx
            while cand in labels:
                cand += 1
            labels.append(cand)
        return labels

    size = max(len(gt_u), len(pr_u))
    rows, cols = padded(gt_u, size), padded(pr_u, size)
    g = np.asarray(gt_first_all)
    p = np.asarray(pred_all)
    table = np.zeros((size, size))
    for i, gl in enumerate(gt_u):
        for j, pl in enumerate(pr_u):
            table[i, j] = float(np.sum((g == gl) & (p == pl)))
    ri, ci = linear_sum_assignment(-table)
    return {rows[i]: [cols[j]] for i, j in zip(ri, ci)}, table


def task_counters(gt, pred, background, possible=None, optimal=False, seed=0):
    """All ``[numerator, denominator]`` pairs of one task.

    gt:   per video, per frame, a non-empty list of labels (the first one is "the" label).
    pred: per video, per frame, one label.
    Returns (stat dict keyed like ``Accuracy.stat()``, extras dict with gt2cluster / per-class tables).
    """
    background = set(int(b) for b in background)
    gt_first = [[int(f[0]) for f in v] for v in gt]
    flat_gt = [x for v in gt_first for x in v]
    flat_pr = [int(x) for v in pred for x in v]
    assert len(flat_gt) == len(flat_pr)
    g2c, table = assign(flat_gt, flat_pr, optimal)
    if possible is None:
        possible = set(flat_gt)
    assert len(set(flat_pr)) <= len(set(possible))
    owned = lambda lab: g2c.get(int(lab), [])
    bkg_clusters = set(c for b in background for c in owned(b))        # predicted labels that mean "background"

    G, P = np.asarray(flat_gt), np.asarray(flat_pr)
    cls_mof, cls_iou, true_total = {}, {}, 0.0
    for gl in sorted(set(flat_gt)):
        hit, union = 0.0, 0
        for cl in owned(gl):
            hit += float(np.sum((G == gl) & (P == cl)))
            union += int(np.sum((G == gl) | (P == cl)))
        cls_mof[gl] = [hit, int(np.sum(G == gl))]
        cls_iou[gl] = [hit, union]
        true_total += hit

    prec, rec = [0.0, 0.0], [0.0, 0.0]
    prec_nb, rec_nb = [0.0, 0.0], [0.0, 0.0]
    true_bg, pred_bg = [0.0, 0.0], [0.0, 0.0]
    iou_nb, multi = [0.0, 0.0], [0.0, 0.0]
    for v_gt, v_pr in zip(gt, pred):
        for labs, p in zip(v_gt, v_pr):
            labs = [int(x) for x in labs]
            p = int(p)
            multi[1] += 1
            multi[0] += len(labs) > 1
            rec[1] += len(labs)
            prec[1] += 1
            tp = any(p in owned(x) for x in labs)
            if tp:
                rec[0] += 1
                prec[0] += 1
            true_bg[1] += 1
            pred_bg[1] += 1
            p_is_bg = p in bkg_clusters
            pred_bg[0] += p_is_bg
            is_bg = any(x in background for x in labs)
            if is_bg:
                assert all(x in background for x in labs)
            if not (is_bg and p_is_bg):
                iou_nb[1] += 1
                iou_nb[0] += tp
            if is_bg:
                true_bg[0] += 1
            else:
                rec_nb[1] += len(labs)
                prec_nb[1] += 1
                if tp:
                    rec_nb[0] += 1
                    prec_nb[0] += 1

    stat = {}
    n_frames = len(flat_gt)
    stat['mof'] = [true_total, n_frames]
    stat['mof_bg'] = [sum(v[0] for v in cls_mof.values()), sum(v[1] for v in cls_mof.values())]
    nb = [v for k, v in cls_mof.items() if k not in background]
    stat['mof_non_bg'] = [sum(v[0] for v in nb), sum(v[1] for v in nb)]
    stat['precision'], stat['recall'] = prec, rec
    ratio = lambda pair: pair[0] / pair[1] if pair[1] else 0.0
    pr_, rc_ = ratio(prec), ratio(rec)
    stat['f1'] = [2 * pr_ * rc_ / (pr_ + rc_) if pr_ + rc_ > 0 else 0.0, 1.0]    # the reference divides by zero here
    stat['precision_non_bg'], stat['recall_non_bg'] = prec_nb, rec_nb
    pn, rn = ratio(prec_nb), ratio(rec_nb)
    stat['f1_non_bg'] = [0.0 if pn == 0 and rn == 0 else 2 * pn * rn / (pn + rn), 1.0]
    stat['true_background'], stat['pred_background'] = true_bg, pred_bg
    stat['iou_multi_non_bg'], stat['multiple_gt_labels'] = iou_nb, multi
    per_class_iou = sum(v[0] / v[1] for v in cls_iou.values())
    stat['iou'] = [per_class_iou, len(cls_iou)]
    stat['iou_bg'] = [per_class_iou, len(cls_iou)]

    # segment-level: edit distance of the run-length encoded label sequences, per video
    lev, longest, n_seg, n_seg_nb = [], [], 0.0, 0.0
    for v_gt, v_pr in zip(gt_first, pred):
        a = [owned(lab)[0] for lab, _ in run_lengths(v_gt)]            # singleton_lookup: must exist and be unique
        b = [int(lab) for lab, _ in run_lengths([int(x) for x in v_pr])]
        n_seg += len(b)
        n_seg_nb += sum(1 for x in b if x not in bkg_clusters)
        lev.append(edit_distance(a, b))
        longest.append(max(len(a), len(b)))
    lev, longest = np.asarray(lev, dtype=np.float64), np.asarray(longest, dtype=np.float64)
    nv = len(gt)
    stat['mean_levenshtein'] = [float(np.mean(lev)), 1.0]
    stat['mean_max_segments'] = [float(np.mean(longest)), 1.0]
    stat['total_levenshtein'] = [float(np.sum(lev)), 1.0]
    stat['num_videos'] = [float(nv), 1.0]
    stat['mean_normed_levenshtein'] = [float(np.mean(lev / longest)), 1.0]
    stat['predicted_segments_per_video'] = [n_seg, float(nv)]
    stat['predicted_segments_non_bg_per_video'] = [n_seg_nb, float(nv)]

    # step recall: for every (remapped) ground-truth label of a video, look at ONE frame predicted as that label
    tot = tot_nb = hit = hit_nb = mid = mid_nb = types = types_nb = 0.0
    for vi, (v_gt, v_pr) in enumerate(zip(gt_first, pred)):
        v_pr = np.asarray([int(x) for x in v_pr])
        remap = [owned(x)[0] for x in v_gt]
        for lab in sorted(set(v_pr.tolist())):
            types += 1
            types_nb += lab not in bkg_clusters
        for lab in sorted(set(remap)):
            not_bg = lab not in bkg_clusters
            tot += 1
            tot_nb += not_bg
            where = np.flatnonzero(v_pr == lab)
            if len(where) == 0:
                continue
            draw = min(where.tolist(), key=lambda t: (frame_hash(seed, vi, t), t))
            centre = min(where.tolist(), key=lambda t: (abs(2 * t - (int(where[0]) + int(where[-1]))), t))
            if remap[draw] == lab:
                hit += 1
                hit_nb += not_bg
            if remap[centre] == lab:
                mid += 1
                mid_nb += not_bg
    stat['single_step_recall'] = [hit, tot]
    stat['step_recall_non_bg'] = [hit_nb, tot_nb]
    stat['center_step_recall'] = [mid, tot]
    stat['center_step_recall_non_bg'] = [mid_nb, tot_nb]
    stat['predicted_label_types_per_video'] = [types, float(nv)]
    stat['predicted_label_types_non_bg_per_video'] = [types_nb, float(nv)]
    extras = dict(gt2cluster=g2c, classes_mof=cls_mof, classes_iou=cls_iou, frames=n_frames,
                  levenshtein=lev, max_segments=longest)
    return stat, extras


def datasplit_counters(tasks, background, videos, subsample=1, annotate_background_with_previous=False, optimal=False,
                       seed=0):
    """``Datasplit.accuracy_corpus`` (src/data/corpus.py:405-604) on top of ``task_counters``: per task,
    multi-label ground truth, predictions made on every ``subsample``-th frame repeated back to the frame rate
    (``np.array(pred + [pred[-1]]).repeat(subsample)[:len(gt)]``, :466-472) and, under
    ``annotate_background_with_previous``, EVERY background id of the corpus -- in ground truth and predictions --
    replaced by the corpus' first background id (:474-480, ``canonicalize_background`` :399-403).

    tasks: {task: class ids}; videos: {task: {name: dict(gt=per-frame label lists, pred=labels)}} in the datasplit's
    order.  Returns {task: stat dict}."""
    background = [int(b) for b in background]
    canon = (lambda x: background[0] if int(x) in background else int(x)) if annotate_background_with_previous else int
    out = {}
    for task, vids in videos.items():
        gts, preds = [], []
        for name, v in vids.items():
            gt = [[canon(x) for x in f] for f in v['gt']]
            pred = [int(x) for x in v['pred']]
            if subsample != 1:
                pred = list(np.array(pred + [pred[-1]]).repeat(subsample)[:len(gt)])
                assert len(pred) == len(gt), (name, len(pred), len(gt))
            gts.append(gt)
            preds.append([canon(x) for x in pred])
        out[task], _ = task_counters(gts, preds, background, tasks[task], optimal, seed=seed)
    return out
