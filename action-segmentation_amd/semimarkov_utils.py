"""Span/label codecs and closed-form sufficient statistics.

Mirrors reference ``src/models/semimarkov/semimarkov_utils.py`` (same function names and conventions):
``labels_to_spans`` :6, ``rle_spans`` :26, ``spans_to_labels`` :51, ``semimarkov_sufficient_stats`` :74.
The codecs are vectorised host code (the HIP decode kernel emits span encodings and frame labels itself);
``semimarkov_sufficient_stats`` is the host statement the reference has, ``semimarkov_sufficient_stats_device`` the same
statistics from one HIP pass over device-resident features (``csrc/smm_fit.hip``), used whenever the module lives on
the GPU.
"""
import numpy as np
import torch


def labels_to_spans(position_labels, max_k):
    """b x N labels -> span encoding: class id at a span start, -1 for its continuation.  A run of one label is
    cut every ``max_k - 1`` frames (the DP only knows segment lengths 1..max_k-1).  utils.py:6-23."""
    if torch.is_tensor(position_labels) and position_labels.is_cuda:
        # device tensors stay on the device (the gradient-based supervised fit encodes every batch, semimarkov.py:251)
        lab = position_labels.detach()
        b, n = lab.shape
        idx = torch.arange(n, device=lab.device).unsqueeze(0).expand(b, n)
        change = torch.ones((b, n), dtype=torch.bool, device=lab.device)
        change[:, 1:] = lab[:, 1:] != lab[:, :-1]
        run_start = torch.cummax(torch.where(change, idx, torch.zeros_like(idx)), dim=1).values
        pos_in_run = idx - run_start
        cont = (pos_in_run % max(max_k - 1, 1) != 0) if max_k is not None else (pos_in_run != 0)
        return torch.where(cont, torch.full_like(lab, -1), lab)
    lab = position_labels.detach().cpu().numpy() if torch.is_tensor(position_labels) else np.asarray(position_labels)
    assert not (lab == -1).any(), "position_labels already appear span encoded (have -1)"
    b, n = lab.shape
    out = lab.copy()
    if n > 1:
        change = np.ones((b, n), dtype=bool)
        change[:, 1:] = lab[:, 1:] != lab[:, :-1]
        idx = np.arange(n)[None, :].repeat(b, 0)
        run_start = np.maximum.accumulate(np.where(change, idx, 0), axis=1)
        pos_in_run = idx - run_start
        if max_k is not None:
            cont = pos_in_run % max(max_k - 1, 1) != 0
        else:
            cont = pos_in_run != 0
        out[cont] = -1
    if torch.is_tensor(position_labels):
        return torch.from_numpy(out).to(position_labels.device)
    return out


def spans_to_labels(spans):
    """Forward-fill -1 with the running label.  utils.py:51-63."""
    if torch.is_tensor(spans) and spans.is_cuda:
        sp = spans.detach()
        n = sp.size(1)
        idx = torch.where(sp != -1, torch.arange(n, device=sp.device).unsqueeze(0).expand_as(sp), torch.zeros_like(sp))
        return torch.gather(sp, 1, torch.cummax(idx, dim=1).values)
    sp = spans.detach().cpu().numpy() if torch.is_tensor(spans) else np.asarray(spans)
    assert (sp[:, 0] != -1).all()
    b, n = sp.shape
    idx = np.where(sp != -1, np.arange(n)[None, :], 0)
    last = np.maximum.accumulate(idx, axis=1)
    out = np.take_along_axis(sp, last, axis=1)
    if torch.is_tensor(spans):
        return torch.from_numpy(out).to(spans.device)
    return out


def rle_spans(spans, lengths):
    """[(symbol, run length), ...] per instance.  utils.py:26-48."""
    sp = spans.detach().cpu().numpy() if torch.is_tensor(spans) else np.asarray(spans)
    res = []
    for i in range(sp.shape[0]):
        row = sp[i, :int(lengths[i])]
        starts = np.flatnonzero(row != -1)
        if len(row) and (len(starts) == 0 or starts[0] != 0):
            starts = np.concatenate([[0], starts])
        ends = np.concatenate([starts[1:], [len(row)]])
        rle = [(int(row[s]), int(e - s)) for s, e in zip(starts, ends)]
        assert sum(c for _, c in rle) == int(lengths[i])
        res.append(rle)
    return res


def semimarkov_sufficient_stats(feature_list, label_list, covariance_type, n_classes, max_k=None):
    """Closed-form statistics of reference utils.py:74-126 without the sklearn dependency.

    Returns (emission_stats, span_stats): emission_stats has ``means_`` (n_classes x D, one-hot-responsibility
    means, nk = count + 10*eps as sklearn's ``_estimate_gaussian_parameters``) and ``covariances_`` (n_classes x D
    rows of the tied GLOBAL biased per-dim variance + 1e-6 for 'tied_diag').
    """
    assert len(feature_list) == len(label_list)
    assert covariance_type == 'tied_diag', "only the reference's tied diagonal covariance is built"
    span_counts = np.zeros(n_classes, dtype=np.float32)
    span_lengths = np.zeros(n_classes, dtype=np.float32)
    span_start_counts = np.zeros(n_classes, dtype=np.float32)
    span_transition_counts = np.zeros((n_classes, n_classes), dtype=np.float32)   # to, from
    d = int(feature_list[0].shape[1])
    sum_x = np.zeros((n_classes, d))
    cnt = np.zeros(n_classes)
    tot = np.zeros(d)
    tot2 = np.zeros(d)
    n_all = 0
    for x, labels in zip(feature_list, label_list):
        x = x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
        y = labels.detach().cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)
        x = x.astype(np.float64)
        rle = rle_spans(labels_to_spans(y[None], max_k), [y.shape[0]])[0]
        syms = np.array([s for s, _ in rle])
        lens = np.array([c for _, c in rle], dtype=np.float32)
        span_start_counts[syms[0]] += 1
        np.add.at(span_counts, syms, 1)
        np.add.at(span_lengths, syms, lens)
        np.add.at(span_transition_counts, (syms[1:], syms[:-1]), 1)
        np.add.at(sum_x, y, x)
        np.add.at(cnt, y, 1)
        tot += x.sum(0)
        tot2 += (x * x).sum(0)
        n_all += x.shape[0]
    eps10 = 10 * np.finfo(np.float64).eps
    means = sum_x / (cnt + eps10)[:, None]
    gmean = tot / (n_all + eps10)
    var = tot2 / (n_all + eps10) - gmean ** 2 + 1e-6

    class _Emissions:
        pass
    em = _Emissions()
    em.means_ = means
    em.covariances_ = np.tile(var[None], (n_classes, 1))
    return em, {
        'span_counts': span_counts,
        'span_lengths': span_lengths,
        'span_start_counts': span_start_counts,
        'span_transition_counts': span_transition_counts,
        'instance_count': len(feature_list),
    }


class _Emissions:
    pass


def _stats_from_sums(sum_x, sum_x2, cnt, n_all, n_classes):
    """sklearn's one-hot-responsibility estimates (nk = count + 10 eps) and the tied diagonal variance (+1e-6)."""
    eps10 = 10 * np.finfo(np.float64).eps
    em = _Emissions()
    em.means_ = sum_x / (cnt + eps10)[:, None]
    gmean = sum_x.sum(0) / (n_all + eps10)
    var = sum_x2 / (n_all + eps10) - gmean ** 2 + 1e-6
    em.covariances_ = np.tile(var[None], (n_classes, 1))
    return em


def semimarkov_sufficient_stats_device(feature_list, label_list, covariance_type, n_classes, max_k=None, device=None):
    """``semimarkov_sufficient_stats`` computed on the GPU (no host pass over the features; no CPU fallback).

    ``feature_list`` / ``label_list``: per video T x D fp32 and T int64 tensors (moved to ``device`` if they are not
    there yet).  Same return value as the host function.
    """
    from . import ops
    assert len(feature_list) == len(label_list)
    assert covariance_type == 'tied_diag', "only the reference's tied diagonal covariance is built"
    device = device or torch.device('cuda', torch.cuda.current_device())
    lengths = [int(f.shape[0]) for f in feature_list]
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    x = torch.cat([torch.as_tensor(f).to(device=device, dtype=torch.float32) for f in feature_list]).contiguous()
    y = torch.cat([torch.as_tensor(l).to(device=device, dtype=torch.int64) for l in label_list]).contiguous()
    assert x.shape[0] == y.shape[0]
    out = ops.fit_stats(x, y, lengths, offsets, n_classes, max_k)
    if int(out['_err'].item()) != 0:
        raise ValueError("labels outside [0, %d)" % n_classes)
    host = {k: v.cpu().numpy() for k, v in out.items() if not k.startswith('_')}
    cnt = host['frame_counts'].astype(np.float64)
    em = _stats_from_sums(host['sum_x'], host['sum_x2'], cnt, float(x.shape[0]), n_classes)
    return em, {
        'span_counts': host['span_counts'].astype(np.float32),
        'span_lengths': host['frame_counts'].astype(np.float32),      # every frame lies in exactly one span of its label
        'span_start_counts': host['span_start_counts'].astype(np.float32),
        'span_transition_counts': host['span_transition_counts'].astype(np.float32),
        'instance_count': len(feature_list),
    }
