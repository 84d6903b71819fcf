"""Dense CPU oracle: span scoring + sequential semiring DP (TEST INFRASTRUCTURE ONLY).

A from-scratch torch/numpy restatement of the reference hot path.  Citations
are to ``/root/reference`` (file:line) or, for the absent third-party DP, to the
published algorithm of ``harvardnlp/pytorch-struct@1c9b038a`` (``torch_struct``,
``SemiMarkov``), which the reference calls at
``src/models/semimarkov/semimarkov_modules.py:624,641,646,651-657,677,679``.

Everything here works in the dtype of its inputs (float32 like the reference, or
float64 for the stable oracle the HIP path is compared with).

Parity status (see oracle/__init__.py): scoring half pinned by golden vectors
from the reference's own code; DP half: **numeric parity unpinned** (pinned only
by the reference's structural known answer and brute-force enumeration).
"""
import math

import numpy as np
import torch

BIG_NEG = -1e9  # semimarkov_modules.py:20


# --------------------------------------------------------------------------- scorers

def emission_log_probs(features, means, cov_diag, constraints=None):
    """Diagonal-Gaussian log density of every frame under every state.

    Follows semimarkov_modules.py:324-381 (MultivariateNormal with
    scale_tril = sqrt(diagonal covariance)):  elp = -0.5*(D*log(2*pi) + M) - sum(log sigma)
    with M = sum_d ((x_d - mu_cd)/sigma_d)^2.
    features b x T x D, means C x D, cov_diag D  ->  b x T x C.
    """
    sigma = cov_diag.sqrt()
    d = features.shape[-1]
    z = (features.unsqueeze(2) - means.view(1, 1, *means.shape)) / sigma.view(1, 1, 1, -1)
    m = (z * z).sum(-1)
    elp = -0.5 * (d * math.log(2 * math.pi) + m) - sigma.log().sum()
    if constraints is not None:
        elp = elp + constraints  # :379-380
    return elp


def initial_log_probs(init_logits, init_constraints=None, valid_classes=None):
    """semimarkov_modules.py:284-296 -- mask (True = forbidden) BEFORE the softmax."""
    logits = init_logits
    if init_constraints is not None:
        logits = torch.where(init_constraints, torch.full_like(logits, BIG_NEG), logits)
    if valid_classes is not None:
        logits = logits[valid_classes]
    return torch.log_softmax(logits, dim=0)


def transition_log_probs(transition_logits, transition_constraints=None, valid_classes=None,
                         allow_self_transitions=True):
    """semimarkov_modules.py:298-322 -- [to, from]; every column normalised."""
    t = transition_logits
    if transition_constraints is not None:
        t = torch.where(transition_constraints, torch.full_like(t, BIG_NEG), t)
    if valid_classes is not None:
        t = t[valid_classes][:, valid_classes]
    if not allow_self_transitions:
        eye = torch.eye(t.shape[0], dtype=torch.bool)
        t = torch.where(eye, torch.full_like(t, BIG_NEG), t)
    return torch.log_softmax(t, dim=0)


def length_log_probs(log_rates, max_k):
    """Poisson(exp(log_rate)).log_prob(k), k = 0..max_k-1  (semimarkov_modules.py:383-398).

    Row index == segment length.  max_k == 1 is the reference's HMM special case: a
    2-row table [0, -1000].
    """
    c = log_rates.shape[-1]
    if max_k == 1:
        return torch.tensor([0.0, -1000.0], dtype=log_rates.dtype).unsqueeze(-1).expand(2, c).clone()
    k = torch.arange(max_k, dtype=log_rates.dtype).unsqueeze(-1).expand(max_k, c)
    rate = torch.exp(log_rates)
    return torch.xlogy(k, rate) - rate - torch.lgamma(k + 1)


# --------------------------------------------------------------------------- span scoring

def sliding_sum(x, k):
    """out[b,t,c] = sum_{j=t}^{t+k-1} x[b,j,c], terms past the end dropped (modules:26-39)."""
    assert k > 0
    out = x.clone()
    n = x.shape[1]
    for j in range(1, min(k, n)):
        out[:, : n - j] += x[:, j:]
    return out


def augment_with_eos(transition, emission, init, length_scores, lengths, allowed_ends_per_instance=None):
    """EOS augmentation of semimarkov_modules.py:455-494 (add_eos=True branch).

    Returns per-instance (trans b x C x C, init b x C, len b x K x C, em b x N x C, lengths+1).
    """
    b, n1, c1 = emission.shape
    k = length_scores.shape[0]
    c, n = c1 + 1, n1 + 1
    kw = dict(dtype=emission.dtype)
    trans = torch.full((b, c, c), BIG_NEG, **kw)
    trans[:, :c1, :c1] = transition
    if allowed_ends_per_instance is None:
        trans[:, c1, :] = 0
    else:
        for i, ends in enumerate(allowed_ends_per_instance):
            assert len(ends) > 0
            trans[i, c1, list(ends)] = 0
    init_a = torch.full((b, c), BIG_NEG, **kw)
    init_a[:, :c1] = init
    len_a = torch.full((b, k, c), BIG_NEG, **kw)
    len_a[:, :, :c1] = length_scores
    len_a[:, 1 if k > 1 else 0, c1] = 0
    em = torch.full((b, n, c), BIG_NEG, **kw)
    for i, t in enumerate(lengths.tolist()):
        em[i, :t, :c1] = emission[i, :t]
        em[i, t, c1] = 0
    return trans, init_a, len_a, em, lengths + 1


def log_hsmm(transition, emission, init, length_scores, lengths, add_eos=True,
             allowed_ends_per_instance=None, wrap_quirk=True):
    """Dense potentials scores[b, n, k, c_to, c_from] (semimarkov_modules.py:416-523).

    transition C x C [to, from], emission b x N x C, init C, length_scores K x C,
    lengths b (true frame counts).  ``wrap_quirk`` reproduces the reference's
    Python negative index at :521 for short videos of a padded batch (it lands on a
    cell no path to the last position reads).
    """
    b, n1, c1 = emission.shape
    k_all = length_scores.shape[0]
    if k_all > n1:  # :450-452
        length_scores = length_scores[:n1]
    k_all = length_scores.shape[0]
    if add_eos:
        trans, init_a, len_a, em, lens = augment_with_eos(
            transition, emission, init, length_scores, lengths, allowed_ends_per_instance)
    else:
        trans = transition.unsqueeze(0).expand(b, c1, c1)
        init_a = init.unsqueeze(0).expand(b, c1)
        len_a = length_scores.unsqueeze(0).expand(b, k_all, c1)
        em, lens = emission, lengths
    n, c = em.shape[1], em.shape[2]
    scores = torch.zeros(b, n - 1, k_all, c, c, dtype=emission.dtype)
    scores += trans.view(b, 1, 1, c, c)
    scores[:, 0] += init_a.view(b, 1, 1, c)
    scores += len_a.view(b, 1, k_all, 1, c)
    window = None
    for k in range(1, k_all):
        # running window sum: S_k = S_{k-1} + (x shifted by k-1); equals sliding_sum(em, k)
        if window is None:
            window = em.clone()
        elif k - 1 < n:
            window[:, : n - (k - 1)] += em[:, k - 1:]
        for i in range(b):
            li = int(lens[i])
            scores[i, : li - 1, k] += window[i, : li - 1].view(li - 1, 1, c)
            pos = li - 1 - k
            if pos >= 0 or wrap_quirk:
                scores[i, pos, k] += em[i, li - 1].view(c, 1)
    return scores


# --------------------------------------------------------------------------- pinned third-party DP

class MaxSemiring:
    """torch_struct MaxSemiring: plus = max (first maximal index wins on CPU), times = +."""
    @staticmethod
    def sum(x, dim=-1):
        return torch.max(x, dim=dim)[0]


class LogSemiring:
    """torch_struct LogSemiring: plus = logsumexp, times = +."""
    @staticmethod
    def sum(x, dim=-1):
        return torch.logsumexp(x, dim=dim)


def semimarkov_dp(edge, lengths, semiring):
    """Sequential scan of torch_struct ``SemiMarkov._dp`` at the pinned commit.

    edge b x (N-1) x K x C x C indexed [b, n, k, c_to, c_from]; lengths b in POSITIONS
    (max(lengths) == N).  beta[n][c] = plus over k=1..min(K-1,n) of alpha[n-k][k][c],
    alpha[n-1][k][c_to] = plus over c_from of (beta[n-1][c_from] times edge[n-1,k,c_to,c_from]).
    Returns (v b, beta list of N tensors b x C).
    """
    b, n_1, k_all, c, _ = edge.shape
    n_pos = n_1 + 1
    assert int(max(lengths)) == n_pos, "one instance must span the whole lattice"
    beta = [torch.zeros(b, c, dtype=edge.dtype)]
    alpha = []
    ks_all = torch.arange(1, k_all)
    for n in range(1, n_pos):
        alpha.append(semiring.sum(beta[n - 1].view(b, 1, 1, c) + edge[:, n - 1], dim=-1))  # b x K x C
        kmax = min(k_all - 1, n)
        stack = torch.stack([alpha[n - k][:, k] for k in range(1, kmax + 1)], dim=-1)  # b x C x kmax
        beta.append(semiring.sum(stack, dim=-1))
    final = torch.stack([beta[int(l) - 1][i] for i, l in enumerate(lengths)], dim=0)  # b x C
    return semiring.sum(final, dim=-1), beta


def marginals(edge, lengths, semiring):
    """torch_struct ``_Struct.marginals``: d v / d edge by autograd (one-hot for Max)."""
    with torch.enable_grad():
        edge = edge.detach().clone().requires_grad_(True)
        v, _ = semimarkov_dp(edge, lengths, semiring)
        (g,) = torch.autograd.grad(v.sum(), edge)
    return v.detach(), g


def viterbi_backpointers(edge, lengths):
    """Explicit-back-pointer Viterbi with torch_struct's tie order.

    Equivalent to ``marginals(edge, lengths, MaxSemiring)`` (smallest k among maximal
    spans, then smallest c_from, and smallest c at the last position) but without the
    autograd graph, so it also runs on lattices of a few hundred MB.
    Returns (v b, segments): segments[i] = list of (n, k, c_to, c_from) edges.
    """
    b, n_1, k_all, c, _ = edge.shape
    n_pos = n_1 + 1
    beta = torch.zeros(b, n_pos, c, dtype=edge.dtype)
    alpha = torch.full((b, n_1, k_all, c), float('-inf'), dtype=edge.dtype)
    bp_from = torch.zeros(b, n_1, k_all, c, dtype=torch.int64)
    bp_k = torch.zeros(b, n_pos, c, dtype=torch.int64)
    for n in range(1, n_pos):
        val, idx = torch.max(beta[:, n - 1].view(b, 1, 1, c) + edge[:, n - 1], dim=-1)
        alpha[:, n - 1] = val
        bp_from[:, n - 1] = idx
        kmax = min(k_all - 1, n)
        ks = torch.arange(1, kmax + 1)
        diag = alpha[:, n - ks, ks]  # b x kmax x C
        val, idx = torch.max(diag, dim=1)
        beta[:, n] = val
        bp_k[:, n] = idx + 1
    v = torch.zeros(b, dtype=edge.dtype)
    segments = []
    for i in range(b):
        n = int(lengths[i]) - 1
        val, cur = torch.max(beta[i, n], dim=0)
        v[i] = val
        cur = int(cur)
        segs = []
        while n > 0:
            k = int(bp_k[i, n, cur])
            frm = int(bp_from[i, n - k, k, cur])
            segs.append((n - k, k, cur, frm))
            n -= k
            cur = frm
        segments.append(segs[::-1])
    return v, segments


def parts_from_segments(segments, shape, dtype=torch.float32):
    parts = torch.zeros(shape, dtype=dtype)
    for i, segs in enumerate(segments):
        for (n, k, c_to, c_from) in segs:
            parts[i, n, k, c_to, c_from] = 1
    return parts


def from_parts(parts):
    """torch_struct ``SemiMarkov.from_parts``: one-hot edges -> span encoding b x N (-1 = continuation)."""
    b, n_1 = parts.shape[:2]
    seq = torch.full((b, n_1 + 1), -1, dtype=torch.int64)
    for (i, n, k, c_to, c_from) in parts.nonzero().tolist():
        if n == 0:
            seq[i, 0] = c_from
        seq[i, n + k] = c_to
    return seq


def to_parts(sequence, num_classes, max_k, lengths=None):
    """torch_struct ``SemiMarkov.to_parts``: span encoding -> 0/1 b x (N-1) x K x C x C."""
    b, n = sequence.shape
    parts = torch.zeros(b, n - 1, max_k, num_classes, num_classes, dtype=torch.int64)
    for i in range(b):
        last, c = None, None
        for pos in range(n):
            if sequence[i, pos] == -1:
                assert pos != 0
                continue
            new_c = int(sequence[i, pos])
            if pos != 0:
                parts[i, last, pos - last, new_c, c] = 1
            last, c = pos, new_c
    return parts


def spans_from_segments(segments, n_pos):
    seq = torch.full((len(segments), n_pos), -1, dtype=torch.int64)
    for i, segs in enumerate(segments):
        for (n, k, c_to, c_from) in segs:
            if n == 0:
                seq[i, 0] = c_from
            seq[i, n + k] = c_to
    return seq


def brute_force(edge, length, semiring_name):
    """Enumerate every segmentation of one instance (tiny lattices only).

    A path is c_0 (span [0,k_1)), c_1, ... ending with a span START at position
    length-1; score = sum of edge[n, k, c_to, c_from].  Returns max / logsumexp.
    """
    n_1, k_all, c, _ = edge.shape
    last = int(length) - 1
    scores = []

    def rec(n, cur, acc):
        if n == last:
            scores.append(acc)
            return
        for k in range(1, k_all):
            if n + k > last:
                break
            for nxt in range(c):
                rec(n + k, nxt, acc + float(edge[n, k, nxt, cur]))

    for c0 in range(c):
        rec(0, c0, 0.0)
    s = torch.tensor(scores, dtype=torch.float64)
    return float(s.max()) if semiring_name == 'max' else float(torch.logsumexp(s, 0))


# --------------------------------------------------------------------------- span / label codecs

def labels_to_spans(labels, max_k):
    """semimarkov_utils.py:6-23.  labels b x N -> spans (-1 = continuation); a run is cut
    every max_k-1 frames."""
    labels = np.asarray(labels)
    b, n = labels.shape
    assert not (labels == -1).any()
    out = labels.copy()
    for i in range(b):
        run = 1
        for t in range(1, n):
            same = labels[i, t] == labels[i, t - 1]
            if max_k is not None:
                same = same and run < max_k - 1
            if same:
                out[i, t] = -1
                run += 1
            else:
                run = 1
    return out


def spans_to_labels(spans):
    """semimarkov_utils.py:51-63: forward-fill -1 with the running label."""
    spans = np.asarray(spans)
    out = spans.copy()
    assert (out[:, 0] != -1).all()
    for t in range(1, out.shape[1]):
        cont = out[:, t] == -1
        out[cont, t] = out[cont, t - 1]
    return out


def rle_spans(spans, lengths):
    """semimarkov_utils.py:26-48: [(symbol, run length), ...] per instance."""
    res = []
    for i in range(len(spans)):
        rle = []
        for sym in np.asarray(spans[i])[: int(lengths[i])].tolist():
            if not rle or sym != -1:
                rle.append([sym, 0])
            rle[-1][1] += 1
        res.append([(s, c) for s, c in rle])
    return res


# --------------------------------------------------------------------------- closed-form supervised fit

def sufficient_stats(feature_list, label_list, n_classes, max_k):
    """semimarkov_utils.py:74-126 restated with numpy.

    Per-class means use sklearn's one-hot-responsibility estimate (nk = count + 10*eps);
    the covariance is the tied, diagonal GLOBAL biased variance + 1e-6 (reg_covar).
    """
    span_counts = np.zeros(n_classes, np.float32)
    span_lengths = np.zeros(n_classes, np.float32)
    start_counts = np.zeros(n_classes, np.float32)
    trans_counts = np.zeros((n_classes, n_classes), np.float32)
    for labels in label_list:
        labels = np.asarray(labels)
        spans = labels_to_spans(labels[None], max_k)
        prev = None
        for j, (sym, length) in enumerate(rle_spans(spans, [spans.shape[1]])[0]):
            if j == 0:
                start_counts[sym] += 1
            span_counts[sym] += 1
            span_lengths[sym] += length
            if prev is not None:
                trans_counts[sym, prev] += 1
            prev = sym
    x = np.vstack([np.asarray(f, np.float64) for f in feature_list])
    y = np.concatenate([np.asarray(l) for l in label_list])
    resp = np.zeros((x.shape[0], n_classes))
    resp[np.arange(x.shape[0]), y] = 1
    nk = resp.sum(0) + 10 * np.finfo(resp.dtype).eps
    means = resp.T @ x / nk[:, None]
    n_all = x.shape[0] + 10 * np.finfo(np.float64).eps
    gmean = x.sum(0) / n_all
    var = (x * x).sum(0) / n_all - gmean ** 2 + 1e-6
    return dict(means=means, var=var, span_counts=span_counts, span_lengths=span_lengths,
                span_start_counts=start_counts, span_transition_counts=trans_counts,
                instance_count=len(feature_list))


def fit_supervised(feature_list, label_list, n_classes, max_k, state_smoothing=1e-2, length_smoothing=1e-1,
                   merge_classes=None):
    """semimarkov_modules.py:195-256 -> dict of float32 parameter arrays."""
    st = sufficient_stats(feature_list, label_list, n_classes, max_k)
    stm = st
    if merge_classes is not None:
        merged = [np.array([merge_classes[int(v)] for v in np.asarray(l)]) for l in label_list]
        stm = sufficient_stats(feature_list, merged, n_classes, max_k)
    with np.errstate(divide='ignore', invalid='ignore'):
        init_p = (st['span_start_counts'] + state_smoothing) / float(st['instance_count'] + state_smoothing * n_classes)
        init_p[np.isnan(init_p)] = 0
        tc = st['span_transition_counts'] + state_smoothing
        trans_p = tc / tc.sum(axis=0)[None, :]
        trans_p[np.isnan(trans_p)] = 0
        mean_len = (stm['span_lengths'] + length_smoothing) / (stm['span_counts'] + length_smoothing)
        return dict(
            init_logits=np.log(init_p).astype(np.float32),
            transition_logits=np.log(trans_p).astype(np.float32),
            poisson_log_rates=np.log(mean_len).astype(np.float32),
            gaussian_means=stm['means'].astype(np.float32),
            gaussian_cov=np.diag(stm['var']).astype(np.float32),
        )


# --------------------------------------------------------------------------- module-level restatement

class RefParams:
    """Plain container mirroring the reference module's state (modules:142-193)."""

    def __init__(self, n_classes, poisson_log_rates, gaussian_means, gaussian_cov_diag, transition_logits, init_logits,
                 max_k, allow_self_transitions=True, init_constraints=None, transition_constraints=None,
                 allowed_ends=None, merge_classes=None):
        self.n_classes = n_classes
        self.poisson_log_rates = poisson_log_rates
        self.gaussian_means = gaussian_means
        self.gaussian_cov_diag = gaussian_cov_diag
        self.transition_logits = transition_logits
        self.init_logits = init_logits
        self.max_k = max_k
        self.allow_self_transitions = allow_self_transitions
        self.init_constraints = init_constraints
        self.transition_constraints = transition_constraints
        self.allowed_ends = allowed_ends
        self.merge_classes = merge_classes

    def to(self, dtype):
        cp = RefParams.__new__(RefParams)
        cp.__dict__.update(self.__dict__)
        for name in ('poisson_log_rates', 'gaussian_means', 'gaussian_cov_diag', 'transition_logits', 'init_logits'):
            setattr(cp, name, getattr(self, name).to(dtype))
        return cp


def factor_tables(p, valid_classes):
    """(elp-independent) log-prob tables for one batch: trans C x C, init C, len K x C.  modules:579-590."""
    vc = valid_classes
    merged = vc if vc is not None else torch.arange(p.n_classes)
    if p.merge_classes is not None:
        merged = torch.tensor([p.merge_classes[int(i)] for i in merged])
    trans = transition_log_probs(p.transition_logits, p.transition_constraints, vc, p.allow_self_transitions)
    init = initial_log_probs(p.init_logits, p.init_constraints, vc)
    lens = length_log_probs(p.poisson_log_rates[merged], p.max_k)
    return trans, init, lens, merged


def allowed_ends_for_batch(p, valid_classes, additional_per_instance, b):
    """modules:566-577: local positions of allowed_ends | additional, per instance (None = unrestricted)."""
    if p.allowed_ends is None:
        return None
    vc = list(range(p.n_classes)) if valid_classes is None else [int(v) for v in valid_classes]
    if additional_per_instance is None:
        additional_per_instance = [set() for _ in range(b)]
    res = [[i for i, ix in enumerate(vc) if ix in (set(p.allowed_ends) | set(add))] for add in additional_per_instance]
    assert all(res), res
    return res


def score_features(p, features, lengths, valid_classes, add_eos=True, additional_allowed_ends_per_instance=None,
                   constraints=None):
    """modules:553-595 -> (scores, elp)."""
    trans, init, lens, merged = factor_tables(p, valid_classes)
    elp = emission_log_probs(features, p.gaussian_means[merged], p.gaussian_cov_diag, constraints)
    ends = allowed_ends_for_batch(p, valid_classes, additional_allowed_ends_per_instance, features.shape[0])
    scores = log_hsmm(trans, elp, init, lens, lengths, add_eos=add_eos, allowed_ends_per_instance=ends)
    return scores, elp


def unmap_spans(spans, valid_classes, n_classes):
    """modules:683-691: local class position -> global id; EOS (local C) -> n_classes; -1 stays."""
    if valid_classes is None:
        return spans
    vc = [int(v) for v in valid_classes]
    table = {i: c for i, c in enumerate(vc)}
    table[-1] = -1
    table[len(vc)] = n_classes
    return torch.tensor([[table[int(v)] for v in row] for row in spans.tolist()], dtype=torch.int64)


def viterbi(p, features, lengths, valid_classes, add_eos=True, additional_allowed_ends_per_instance=None,
            constraints=None):
    """modules:660-696 -> (pred_spans b x (Tmax+1) int64 global ids, best score b, elp)."""
    r = viterbi_full(p, features, lengths, valid_classes, add_eos, additional_allowed_ends_per_instance, constraints)
    return r['spans'], r['v'], r['elp']


def viterbi_full(p, features, lengths, valid_classes, add_eos=True, additional_allowed_ends_per_instance=None,
                 constraints=None):
    scores, elp = score_features(p, features, lengths, valid_classes, add_eos,
                                 additional_allowed_ends_per_instance, constraints)
    pos_lengths = lengths + 1 if add_eos else lengths
    v, segs = viterbi_backpointers(scores, pos_lengths)
    local = spans_from_segments(segs, scores.shape[1] + 1)
    return dict(spans=unmap_spans(local, valid_classes, p.n_classes), local_spans=local, v=v, elp=elp,
                scores=scores, segments=segs, pos_lengths=pos_lengths)


def map_spans_to_local(spans, valid_classes, n_classes):
    """Inverse of unmap_spans (modules:626-639)."""
    if valid_classes is None:
        return spans.clone()
    table = {int(c): i for i, c in enumerate(valid_classes)}
    table[-1] = -1
    table[n_classes] = len(valid_classes)
    return torch.tensor([[table[int(v)] for v in row] for row in spans.tolist()], dtype=torch.int64)


def rescore(scores, local_spans, pos_lengths):
    """Score of a span encoding under dense potentials: sum(scores * to_parts) (torch_struct ``score``)."""
    k, c = scores.shape[2], scores.shape[3]
    out = []
    for i in range(scores.shape[0]):
        li = int(pos_lengths[i])
        parts = to_parts(local_spans[i:i + 1, :li], c, k)
        out.append((scores[i:i + 1, :li - 1] * parts.to(scores.dtype)).sum())
    return torch.stack(out)


def log_partition(p, features, lengths, valid_classes, add_eos=True, additional_allowed_ends_per_instance=None,
                  constraints=None):
    """modules:597-658 with spans=None -> logZ per instance (the reference returns its mean)."""
    scores, _ = score_features(p, features, lengths, valid_classes, add_eos,
                               additional_allowed_ends_per_instance, constraints)
    pos_lengths = lengths + 1 if add_eos else lengths
    v, _ = semimarkov_dp(scores, pos_lengths, LogSemiring)
    return v
