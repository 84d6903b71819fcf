"""Helpers shared by the tests: rebuild oracle parameter sets from the golden fixture."""
import numpy as np
import torch

from oracle import dense_ref as O

CASES = {
    'tiny': dict(K=4),
    'subset_merge': dict(K=8, merge={0: 0, 1: 1, 2: 2, 3: 3, 4: 1, 5: 5, 6: 6, 7: 1, 8: 8}),
    'k_gt_t': dict(K=8),
    'hmm_k1': dict(K=1),
    'constrained': dict(K=6, ends={4, 6}, additional=[[], [2], [3]]),
    'no_eos': dict(K=4, add_eos=False),
}


def case_inputs(g, case, dtype=torch.float64):
    pre = case + '/'
    cfg = CASES[case]
    t = lambda k: torch.from_numpy(g[pre + k])
    prm = lambda k: torch.from_numpy(g[pre + 'param/' + k])
    n_classes = g[pre + 'param/init_logits'].shape[0]
    p = O.RefParams(
        n_classes, prm('poisson_log_rates'), prm('gaussian_means'), torch.diagonal(prm('gaussian_cov')).clone(),
        prm('transition_logits'), prm('init_logits'), cfg['K'], True,
        prm('init_constraints') if pre + 'param/init_constraints' in g else None,
        prm('transition_constraints') if pre + 'param/transition_constraints' in g else None,
        cfg.get('ends'), cfg.get('merge')).to(dtype)
    feats = t('features').to(dtype)
    lengths = t('lengths')
    valid = t('valid_classes') if pre + 'valid_classes' in g else None
    cons = t('constraints').to(dtype) if pre + 'constraints' in g else None
    return p, feats, lengths, valid, cons, cfg


def assert_spans_equivalent(spans, ref_spans, lengths, eos_id, scores=None, v=None, pos_lengths=None):
    """Span encodings must give identical frame labels and EOS placement.  Boundaries may differ only
    between consecutive spans of ONE class -- (a,b) and (b,a) splits of a run score exactly the same in
    real arithmetic, so which one wins is rounding noise -- and then the path must re-score to the optimum."""
    spans, ref_spans = np.asarray(spans), np.asarray(ref_spans)
    for i, t in enumerate(np.asarray(lengths).tolist()):
        assert spans[i, t] == ref_spans[i, t] == eos_id, (i, spans[i, t], ref_spans[i, t])
        assert (spans[i, t + 1:] == -1).all()
        np.testing.assert_array_equal(O.spans_to_labels(spans[i:i + 1, :t]), O.spans_to_labels(ref_spans[i:i + 1, :t]))
    if scores is not None and not np.array_equal(spans, ref_spans):
        np.testing.assert_allclose(O.rescore(scores, torch.from_numpy(spans), pos_lengths).numpy(),
                                   np.asarray(v), rtol=1e-9, atol=1e-7)


def span_start_differences(spans, ref_spans, lengths):
    """Positions at which two span encodings disagree, after asserting that every one of them lies INSIDE a run of one
    class in both decodings (frame labels equal everywhere; a boundary that only one of the two draws splits a run into
    two consecutive spans of the same class).  SURVEY 8c(1): this is the only way the reference's fp32 run differs from
    its own fp64 run on the golden cases."""
    spans, ref_spans = np.asarray(spans), np.asarray(ref_spans)
    out = []
    for i, t in enumerate(np.asarray(lengths).tolist()):
        la = O.spans_to_labels(spans[i:i + 1, :t])[0]
        lb = O.spans_to_labels(ref_spans[i:i + 1, :t])[0]
        np.testing.assert_array_equal(la, lb)
        for n in np.flatnonzero(spans[i, :t] != ref_spans[i, :t]).tolist():
            assert n > 0 and la[n] == la[n - 1], (i, n)          # one decoding starts a new span of the SAME class here
            assert {int(spans[i, n]), int(ref_spans[i, n])} == {-1, int(la[n])}, (i, n)
            out.append((i, n))
    return out


def crosstask_magnitude_case(seed, t=800, c=8, k=24, d=200, mean_scale=0.05):
    """SURVEY App. C.3's shape: per-frame emission log-probs of about -280 at D = 200, a best score of about -2.2e5 (one
    fp32 ulp there is 2^-6), class means close enough (0.05 sigma per dimension) that the fp32 reference path resolves
    some boundaries by rounding.  -> (RefParams fp32, features fp32 [1, t, d], lengths)."""
    g = torch.Generator().manual_seed(100 + seed)
    rng = np.random.default_rng(100 + seed)
    mu = torch.randn(c, d, generator=g) * mean_scale
    sigma = 0.7 + 0.6 * torch.rand(d, generator=g)
    rates = rng.uniform(4, 20, size=c)
    lab, cur, total = [], int(rng.integers(0, c)), 0
    while total < t:
        ln = int(np.clip(rng.poisson(rates[cur]), 1, k - 1))
        lab.append(np.full(ln, cur))
        total += ln
        nxt = int(rng.integers(0, c - 1))
        cur = nxt + (nxt >= cur)
    lab = torch.from_numpy(np.concatenate(lab)[:t])
    x = (mu[lab] + sigma * torch.randn(t, d, generator=g)).float()
    p32 = O.RefParams(c, torch.log(torch.tensor(rates)).float(), mu.float(), (sigma ** 2).float(),
                      torch.randn(c, c, generator=g).float(), torch.rand(c, generator=g).float(), k, True)
    return p32, x[None], torch.tensor([t])


def fp32_near_tie_certificate(p32, feats, lengths, spans):
    """SURVEY 8c(1), the fp32 clause: against the reference path run in fp32 a decoding is either identical, or --
    re-scored under the dense fp64 potentials -- within 4 fp32 ulps of the fp32 run's optimum.  Also returned: how much
    BETTER than the fp32 run's own path it scores under the exact potentials (>= 0: the difference is the fp32 run's
    rounding, not this decoding's).  -> dict(frames_differing, ulps_from_fp32_optimum, gain_over_fp32_path)."""
    t = int(lengths[0])
    r32 = O.viterbi_full(p32, feats.float(), lengths, None)
    r64 = O.viterbi_full(p32.to(torch.float64), feats.double(), lengths, None)
    spans = torch.as_tensor(np.asarray(spans))
    l32 = O.spans_to_labels(r32['spans'][:, :t].numpy())[0]
    mine = O.spans_to_labels(spans[:, :t].numpy())[0]
    s_mine = O.rescore(r64['scores'], spans, r64['pos_lengths']).item()
    s_32 = O.rescore(r64['scores'], r32['local_spans'], r64['pos_lengths']).item()
    v32 = float(r32['v'][0])
    ulp = float(np.spacing(np.float32(abs(v32))))
    return dict(frames_differing=int((mine != l32).sum()), ulps_from_fp32_optimum=(s_mine - v32) / ulp,
                gain_over_fp32_path=s_mine - s_32, labels_equal_fp64_run=bool(
                    (mine == O.spans_to_labels(r64['spans'][:, :t].numpy())[0]).all()))
