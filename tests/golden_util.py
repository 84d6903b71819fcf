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
