"""GPU parity of the product SemiMarkovModule.viterbi (features -> spans) against the oracles."""
import numpy as np
import pytest
import torch

from oracle import dense_ref as O
from oracle import factored as F
from golden_util import (CASES, case_inputs, assert_spans_equivalent, span_start_differences, crosstask_magnitude_case,
                         fp32_near_tie_certificate)
from module_util import module_from_golden, make_args

pytestmark = pytest.mark.gpu
EOS_CASES = [c for c in CASES if CASES[c].get('add_eos', True)]


@pytest.mark.parametrize('case', EOS_CASES)
def test_viterbi_matches_reference_path_on_golden_cases(golden, case):
    """End to end against the fp64 run of the reference path (dense potentials + restated pytorch-struct DP)."""
    dev = torch.device('cuda:0')
    m = module_from_golden(golden, case).to(dev)
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    vc = None if valid is None else [valid for _ in range(feats.shape[0])]
    spans, elp = m.viterbi(feats.float().to(dev), lengths.to(dev), vc, add_eos=True,
                           additional_allowed_ends_per_instance=cfg.get('additional'),
                           constraints=None if cons is None else cons.float().to(dev), return_elp=True)
    assert spans.device.type == 'cpu' and spans.dtype == torch.int64
    assert spans.shape == (feats.shape[0], feats.shape[1] + 1)
    r = O.viterbi_full(p, feats, lengths, valid, True, cfg.get('additional'), cons)
    for i, t in enumerate(lengths.tolist()):
        np.testing.assert_allclose(elp[i, :t].cpu().numpy(), r['elp'][i, :t].numpy(), rtol=3e-7, atol=1e-5)
    # same frame labels / EOS placement as the reference path; boundaries inside one-class runs certified by re-scoring
    assert_spans_equivalent(spans.numpy(), r['spans'].numpy(), lengths, p.n_classes)
    local = O.map_spans_to_local(spans, valid, p.n_classes)
    np.testing.assert_allclose(O.rescore(r['scores'], local, r['pos_lengths']).numpy(), r['v'].numpy(),
                               rtol=1e-9, atol=1e-7)
    # and identical to what the reference's own host code produced with the restated DP (fixture)
    assert_spans_equivalent(spans.numpy(), golden[case + '/f64/ref_spans'], lengths, p.n_classes)


@pytest.mark.parametrize('case', EOS_CASES)
def test_viterbi_against_the_reference_fp32_run(golden, case):
    """SURVEY 8c(1), fp32 clause, on the fixtures: the reference's OWN host code run in fp32 (``*/f32/ref_spans``; the
    reference computes in fp32) against ``viterbi()``: identical frame labels and EOS placement, and every span start
    that differs sits inside a run of one class -- which is also the only way the reference's fp32 run differs from its
    own fp64 run (tests/test_oracle_golden.py pins that: 2 positions each on ``k_gt_t`` and ``constrained``)."""
    dev = torch.device('cuda:0')
    m = module_from_golden(golden, case).to(dev)
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    vc = None if valid is None else [valid for _ in range(feats.shape[0])]
    spans = m.viterbi(feats.float().to(dev), lengths.to(dev), vc, add_eos=True,
                      additional_allowed_ends_per_instance=cfg.get('additional'),
                      constraints=None if cons is None else cons.float().to(dev))
    ref32 = golden[case + '/f32/ref_spans']
    assert_spans_equivalent(spans.numpy(), ref32, lengths, p.n_classes)
    diffs = span_start_differences(spans.numpy(), ref32, lengths)       # asserts "inside one-class runs" position by position
    assert len(diffs) <= 2, diffs
    # ... and against the fp64 run of the same host code there is nothing to certify on these cases
    assert len(span_start_differences(spans.numpy(), golden[case + '/f64/ref_spans'], lengths)) == 0


def test_fp32_near_tie_certificate_at_crosstask_magnitudes():
    """T = 800, 8 states, K = 24, D = 200, per-frame emissions ~ -280, best score ~ -2.2e5 (one
    fp32 ulp = 2^-6), twelve seeds.  ``viterbi()`` equals the reference path run in fp64 frame for frame; against the
    reference path run in fp32 (the precision the reference really computes in) it differs on one of the twelve videos,
    by 4 frames, and there carries the certificate: re-scored under the dense fp64 potentials it is within 4 fp32 ulps
    of the fp32 run's optimum AND scores higher than the fp32 run's own path -- the difference is the fp32 run's
    rounding."""
    from action_segmentation_amd.semimarkov_modules import SemiMarkovModule
    dev = torch.device('cuda:0')
    differing = []
    for seed in range(12):
        p32, feats, lengths = crosstask_magnitude_case(seed)
        c, d = p32.n_classes, feats.shape[2]
        m = SemiMarkovModule(make_args(p32.max_k), c, d, allow_self_transitions=True)
        with torch.no_grad():
            m.poisson_log_rates.copy_(p32.poisson_log_rates)
            m.gaussian_means.copy_(p32.gaussian_means)
            m.gaussian_cov.copy_(torch.diag(p32.gaussian_cov_diag))
            m.transition_logits.copy_(p32.transition_logits)
            m.init_logits.copy_(p32.init_logits)
        m = m.to(dev)
        spans = m.viterbi(feats.to(dev), lengths.to(dev), None, add_eos=True)
        local = O.map_spans_to_local(spans, None, c)
        cert = fp32_near_tie_certificate(p32, feats, lengths, local.numpy())
        assert cert['labels_equal_fp64_run'], (seed, cert)
        assert abs(cert['ulps_from_fp32_optimum']) <= 4.0, (seed, cert)
        assert cert['gain_over_fp32_path'] >= 0.0, (seed, cert)
        if cert['frames_differing']:
            differing.append((seed, cert['frames_differing']))
            assert cert['gain_over_fp32_path'] > 0.0, (seed, cert)
    assert differing == [(6, 4)], differing


@pytest.mark.parametrize('case', EOS_CASES)
def test_viterbi_bit_exact_vs_factored_oracle_on_module_tables(golden, case):
    dev = torch.device('cuda:0')
    m = module_from_golden(golden, case).to(dev)
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    b = feats.shape[0]
    vc = None if valid is None else [valid for _ in range(b)]
    spans = m.viterbi(feats.float().to(dev), lengths.to(dev), vc,
                      additional_allowed_ends_per_instance=cfg.get('additional'),
                      constraints=None if cons is None else cons.float().to(dev))
    with torch.no_grad():
        tab = {k: v.cpu().numpy() for k, v in m.factor_tables(valid).items()}
    c = tab['init'].shape[0]
    w = tab['w']                                   # D x C
    x = feats.float().numpy().astype(np.float64)
    elp = tab['cst'] + x @ w - 0.5 * (x * x) @ tab['inv_var'][:, None]
    if cons is not None:
        elp = elp + cons.float().numpy().astype(np.float64)
    ends = O.allowed_ends_for_batch(p, valid, cfg.get('additional'), b)
    fs, fv = F.viterbi(elp, lengths.numpy(), tab['trans'], tab['init'], tab['len'],
                       F.endpen_from_allowed_ends(ends, b, c))
    table = np.concatenate([tab['class_map'], [-1]])
    # emission sums differ in association (GPU: running FMA; here: matmul), so equality is up to exact ties only
    assert_spans_equivalent(spans.numpy(), table[fs], lengths, p.n_classes)


def test_viterbi_decode_alias_and_larger_random_batch():
    from action_segmentation_amd.semimarkov_modules import SemiMarkovModule
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    n_classes, d, k, b, tmax = 12, 40, 50, 6, 700
    m = SemiMarkovModule(make_args(k), n_classes, d, allow_self_transitions=True)
    with torch.no_grad():
        m.poisson_log_rates.copy_(torch.rand(n_classes, generator=g) * 2 + 1.5)
        m.gaussian_means.copy_(torch.randn(n_classes, d, generator=g) * 0.5)
        m.gaussian_cov.copy_(torch.diag(0.5 + torch.rand(d, generator=g)))
        m.transition_logits.copy_(torch.randn(n_classes, n_classes, generator=g))
    m = m.to(dev)
    lengths = torch.randint(300, tmax + 1, (b,), generator=g); lengths[2] = tmax
    labels = torch.randint(0, n_classes, (b, tmax // 25 + 1), generator=g).repeat_interleave(25, dim=1)[:, :tmax]
    feats = m.gaussian_means.detach().cpu()[labels] + torch.randn(b, tmax, d, generator=g)
    for i, t in enumerate(lengths.tolist()):
        feats[i, t:] = 0
    valid = torch.tensor([0, 2, 3, 5, 6, 7, 9, 11])
    spans = m.viterbi_decode(feats.to(dev), lengths.to(dev), [valid] * b)
    p = O.RefParams(n_classes, m.poisson_log_rates.detach().cpu(), m.gaussian_means.detach().cpu(),
                    torch.diagonal(m.gaussian_cov.detach().cpu()).clone(), m.transition_logits.detach().cpu(),
                    m.init_logits.detach().cpu(), k, True).to(torch.float64)
    trans, init, lens, merged = O.factor_tables(p, valid)
    elp = O.emission_log_probs(feats.double(), p.gaussian_means[merged], p.gaussian_cov_diag)
    fs, fv = F.viterbi(elp.numpy(), lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy())
    table = np.array(valid.tolist() + [n_classes, -1])
    assert_spans_equivalent(spans.numpy(), table[fs], lengths, n_classes)


@pytest.mark.parametrize('case', EOS_CASES)
def test_log_likelihood_matches_reference_path(golden, case):
    """log Z (HIP forward) and the gold-span joint score against the dense reference path (fp64)."""
    dev = torch.device('cuda:0')
    m = module_from_golden(golden, case).to(dev)
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    b = feats.shape[0]
    vc = None if valid is None else [valid for _ in range(b)]
    kw = dict(additional_allowed_ends_per_instance=cfg.get('additional'),
              constraints=None if cons is None else cons.float().to(dev))
    ll, log_det = m.log_likelihood(feats.float().to(dev), lengths.to(dev), vc, spans=None, **kw)
    ref = float(golden[case + '/f64/ref_mean_logz'])
    assert abs(ll.item() - ref) <= 1e-6 * abs(ref) + 1e-4 and log_det.item() == 0.0
    # gold-span score: score the Viterbi path -> must equal the Viterbi value, and be <= log Z
    r = O.viterbi_full(p, feats, lengths, valid, True, cfg.get('additional'), cons)
    gold = r['spans'][:, :feats.shape[1]].clone()
    js, _ = m.log_likelihood(feats.float().to(dev), lengths.to(dev), vc, spans=gold.to(dev), **kw)
    np.testing.assert_allclose(js.item(), r['v'].mean().item(), rtol=1e-9, atol=1e-6)
    assert js.item() <= ll.item() + 1e-6


@pytest.mark.parametrize('case', ['tiny', 'subset_merge', 'constrained'])
def test_log_likelihood_gradients_match_dense_autograd(golden, case):
    """d mean(logZ) / d parameters through the HIP forward/backward kernels vs autograd through the dense reference
    path (fp64 log_hsmm potentials + LogSemiring DP)."""
    dev = torch.device('cuda:0')
    m = module_from_golden(golden, case).to(dev)
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    b = feats.shape[0]
    vc = None if valid is None else [valid for _ in range(b)]
    m.zero_grad()
    ll, _ = m.log_likelihood(feats.float().to(dev), lengths.to(dev), vc, spans=None,
                             additional_allowed_ends_per_instance=cfg.get('additional'),
                             constraints=None if cons is None else cons.float().to(dev))
    ll.backward()
    # dense reference in fp64 with autograd on CPU
    names = ['poisson_log_rates', 'gaussian_means', 'transition_logits', 'init_logits']
    q = p.to(torch.float64)
    leaves = {n: getattr(q, n).clone().requires_grad_(True) for n in names}
    for n, v in leaves.items():
        setattr(q, n, v)
    scores, _ = O.score_features(q, feats, lengths, valid, True, cfg.get('additional'), cons)
    z, _ = O.semimarkov_dp(scores, lengths + 1, O.LogSemiring)
    z.mean().backward()
    assert abs(ll.item() - z.mean().item()) <= 1e-6 * abs(z.mean().item()) + 1e-4
    for n in names:
        got = getattr(m, n).grad.detach().cpu().double().numpy()
        ref = leaves[n].grad.numpy()
        np.testing.assert_allclose(got, ref, rtol=5e-4, atol=5e-4 * max(1.0, np.abs(ref).max()), err_msg=n)


def test_no_eos_viterbi_and_log_likelihood_match_reference_path(golden):
    """add_eos=False through the module API (reference :597-696 with add_eos=False) on the reference-generated 'no_eos'
    golden case: Viterbi spans / labels and log Z + parameter gradients against the dense reference path in fp64."""
    case = 'no_eos'
    dev = torch.device('cuda:0')
    m = module_from_golden(golden, case).to(dev)
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    b, tmax = feats.shape[:2]
    vc = None if valid is None else [valid for _ in range(b)]
    spans = m.viterbi(feats.float().to(dev), lengths.to(dev), vc, add_eos=False)
    assert tuple(spans.shape) == (b, tmax) and (spans != m.n_classes).all()
    r = O.viterbi_full(p, feats, lengths, valid, False, None, cons)
    ref = r['spans'].numpy()[:, :tmax]
    for i, t in enumerate(lengths.tolist()):
        np.testing.assert_array_equal(O.spans_to_labels(spans.numpy()[i:i + 1, :t]), O.spans_to_labels(ref[i:i + 1, :t]))
    # log Z and its gradient
    m.zero_grad()
    ll, _ = m.log_likelihood(feats.float().to(dev), lengths.to(dev), vc, spans=None, add_eos=False)
    ll.backward()
    names = ['poisson_log_rates', 'gaussian_means', 'transition_logits', 'init_logits']
    q = p.to(torch.float64)
    leaves = {n: getattr(q, n).clone().requires_grad_(True) for n in names}
    for n, v in leaves.items():
        setattr(q, n, v)
    scores, _ = O.score_features(q, feats, lengths, valid, False, None, cons)
    z, _ = O.semimarkov_dp(scores, lengths, O.LogSemiring)
    z.mean().backward()
    assert abs(ll.item() - z.mean().item()) <= 1e-6 * abs(z.mean().item()) + 1e-4
    for n in names:
        got = getattr(m, n).grad.detach().cpu().double().numpy()
        refg = leaves[n].grad.numpy()
        np.testing.assert_allclose(got, refg, rtol=5e-4, atol=5e-4 * max(1.0, np.abs(refg).max()), err_msg=n)
    # gold-span score of the Viterbi path (to_parts has no edge for the last span: reference :641-655)
    js, _ = m.log_likelihood(feats.float().to(dev), lengths.to(dev), vc, spans=r['spans'][:, :tmax].to(dev), add_eos=False)
    parts = O.to_parts(r['spans'][:, :tmax], scores.shape[-1], scores.shape[2], lengths)
    np.testing.assert_allclose(js.item(), (scores * parts).sum(dim=(1, 2, 3, 4)).mean().item(), rtol=1e-9, atol=1e-6)


def test_span_codecs_run_on_the_device():
    """labels_to_spans / spans_to_labels given device tensors stay on the device (reference semimarkov_utils.py:6-63;
    the gradient-based supervised fit encodes every batch, semimarkov.py:251) and equal the host statement, including
    the reference's own vectors (src/models/test_semimarkov.py:251-259)."""
    from action_segmentation_amd import semimarkov_utils as U
    dev = torch.device('cuda:0')
    labels = torch.tensor([[0, 1, 1, 2, 2, 2], [0, 1, 2, 3, 3, 4]])
    spans = torch.tensor([[0, 1, -1, 2, -1, -1], [0, 1, 2, 3, -1, 4]])
    got = U.labels_to_spans(labels.to(dev), max_k=None)
    assert got.is_cuda and torch.equal(got.cpu(), spans)
    back = U.spans_to_labels(spans.to(dev))
    assert back.is_cuda and torch.equal(back.cpu(), labels)
    g = torch.Generator().manual_seed(3)
    for max_k in (2, 5, 17, None):
        runs = torch.randint(0, 6, (4, 40), generator=g)
        lab = torch.repeat_interleave(runs, 7, dim=1)[:, :257]
        host = U.labels_to_spans(lab, max_k)
        devv = U.labels_to_spans(lab.to(dev), max_k)
        assert torch.equal(devv.cpu(), host)
        assert torch.equal(U.spans_to_labels(devv).cpu(), lab)


@pytest.mark.parametrize('variant', ['plain', 'constrained', 'merged', 'self', 'big'])
def test_factor_tables_kernel_matches_torch_tables(variant):
    """smm_factor_tables_f64 / _bwd_f64 (one launch each way) against the batched differentiable torch statement of
    reference :284-414 on the same parameters: table values and parameter gradients, several class sets of different
    sizes sharing classes, with transition constraints / merge_classes / self transitions."""
    import types
    from action_segmentation_amd.semimarkov_modules import SemiMarkovModule
    torch.manual_seed(5)
    n_classes, d, k = (14, 37, 70) if variant != 'big' else (45, 300, 1024)
    kw = {}
    if variant == 'constrained':
        kw = dict(allowed_starts={0, 3, 7}, allowed_ends={2, 5},
                  allowed_transitions={i: {(i + 1) % n_classes, (i + 3) % n_classes, (i + 6) % n_classes}
                                       for i in range(n_classes)})
    if variant == 'merged':
        kw = dict(merge_classes={i: (0 if i in (0, 4, 9) else i) for i in range(n_classes)})
    m = SemiMarkovModule(make_args(k), n_classes, d, allow_self_transitions=(variant == 'self'), **kw).cuda()
    with torch.no_grad():
        m.init_logits.normal_(); m.transition_logits.normal_(); m.poisson_log_rates.uniform_(0.5, 3.5)
        m.gaussian_means.normal_()
        m.gaussian_cov.copy_(torch.diag(0.5 + torch.rand(d)))
    dev = torch.device('cuda:0')
    sets = ([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], [0, 5, 4], [13, 2, 9, 0, 7], None)
    if variant == 'big':            # 32 states (the kernels' maximum), the full length table, D > 256
        sets = (list(range(3, 35)), [44, 0, 17], list(range(20, 43)))
    groups = [dict(valid_classes=None if v is None else torch.tensor(v)) for v in sets]
    names = ('trans', 'init', 'len', 'w', 'cst', 'inv_var')
    grads, tabs = {}, {}
    for use_hip in (False, True):
        pc = types.SimpleNamespace(groups=groups)
        m.zero_grad()
        st, n_states, cm, k_rows = m._stacked_tables_batched(pc, dev, use_hip=use_hip)
        assert (cm, k_rows, n_states) == ((14, k, [12, 3, 5, 14]) if variant != 'big' else (32, k, [32, 3, 23]))
        g = torch.Generator(device='cpu').manual_seed(3)
        loss = sum((st[n] * torch.randn(st[n].shape, generator=g, dtype=torch.float64).to(dev)).sum() for n in names)
        loss.backward()
        tabs[use_hip] = {n: st[n].detach().cpu().numpy() for n in names}
        grads[use_hip] = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}
    for n in names:
        np.testing.assert_allclose(tabs[True][n], tabs[False][n], rtol=1e-13, atol=1e-12, err_msg=n)
    assert set(grads[True]) == set(grads[False]) == {'init_logits', 'transition_logits', 'poisson_log_rates', 'gaussian_means'}
    for n in grads[False]:
        np.testing.assert_allclose(grads[True][n], grads[False][n], rtol=2e-5, atol=1e-5 * np.abs(grads[False][n]).max(),
                                   err_msg=n)
