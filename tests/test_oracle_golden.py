"""The oracle against the reference's golden vectors / known answers (CPU only)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dense_ref as O
from golden_util import CASES, case_inputs, span_start_differences, crosstask_magnitude_case, fp32_near_tie_certificate

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('case', list(CASES))
@pytest.mark.parametrize('tag', ['f32', 'f64'])
def test_scorers_and_dense_scores_match_reference(golden, case, tag):
    dtype = torch.float32 if tag == 'f32' else torch.float64
    tol = dict(rtol=2e-5, atol=2e-4) if tag == 'f32' else dict(rtol=1e-12, atol=1e-9)
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, dtype)
    trans, init, lens, merged = O.factor_tables(p, valid)
    pre = '%s/%s/' % (case, tag)
    np.testing.assert_allclose(init.numpy(), golden[pre + 'init'], **tol)
    np.testing.assert_allclose(trans.numpy(), golden[pre + 'trans'], **tol)
    # the reference evaluates lgamma(k+1) on a .float() time axis even in its fp64 run (modules:387-388),
    # so its fp64 length table (and everything containing it) carries fp32 rounding of lgamma
    ltol = tol if tag == 'f32' else dict(rtol=1e-7, atol=2e-6)
    np.testing.assert_allclose(lens.numpy(), golden[pre + 'len'], **ltol)
    scores, elp = O.score_features(p, feats, lengths, valid, cfg.get('add_eos', True), cfg.get('additional'), cons)
    np.testing.assert_allclose(elp.numpy(), golden[pre + 'elp'], **tol)
    assert scores.shape == golden[pre + 'scores'].shape
    np.testing.assert_allclose(scores.numpy(), golden[pre + 'scores'], **ltol)


@pytest.mark.parametrize('case', [c for c in CASES if CASES[c].get('add_eos', True)])
def test_module_level_viterbi_matches_reference_host_code(golden, case):
    """Reference host code (class un-mapping, EOS) + restated DP.  Span boundaries inside a run of ONE class
    are mathematically tied ((a,b) vs (b,a) splits score the same), so they are compared through the frame
    labels plus a re-scoring certificate; everything else must be identical."""
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    r = O.viterbi_full(p, feats, lengths, valid, True, cfg.get('additional'), cons)
    ref = torch.from_numpy(golden[case + '/f64/ref_spans'])
    for i, t in enumerate(lengths.tolist()):
        assert ref[i, t] == p.n_classes and r['spans'][i, t] == p.n_classes  # EOS id at position lengths[i]
        assert (ref[i, t + 1:] == -1).all() and (r['spans'][i, t + 1:] == -1).all()
        np.testing.assert_array_equal(O.spans_to_labels(r['spans'][i:i + 1, :t].numpy()),
                                      O.spans_to_labels(ref[i:i + 1, :t].numpy()))
    ref_local = O.map_spans_to_local(ref, valid, p.n_classes)
    np.testing.assert_allclose(O.rescore(r['scores'], ref_local, r['pos_lengths']).numpy(), r['v'].numpy(),
                               rtol=1e-7, atol=1e-6)
    logz = O.log_partition(p, feats, lengths, valid, True, cfg.get('additional'), cons)
    np.testing.assert_allclose(logz.mean().item(), float(golden[case + '/f64/ref_mean_logz']), rtol=1e-7)
    assert (logz >= r['v'] - 1e-9).all()


def test_known_answer_log_hsmm(golden):
    """src/models/test_semimarkov.py:266-323: periodic labels every 4 frames, EOS at lengths-1."""
    scores = torch.from_numpy(golden['kat/scores'])
    lengths = torch.from_numpy(golden['kat/lengths']) + 1
    b, c, n, step = 10, 4, 100, 4
    for decode in ('autograd', 'backpointers'):
        if decode == 'autograd':
            v, parts = O.marginals(scores, lengths, O.MaxSemiring)
            seq = O.from_parts(parts)
        else:
            v, segs = O.viterbi_backpointers(scores, lengths)
            seq = O.spans_from_segments(segs, scores.shape[1] + 1)
        for s in range(n // step):
            assert (seq[:, step * s] == s % c).all()
        assert (seq[torch.arange(b), lengths - 1] == c).all()
        np.testing.assert_allclose(v.numpy(), [108.] + [100.] * 9)


@pytest.mark.parametrize('seed', range(6))
def test_backpointer_viterbi_equals_autograd_argmax(seed):
    g = torch.Generator().manual_seed(seed)
    b, n, k, c = 3, 9, 4, 3
    # small integers -> many exact ties, so the tie ORDER is what is being compared
    edge = torch.randint(-3, 3, (b, n - 1, k, c, c), generator=g).double()
    lengths = torch.tensor([n, n - 2, n - 1])
    v1, parts = O.marginals(edge, lengths, O.MaxSemiring)
    v2, segs = O.viterbi_backpointers(edge, lengths)
    np.testing.assert_array_equal(v1.numpy(), v2.numpy())
    np.testing.assert_array_equal(parts.numpy(), O.parts_from_segments(segs, edge.shape, torch.float64).numpy())
    seq = O.from_parts(parts)
    np.testing.assert_array_equal(O.to_parts(seq, c, k, lengths).numpy()[parts.numpy() > 0], 1)


@pytest.mark.parametrize('seed', range(4))
def test_dp_equals_brute_force_enumeration(seed):
    g = torch.Generator().manual_seed(seed)
    n, k, c = 7, 4, 3
    edge = torch.randn(2, n - 1, k, c, c, generator=g, dtype=torch.float64)
    lengths = torch.tensor([n, n - 2])
    vmax, _ = O.semimarkov_dp(edge, lengths, O.MaxSemiring)
    vlog, _ = O.semimarkov_dp(edge, lengths, O.LogSemiring)
    for i in range(2):
        assert abs(float(vmax[i]) - O.brute_force(edge[i], lengths[i], 'max')) < 1e-10
        assert abs(float(vlog[i]) - O.brute_force(edge[i], lengths[i], 'log')) < 1e-10


def test_sliding_sum_probe_value():
    x = torch.arange(6.).view(1, 6, 1)
    assert O.sliding_sum(x, 3).flatten().tolist() == [3, 6, 9, 12, 9, 5]


def test_codecs_match_reference_vectors():
    with open(os.path.join(HERE, 'golden', 'codec_vectors.json')) as f:
        cv = json.load(f)
    assert O.labels_to_spans(cv['labels'], cv['max_k']).tolist() == cv['spans']
    assert cv['spans'] == [[0, 1, -1, 2, -1, -1], [0, 1, 2, 3, -1, 4]]  # test_semimarkov.py:252
    assert O.spans_to_labels(cv['spans']).tolist() == cv['back'] == cv['labels']
    assert [[list(t) for t in r] for r in O.rle_spans(cv['spans'], [6, 6])] == cv['rle']
    assert [[list(t) for t in r] for r in O.rle_spans(cv['spans'], cv['trunc_lengths'])] == cv['rle_trunc']
    assert O.labels_to_spans(cv['rand_labels'], cv['rand_max_k']).tolist() == cv['rand_spans']
    assert O.spans_to_labels(cv['rand_spans']).tolist() == cv['rand_back'] == cv['rand_labels']


def test_fit_supervised_matches_reference(golden):
    feats = [golden['fit/features%d' % i] for i in range(5)]
    labels = [golden['fit/labels%d' % i] for i in range(5)]
    out = O.fit_supervised(feats, labels, int(golden['fit/n_classes']), int(golden['fit/max_k']))
    for name, val in out.items():
        np.testing.assert_allclose(val, golden['fit/param/' + name], rtol=2e-6, atol=1e-6, err_msg=name)


# ----------------------------------------------------------------------------------------------- the fp32 clause of SURVEY 8c(1)
EOS_CASES = [c for c in CASES if CASES[c].get('add_eos', True)]


def test_reference_fp32_run_differs_from_its_own_fp64_run_only_inside_one_class_runs(golden):
    """The fixtures hold the reference's host code run in fp32 AND in fp64 (``*/f32/ref_spans``, ``*/f64/ref_spans``).
    "Boundaries bit-exact against the fp32 reference" is not a property the reference has against ITSELF: on two of the
    five cases the two runs draw a boundary between consecutive spans of one class at different positions (equal frame
    labels, equal EOS placement).  Pinned here so that the parity bar is stated against facts."""
    differing = {}
    for case in EOS_CASES:
        p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
        a, b = golden[case + '/f32/ref_spans'], golden[case + '/f64/ref_spans']
        for i, t in enumerate(lengths.tolist()):
            assert a[i, t] == b[i, t] == p.n_classes
        differing[case] = len(span_start_differences(a, b, lengths))
    assert differing == {'tiny': 0, 'subset_merge': 0, 'k_gt_t': 2, 'hmm_k1': 0, 'constrained': 2}, differing


@pytest.mark.parametrize('case', EOS_CASES)
def test_factored_twin_against_the_reference_fp32_spans(golden, case):
    """The C twin (what the HIP kernel equals bit for bit) against the reference's own fp32 run: same frame labels, same
    EOS; span starts differ only inside one-class runs."""
    from oracle import factored as F
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    b = feats.shape[0]
    trans, init, lens, merged = O.factor_tables(p, valid)
    elp = O.emission_log_probs(feats, p.gaussian_means[merged], p.gaussian_cov_diag, cons)
    ends = O.allowed_ends_for_batch(p, valid, cfg.get('additional'), b)
    fs, fv = F.viterbi(elp.numpy(), lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy(),
                       F.endpen_from_allowed_ends(ends, b, trans.shape[0]))
    table = np.array((list(range(p.n_classes)) if valid is None else [int(v) for v in valid]) + [p.n_classes, -1])
    span_start_differences(table[fs], golden[case + '/f32/ref_spans'], lengths)


def test_fp32_near_tie_certificate_at_crosstask_magnitudes_factored_twin():
    """SURVEY App. C.3 / 8c(1) on the CPU (the GPU test of the same name runs the HIP path): T = 800, 8 states, K = 24,
    D = 200, twelve seeds.  The fp64 factored path equals the reference path run in fp64 frame for frame; against the
    reference path run in fp32 it differs on one of the twelve videos (4 frames), where it scores HIGHER than the fp32
    run's own path under exact potentials and lies within 4 fp32 ulps (2^-6 each at |v| = 2.2e5) of the fp32 optimum."""
    from oracle import factored as F
    n_diff = 0
    for seed in range(12):
        p32, feats, lengths = crosstask_magnitude_case(seed)
        p64 = p32.to(torch.float64)
        trans, init, lens, merged = O.factor_tables(p64, None)
        elp = O.emission_log_probs(feats.double(), p64.gaussian_means[merged], p64.gaussian_cov_diag)
        fs, fv = F.viterbi(elp.numpy(), lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy())
        cert = fp32_near_tie_certificate(p32, feats, lengths, fs)
        assert cert['labels_equal_fp64_run'], (seed, cert)
        assert abs(cert['ulps_from_fp32_optimum']) <= 4.0, (seed, cert)
        assert cert['gain_over_fp32_path'] >= 0.0, (seed, cert)
        if cert['frames_differing']:
            n_diff += 1
            assert cert['gain_over_fp32_path'] > 0.0, (seed, cert)
    assert n_diff == 1
