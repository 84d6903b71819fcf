"""Factored C oracle (oracle/smm_oracle.c) against the dense restatement of the reference path (CPU only)."""
import numpy as np
import pytest
import torch

from oracle import dense_ref as O
from oracle import factored as F
from golden_util import CASES, case_inputs, assert_spans_equivalent


def random_problem(seed, b, tmax, c, k, integer=False, ends=False):
    g = torch.Generator().manual_seed(seed)
    lengths = torch.randint(max(1, tmax // 2), tmax + 1, (b,), generator=g)
    lengths[int(torch.randint(0, b, (1,), generator=g))] = tmax
    if integer:
        r = lambda *s: torch.randint(-4, 1, s, generator=g).double()
    else:
        r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 2 - 1
    elp, trans, init, lens = r(b, tmax, c), r(c, c), r(c), r(k, c)
    allowed = None
    if ends:
        allowed = [sorted(set(torch.randint(0, c, (2,), generator=g).tolist())) for _ in range(b)]
    return elp, lengths, trans, init, lens, allowed


def dense_decode(elp, lengths, trans, init, lens, allowed):
    scores = O.log_hsmm(trans, elp, init, lens, lengths, add_eos=True, allowed_ends_per_instance=allowed)
    v, segs = O.viterbi_backpointers(scores, lengths + 1)
    return scores, v, O.spans_from_segments(segs, scores.shape[1] + 1)


@pytest.mark.parametrize('seed', range(8))
def test_integer_lattices_are_bit_identical_including_tie_order(seed):
    """All arithmetic is exact on small integers, so value AND the (k asc, from asc) tie order must agree."""
    b, tmax, c, k = 3, 9 + seed % 3, 3, 3 + seed % 4
    elp, lengths, trans, init, lens, allowed = random_problem(seed, b, tmax, c, k, integer=True, ends=seed % 2 == 1)
    scores, v, spans = dense_decode(elp, lengths, trans, init, lens, allowed)
    fs, fv = F.viterbi(elp.numpy(), lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy(),
                       F.endpen_from_allowed_ends(allowed, b, c))
    np.testing.assert_array_equal(fv, v.numpy())
    np.testing.assert_array_equal(fs, spans.numpy())


@pytest.mark.parametrize('seed', range(8))
@pytest.mark.parametrize('shape', [(3, 12, 3, 4), (2, 40, 6, 8), (2, 5, 3, 8), (2, 30, 5, 31)])
def test_random_lattices_match_dense_oracle(seed, shape):
    b, tmax, c, k = shape
    elp, lengths, trans, init, lens, allowed = random_problem(100 + seed, b, tmax, c, k, ends=seed % 3 == 0)
    scores, v, spans = dense_decode(elp, lengths, trans, init, lens, allowed)
    fs, fv = F.viterbi(elp.numpy(), lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy(),
                       F.endpen_from_allowed_ends(allowed, b, c))
    np.testing.assert_allclose(fv, v.numpy(), rtol=1e-12, atol=1e-9)
    assert_spans_equivalent(fs, spans.numpy(), lengths, c, scores, v, lengths + 1)


def test_known_answer_through_factored_path():
    """Inputs of src/models/test_semimarkov.py:266-323 given to the factored DP directly."""
    b, c, n, k, step = 10, 4, 100, 5, 4
    padded = n + 2 * step
    lengths = np.full(b, n); lengths[0] = padded
    em = np.full((b, padded, c), -1e9)
    for t in range(padded):
        em[:, t, (t // step) % c] = 1
    init = np.full(c, -1e9); init[0] = 0
    ls = np.full((k, c), -1e9); ls[step] = 0
    spans, v = F.viterbi(em, lengths, np.zeros((c, c)), init, ls)
    for s in range(n // step):
        assert (spans[:, step * s] == s % c).all()
    assert (spans[np.arange(b), lengths] == c).all()
    np.testing.assert_array_equal(v, [108.] + [100.] * 9)


@pytest.mark.parametrize('case', [c for c in CASES if CASES[c].get('add_eos', True)])
def test_golden_cases_frame_labels(golden, case):
    p, feats, lengths, valid, cons, cfg = case_inputs(golden, case, torch.float64)
    r = O.viterbi_full(p, feats, lengths, valid, True, cfg.get('additional'), cons)
    trans, init, lens, merged = O.factor_tables(p, valid)
    ends = O.allowed_ends_for_batch(p, valid, cfg.get('additional'), feats.shape[0])
    c = trans.shape[0]
    elp = F.emission(feats.numpy(), lengths.numpy(), p.gaussian_means[merged].numpy(),
                     (1.0 / p.gaussian_cov_diag).numpy(),
                     float(-0.5 * feats.shape[2] * np.log(2 * np.pi) - 0.5 * p.gaussian_cov_diag.log().sum()),
                     None if cons is None else cons.numpy())
    np.testing.assert_allclose(elp, r['elp'].numpy(), rtol=1e-12, atol=1e-10)
    fs, fv = F.viterbi(elp, lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy(),
                       F.endpen_from_allowed_ends(ends, feats.shape[0], c))
    np.testing.assert_allclose(fv, r['v'].numpy(), rtol=1e-12, atol=1e-9)
    for i, t in enumerate(lengths.tolist()):
        assert fs[i, t] == c
        np.testing.assert_array_equal(O.spans_to_labels(fs[i:i + 1, :t]),
                                      O.spans_to_labels(r['local_spans'][i:i + 1, :t].numpy()))
    # boundaries may differ only inside runs of one class (mathematical ties): certificate by re-scoring
    np.testing.assert_allclose(O.rescore(r['scores'], torch.from_numpy(fs), r['pos_lengths']).numpy(),
                               r['v'].numpy(), rtol=1e-9, atol=1e-7)


@pytest.mark.parametrize('seed', range(4))
def test_logz_and_posteriors_match_dense_autograd(seed):
    b, tmax, c, k = 2, 11, 3, 5
    elp, lengths, trans, init, lens, allowed = random_problem(200 + seed, b, tmax, c, k, ends=seed % 2 == 0)
    leaves = [t.clone().requires_grad_(True) for t in (elp, trans, init, lens)]
    scores = O.log_hsmm(leaves[1], leaves[0], leaves[2], leaves[3], lengths, True, allowed)
    z, _ = O.semimarkov_dp(scores, lengths + 1, O.LogSemiring)
    up = torch.tensor([1.0, 0.5], dtype=torch.float64)
    (z * up).sum().backward()
    fz, g = F.logz(elp.numpy(), lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy(),
                   F.endpen_from_allowed_ends(allowed, b, c), grad=True, upstream=up.numpy())
    np.testing.assert_allclose(fz, z.detach().numpy(), rtol=1e-12)
    for name, leaf in zip(('elp', 'trans', 'init', 'len'), leaves):
        np.testing.assert_allclose(g[name], leaf.grad.numpy(), rtol=1e-9, atol=1e-11, err_msg=name)


@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('integer', [True, False])
def test_no_eos_factored_matches_dense(seed, integer):
    """add_eos=False (reference semimarkov_modules.py:494-505): the C twin's closed form of the last position against
    the dense potentials of log_hsmm(add_eos=False) scanned by the restated pytorch-struct DP."""
    b, tmax, c, k = 3, 9 + seed, 3 + seed % 2, 3 + seed % 4
    elp, lengths, trans, init, lens, _ = random_problem(300 + seed, b, tmax, c, k, integer=integer)
    lengths = lengths.clamp(min=2)
    scores = O.log_hsmm(trans, elp, init, lens, lengths, add_eos=False)
    v, segs = O.viterbi_backpointers(scores, lengths)
    spans = O.spans_from_segments(segs, scores.shape[1] + 1)
    fs, fv = F.viterbi(elp.numpy(), lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy(), None, no_eos=True)
    fs = fs[:, :tmax]
    if integer:
        np.testing.assert_array_equal(fv, v.numpy())
        np.testing.assert_array_equal(fs, spans.numpy())
    else:
        np.testing.assert_allclose(fv, v.numpy(), rtol=1e-12, atol=1e-9)
        for i, t in enumerate(lengths.tolist()):     # same frame labels (boundaries inside a run of one class may differ)
            np.testing.assert_array_equal(O.spans_to_labels(fs[i:i + 1, :t]), O.spans_to_labels(spans.numpy()[i:i + 1, :t]))
    assert (fs != c).all()                      # no EOS label anywhere
    for i, t in enumerate(lengths.tolist()):
        assert fs[i, t - 1] >= 0                # the last frame starts the closing span


def test_factored_matches_dense_at_the_largest_dense_shape():
    """Factored vs dense agreement at CrossTask magnitudes on the largest lattice whose dense potentials fit the CPU
    suite comfortably (T = 2000, K = 256, C = 7: 262 MB of fp64 potentials): per-frame emission ~ -290 (D = 200), so cumE
    reaches 6e5 and h = beta - cumE cancels ~19 bits -- the prefix-sum form must still pick the dense arg-max."""
    g = torch.Generator().manual_seed(5)
    t, k, c, d = 2000, 256, 7, 200
    mu = torch.randn(c, d, generator=g, dtype=torch.float64) * 0.3
    labels = torch.repeat_interleave(torch.arange(40) % c, 50)[:t]
    x = mu[labels] + torch.randn(t, d, generator=g, dtype=torch.float64)
    elp = (-0.5 * ((x[:, None, :] - mu[None]) ** 2).sum(-1) - 0.5 * d * np.log(2 * np.pi))[None]
    assert -330 < float(elp.max(-1).values.mean()) < -250
    trans = torch.log_softmax(torch.randn(c, c, generator=g, dtype=torch.float64), 0)
    init = torch.log_softmax(torch.randn(c, generator=g, dtype=torch.float64), 0)
    kk = torch.arange(k, dtype=torch.float64)[:, None]
    rate = torch.rand(c, generator=g, dtype=torch.float64) * 60 + 20
    lens = kk * rate.log() - rate - torch.lgamma(kk + 1)
    lengths = torch.tensor([t])
    scores = O.log_hsmm(trans, elp, init, lens, lengths, add_eos=True)
    v, segs = O.viterbi_backpointers(scores, lengths + 1)
    spans = O.spans_from_segments(segs, scores.shape[1] + 1)
    fs, fv = F.viterbi(elp.numpy(), lengths.numpy(), trans.numpy(), init.numpy(), lens.numpy(), None)
    np.testing.assert_allclose(fv, v.numpy(), rtol=1e-13)
    np.testing.assert_array_equal(O.spans_to_labels(fs)[0, :t], O.spans_to_labels(spans.numpy())[0, :t])
    assert_spans_equivalent(fs, spans.numpy(), lengths, c, scores, v, lengths + 1)
