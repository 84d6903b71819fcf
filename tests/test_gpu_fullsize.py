"""Feature-driven parity at BASELINE.json's full sizes: features (D = 200, per-frame emission ~ -290 like PCA-200
CrossTask features) -> smm_decode_f32 (emission kernel + Viterbi kernel) against the C twin's emission + Viterbi
(reference semimarkov_modules.py:660-696 end to end).  At T = 14 000 the prefix sums reach 4e6 and h = beta - cumE
cancels 20+ bits: this is where the factored form has to be right, not just fast."""
import numpy as np
import pytest
import torch

from oracle import dense_ref as O
from oracle import factored as F

pytestmark = pytest.mark.gpu


def make_corpus(seed, lengths, c, k, d=200, rate=(20, 400)):
    """One task: HSMM-sampled labels, x = mu[label] + sigma * eps (synth.py's generator, inlined for explicit shapes)."""
    g = np.random.default_rng(seed)
    sigma = g.uniform(0.7, 1.3, size=d)
    mu = g.normal(0, 0.3, size=(c, d))
    rates = g.uniform(rate[0], rate[1], size=c)
    xs, labs = [], []
    for t in lengths:
        out, cur, tot = [], 0, 0
        while tot < t:
            ln = int(np.clip(g.poisson(rates[cur]), 1, k - 1))
            out.append(np.full(ln, cur)); tot += ln; cur = (cur + 1) % c
        lab = np.concatenate(out)[:t]
        xs.append((mu[lab] + sigma * g.standard_normal((t, d))).astype(np.float32))
        labs.append(lab)
    # parameters: the truth, slightly perturbed (so that the decode is not trivially the ground truth)
    mu_hat = mu + g.normal(0, 0.02, size=mu.shape)
    var = sigma ** 2
    trans = np.log(g.dirichlet(np.ones(c) * 0.5, size=c).T + 1e-3)
    trans -= np.log(np.exp(trans).sum(0, keepdims=True))
    init = np.log(g.dirichlet(np.ones(c)))
    kk = np.arange(k)[:, None]
    from scipy.special import gammaln
    lens = kk * np.log(rates) - rates - gammaln(kk + 1)
    return dict(xs=xs, labs=labs, mu=mu_hat, var=var, trans=trans, init=init, lens=lens, c=c, k=k, d=d)


def decode_both(cp):
    from action_segmentation_amd import ops
    dev = torch.device('cuda:0')
    lengths = np.array([x.shape[0] for x in cp['xs']], dtype=np.int64)
    b, tmax, c, k, d = len(lengths), int(lengths.max()), cp['c'], cp['k'], cp['d']
    off = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    x = torch.from_numpy(np.concatenate(cp['xs'], 0)).to(dev)
    mu, var = cp['mu'], cp['var']
    w = (mu / var).T.copy()
    lognorm = -0.5 * np.log(var).sum() - 0.5 * d * np.log(2 * np.pi)
    cst = -0.5 * (mu * mu / var).sum(1) + lognorm
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
    batch = ops.Batch(lengths, [c], k, c_max=c, frame_offset=off, kp=[min(k, tmax)] * b, d=d, t_max=tmax,
                      total_frames=int(lengths.sum()))
    out = ops.decode(batch, x, t(w[None]), t(cst[None]), t(1.0 / var), t(cp['trans'][None]), t(cp['init'][None]),
                     t(cp['lens'][None]), want_elp=True)
    torch.cuda.synchronize()
    ops.check_decoded(batch, out)
    # the C twin: emission in the direct (x - mu)^2 form, fp64
    xp = np.zeros((b, tmax, d), np.float32)
    for i, xi in enumerate(cp['xs']):
        xp[i, :xi.shape[0]] = xi
    elp = F.emission(xp, lengths, mu, 1.0 / var, lognorm)
    spans, v = F.viterbi(elp, lengths, cp['trans'], cp['init'], cp['lens'])
    return out, spans, v, elp, lengths, off


def check_equivalent(cp, out, spans, v, elp, lengths, off):
    c = cp['c']
    got = out['spans'].cpu().numpy()
    labels = out['labels'].cpu().numpy()
    best = out['best'].cpu().numpy()
    # emission: expanded form on the fp64 matrix cores vs the direct form, |elp| ~ 300
    e32 = out['elp'].cpu().numpy()
    for i, t in enumerate(lengths):
        np.testing.assert_allclose(e32[off[i]:off[i] + t], elp[i, :t], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(best, v, rtol=1e-12)           # path scores ~1e6: agree to 1e-6 absolute
    n_diff = 0
    for i, t in enumerate(lengths):
        assert got[i, t] == spans[i, t] == c                      # EOS where the video ends
        ref_lab = O.spans_to_labels(spans[i:i + 1, :t])[0]
        np.testing.assert_array_equal(labels[off[i]:off[i] + t], ref_lab, err_msg='video %d' % i)
        np.testing.assert_array_equal(O.spans_to_labels(got[i:i + 1, :t])[0], ref_lab)
        n_diff += int((got[i, :t] != spans[i, :t]).sum())
    return n_diff


def test_cfg1_one_long_video():
    """BASELINE configs[0]: one video, T = 10 000, 20 states, K = 1024, D = 200 (a gang of three CUs)."""
    cp = make_corpus(1, [10000], 20, 1024)
    res = decode_both(cp)
    check_equivalent(cp, *res)
    acc = np.mean(res[0]['labels'].cpu().numpy() == cp['labs'][0])
    assert acc > 0.98


def test_cfg2_batch_of_64():
    """BASELINE configs[1]: 64 videos, T = 2048, 16 states, K = 256, D = 200."""
    cp = make_corpus(2, [2048] * 64, 16, 256, rate=(20, 200))
    res = decode_both(cp)
    check_equivalent(cp, *res)


@pytest.mark.parametrize('c', [23, 21, 15])
def test_cfg3_longest_videos(c):
    """BASELINE configs[2] at its extremes: T = 14 000 (cumE ~ 4e6), K = 1024, 23 states (always a gang), 21 (the most one
    workgroup holds) and 15, next to shorter videos of the same task in one ragged launch."""
    cp = make_corpus(30 + c, [14000, 9000, 14000, 600, 3000], c, 1024)
    res = decode_both(cp)
    check_equivalent(cp, *res)


def test_cfg3_gangs_and_singles_agree(monkeypatch):
    """The same ragged launch with gangs switched off decodes to the same bits."""
    from action_segmentation_amd import ops
    cp = make_corpus(77, [14000, 12000, 5000], 19, 1024)
    a = decode_both(cp)
    monkeypatch.setenv('SMM_PAIRS', '0')
    b = decode_both(cp)
    np.testing.assert_array_equal(a[0]['spans'].cpu().numpy(), b[0]['spans'].cpu().numpy())
    np.testing.assert_array_equal(a[0]['best'].cpu().numpy(), b[0]['best'].cpu().numpy())
