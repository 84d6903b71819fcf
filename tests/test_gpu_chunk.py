"""Time-split Viterbi decode of long videos (csrc/smm_chunk.hip): a video cut along the TIME axis into units that run side by
side, certified and stitched -- or decoded again in one piece when a cut cannot be certified or a decision is closer than
rounding can tell.  Either way the outputs must be the one-piece decode's, i.e. the C twin's, bit for bit (reference
semimarkov_modules.py:677-679 per video).  SMM_CHUNK_P forces small units so that moderate videos are cut many times."""
import numpy as np
import pytest
import torch

import test_gpu_viterbi as tv
from test_gpu_fullsize import make_corpus, decode_both, check_equivalent

pytestmark = pytest.mark.gpu


def decode_split_and_whole(p, monkeypatch, unit, wc=None, class_map=None):
    monkeypatch.setenv('SMM_CHUNK', '1')
    monkeypatch.setenv('SMM_CHUNK_P', str(unit))
    if wc is not None:
        monkeypatch.setenv('SMM_CHUNK_WC', str(wc))
    split = tv.run_gpu(p, class_map=class_map)
    monkeypatch.setenv('SMM_CHUNK', '0')
    whole = tv.run_gpu(p, class_map=class_map)
    assert whole['_err'][4] == 0 and whole['_err'][5] == 0
    for key in ('best', 'spans', 'labels', 'n_segs'):
        np.testing.assert_array_equal(split[key], whole[key], err_msg=key)
    spans, v = tv.run_oracle(p)
    if class_map is None:
        tv.check(p, split, spans, v)
    assert split['_err'][0] == 0
    return split


@pytest.mark.parametrize('shape', [([6000, 2500, 5200], 13, 1024), ([9000], 23, 1024), ([4200, 4100], 7, 600), ([7000, 800], 28, 1024), ([6100], 32, 1024)])
def test_time_split_decode_equals_the_one_piece_decode_on_structured_lattices(shape, monkeypatch):
    """CrossTask-like lattices (one state explains the frames, Poisson lengths): every cut certifies, every decision is clear
    of rounding, nothing is decoded twice -- and not a bit differs from the one-piece decode and the C twin."""
    lengths, c, k = shape
    p = tv.structured_problem(hash((tuple(lengths), c)) % 1000 + 1, lengths, c, k)
    out = decode_split_and_whole(p, monkeypatch, unit=1)          # the smallest units the planner makes: warm-up + 2 (kp - 1)
    n_long = sum(1 for t in lengths if t >= 2 * (512 + 2 * (min(k, max(lengths)) - 1)))
    assert out['_err'][4] >= max(1, n_long - 1), out['_err']
    assert out['_err'][5] == 0, out['_err']


def test_time_split_decode_with_end_penalties_and_class_map(monkeypatch):
    p = tv.structured_problem(5, [5000, 4700], 9, 1024)
    g = np.random.default_rng(3)
    p['endpen'] = np.full((2, 9), -1e9)
    for i in range(2):
        p['endpen'][i, g.integers(0, 9, size=3)] = 0.0
    cmap = np.array([40, 41, 7, 3, 99, 12, 13, 14, 15, 1000])
    out = decode_split_and_whole(p, monkeypatch, unit=1, class_map=cmap)
    assert out['_err'][4] == 2
    spans, v = tv.run_oracle(p)
    np.testing.assert_array_equal(out['best'], v)
    np.testing.assert_array_equal(out['spans'], np.where(spans >= 0, cmap[np.maximum(spans, 0)], -1))


@pytest.mark.parametrize('kind', ['random', 'integer', 'flat'])
def test_time_split_decode_repairs_what_it_cannot_certify(kind, monkeypatch):
    """Lattices on which the recursion does NOT forget its start (unstructured random potentials), on which every decision is
    an exact tie (small integers) or nothing ever falls behind (flat): cuts fail to certify or decisions sit inside the
    rounding margin, the videos are decoded again in one piece by the same call -- same bits as ever."""
    b, tmax, c, k = 2, 3600, 6, 520
    p = tv.make_problem(11, b, tmax, c, k, integer=(kind == 'integer'))
    p['lengths'] = np.asarray([tmax, 3300])
    if kind == 'flat':
        g = np.random.default_rng(17)
        p['elp'] = p['elp'][:, :, :1] + 1e-3 * g.standard_normal(p['elp'].shape)
        p['lens'] = -np.log(k) - 0.05 * g.random(p['lens'].shape)
        tr = g.standard_normal(p['trans'].shape)
        p['trans'] = tr - np.log(np.exp(tr).sum(0, keepdims=True))
        p['init'] = np.full_like(p['init'], -np.log(c))
    out = decode_split_and_whole(p, monkeypatch, unit=1)
    assert out['_err'][4] == 2, out['_err']
    if kind == 'integer':
        assert out['_err'][5] == 2, out['_err']          # exact ties everywhere: never clear of the margin


def test_time_split_decode_on_the_ring_kernels(monkeypatch):
    """Span limits up to 512 (the ring kernels, no BAND mode): the same units, the same stitch, the [k][c] length table;
    span limits up to 64 (the window back-trace's launches) are left alone."""
    p = tv.structured_problem(21, [3000, 2900, 700], 16, 256, rate=(10, 120))
    out = decode_split_and_whole(p, monkeypatch, unit=1)
    assert out['_err'][4] == 2 and out['_err'][5] == 0, out['_err']
    p = tv.structured_problem(22, [2500, 2400], 9, 400, rate=(10, 200))
    out = decode_split_and_whole(p, monkeypatch, unit=1)
    assert out['_err'][4] == 2, out['_err']
    p = tv.structured_problem(23, [1500, 1400], 12, 40, rate=(4, 20))
    out = decode_split_and_whole(p, monkeypatch, unit=1)
    assert out['_err'][4] == 0 and out['_err'][5] == 0, out['_err']


def test_cfg2_batch_of_64_is_split_by_default(monkeypatch):
    """BASELINE configs[1] (64 videos x 2048 frames, 16 states, K = 256): 64 one-CU videos on 256 CUs -- the planner cuts each
    into units and the outputs stay the C twin's."""
    monkeypatch.delenv('SMM_CHUNK', raising=False)
    monkeypatch.delenv('SMM_CHUNK_P', raising=False)
    cp = make_corpus(2, [2048] * 64, 16, 256, rate=(20, 200))
    res = decode_both(cp)
    check_equivalent(cp, *res)
    err = res[0]['_err'].cpu().numpy()
    assert err[4] == 64, err


def test_one_class_run_ties_are_resolved_not_repaired(monkeypatch):
    """Runs that last about TWICE what their class's Poisson length table expects are decoded as two spans of one class, and the
    two orders of the pair -- (k2, k1) and (k1, k2) -- tie to within rounding: two lengths inside the stitch's margin.
    The stitch verifies both orders and lets the exact score pass choose as the one-piece decode does; the outputs are the
    one-piece decode's and the twin's, span starts included, and (nearly) nothing goes to the repair launch."""
    from scipy.special import gammaln
    lengths, c, k = [7000, 6500, 6800, 7200], 9, 1024
    seed = 31
    p = tv.structured_problem(seed, lengths, c, k, rate=(150, 260))
    # structured_problem's own length tables (the rates its runs were drawn with), except for ONE class whose table expects
    # 0.55 of what its runs last: that class's runs -- one in nine -- are worth cutting in two, the others are not
    g0 = np.random.default_rng(seed)
    rates = g0.uniform(150, 260, size=c)                        # (the generator's first draw: the true rates)
    rates[3] *= 0.55
    kk = np.arange(k)[:, None]
    p['lens'] = kk * np.log(rates) - rates - gammaln(kk + 1)
    g = np.random.default_rng(4)
    tr = np.full((c, c), -6.0) + g.uniform(-0.5, 0.5, size=(c, c))
    np.fill_diagonal(tr, -1.5)                                  # a self transition is affordable
    p['trans'] = tr - np.log(np.exp(tr).sum(0, keepdims=True))
    out = decode_split_and_whole(p, monkeypatch, unit=1)
    spans, _ = tv.run_oracle(p)
    same_class_pairs = 0
    for i, t in enumerate(lengths):
        labs = spans[i, :t][spans[i, :t] >= 0]
        same_class_pairs += int((labs[1:] == labs[:-1]).sum())
    assert same_class_pairs >= 4, same_class_pairs               # the lattice does what it was built for
    assert out['_err'][4] == 4 and out['_err'][7] >= 8, out['_err']   # (the twin shows four two-span runs per video)
    assert out['_err'][5] == 0, out['_err']


def test_a_short_warm_up_fails_the_certificate_not_the_decode(monkeypatch):
    """With a warm-up of 16 positions the units have not forgotten their start when their certified window begins: the cut
    does not certify, the video is repaired -- the outputs never depend on the split having worked."""
    p = tv.structured_problem(8, [6000], 11, 1024, rate=(200, 400))
    out = decode_split_and_whole(p, monkeypatch, unit=1, wc=16)
    assert out['_err'][4] == 1


def test_cfg1_one_long_video_is_split_by_default(monkeypatch):
    """BASELINE configs[0] (one video, T = 10 000, 20 states, K = 1024, D = 200) through smm_decode_f32 with the planner's own
    unit size: one video on a GPU of 256 CUs is cut into as many units as the certified overlap allows."""
    monkeypatch.delenv('SMM_CHUNK', raising=False)
    monkeypatch.delenv('SMM_CHUNK_P', raising=False)
    cp = make_corpus(1, [10000], 20, 1024)
    res = decode_both(cp)
    check_equivalent(cp, *res)
    err = res[0]['_err'].cpu().numpy()
    assert err[4] == 1 and err[5] == 0, err
