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
    """BASELINE configs[0]: one video, T = 10 000, 20 states, K = 1024, D = 200 (one workgroup, BAND mode)."""
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
    """BASELINE configs[2] at its extremes: T = 14 000 (cumE ~ 4e6), K = 1024, 23 states (CrossTask's largest task), 21
    and 15, next to shorter videos of the same task in one ragged launch."""
    cp = make_corpus(30 + c, [14000, 9000, 14000, 600, 3000], c, 1024)
    res = decode_both(cp)
    check_equivalent(cp, *res)


@pytest.mark.parametrize('c', [23, 12])
def test_cfg3_longest_videos_at_d300(c):
    """The reference's i3d + resnet + audio feature setting after PCA gives D = 300 (src/main.py:283-286): the emission
    kernel's LDS weight table is 8 * 304 * 37 bytes = 90 KB above 16 states there (one workgroup per CU instead of two).
    Feature-driven, full size: T = 14 000, K = 1024 next to short videos, against the C twin."""
    cp = make_corpus(300 + c, [14000, 700, 9000, 14000], c, 1024, d=300)
    res = decode_both(cp)
    check_equivalent(cp, *res)


def test_cfg3_with_and_without_the_speculative_transition_agree(monkeypatch):
    """The same ragged launch with the chain wave's speculative transition switched off decodes to the same bits."""
    from action_segmentation_amd import ops
    cp = make_corpus(77, [14000, 12000, 5000], 19, 1024)
    monkeypatch.delenv('SMM_SPEC', raising=False)
    a = decode_both(cp)
    monkeypatch.setenv('SMM_SPEC', '0')
    b = decode_both(cp)
    np.testing.assert_array_equal(a[0]['spans'].cpu().numpy(), b[0]['spans'].cpu().numpy())
    np.testing.assert_array_equal(a[0]['best'].cpu().numpy(), b[0]['best'].cpu().numpy())


# ------------------------------------------------------------------------------------------------ log-partition, full size
def _constrained_corpus(seed, lengths, c, k, d=200, rate=(10, 50), all_cyclic=False):
    """BASELINE configs[3] shape: a left-to-right chain (transition mask -1e9 BEFORE the softmax, reference
    semimarkov_modules.py:298-322; only the first state may start, :284-296; only the last may end, :462-471) with
    narration constraints (-1e4 on a step's column outside its window, 0 on background columns, semimarkov.py:149-157)."""
    cp = make_corpus(seed, lengths, c, k, d=d, rate=rate)
    g = np.random.default_rng(seed + 1000)
    # videos that the chain can explain: ONE pass through the states (runs longer than K - 1 frames are covered by
    # self-transitions); the odd videos keep make_corpus' cyclic labels, which the chain can only explain badly
    sigma = np.sqrt(cp['var'])
    for i, t in enumerate(lengths):
        if not all_cyclic and i % 2 == 0 and t >= c:
            cuts = np.sort(g.choice(np.arange(1, t), size=c - 1, replace=False))
            lab = np.repeat(np.arange(c), np.diff(np.concatenate([[0], cuts, [t]])))
            cp['labs'][i] = lab
            cp['xs'][i] = (cp['mu'][lab] + sigma * g.standard_normal((t, d))).astype(np.float32)
    logits = g.standard_normal((c, c))
    allowed = np.eye(c, dtype=bool)
    for f in range(c - 1):
        allowed[f + 1, f] = True                                       # [to, from]
    masked = np.where(allowed, logits, -1e9)
    mx = masked.max(0, keepdims=True)
    cp['trans'] = masked - (mx + np.log(np.exp(masked - mx).sum(0, keepdims=True)))
    il = np.where(np.arange(c) == 0, g.standard_normal(c), -1e9)
    cp['init'] = il - (il.max() + np.log(np.exp(il - il.max()).sum()))
    cp['endpen'] = np.full((len(lengths), c), -1e9)
    cp['endpen'][:, c - 1] = 0.0
    cons = []
    for lab in cp['labs']:
        t = lab.shape[0]
        cn = np.zeros((t, c), np.float32)
        for j in range(1, c, 2):                                       # odd states = steps
            pos = np.flatnonzero(lab == j)
            lo, hi = (0, t) if len(pos) == 0 else (max(0, pos.min() - int(g.integers(0, 20))), min(t, pos.max() + 1 + int(g.integers(0, 20))))
            cn[:lo, j] = -1e4
            cn[hi:, j] = -1e4
        cons.append(cn)
    cp['cons'] = cons
    return cp


def _logz_both(cp, measure=False):
    """(``measure``: return the largest errors instead of asserting the unit tolerances)
    features -> smm_emission_f64 -> smm_logz_f64 + smm_logz_bwd_f64 -> smm_emission_bwd_f64 on the GPU; the C twin's
    emission + exact forward-backward on the host, and the chain rule through the emission scorer in numpy."""
    from action_segmentation_amd import ops
    dev = torch.device('cuda:0')
    lengths = np.array([x.shape[0] for x in cp['xs']], dtype=np.int64)
    b, tmax, c, k, d = len(lengths), int(lengths.max()), cp['c'], cp['k'], cp['d']
    kp = min(k, tmax)
    off = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    x = torch.from_numpy(np.concatenate(cp['xs'], 0)).to(dev)
    mu, var = cp['mu'], cp['var']
    w = (mu / var).T.copy()
    lognorm = -0.5 * np.log(var).sum() - 0.5 * d * np.log(2 * np.pi)
    cst = -0.5 * (mu * mu / var).sum(1) + lognorm
    t = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
    cons = cp.get('cons')
    cons_dev = None if cons is None else torch.from_numpy(np.concatenate(cons, 0)).to(dev).contiguous()
    batch = ops.Batch(lengths, [c], k, c_max=c, frame_offset=off, kp=[kp] * b, d=d, t_max=tmax, total_frames=int(lengths.sum()))
    tabs = (t(cp['trans'][None]), t(cp['init'][None]), t(cp['lens'][None]))
    ep = t(cp.get('endpen'))
    elp64, _ = ops.emission(batch, x, t(w[None]), t(cst[None]), t(1.0 / var), cons=cons_dev)
    ws = torch.empty(batch.workspace_bytes(), dtype=torch.uint8, device=dev)
    up = np.linspace(0.5, 1.5, b)
    z = ops.logz(batch, elp64, *tabs, endpen=ep, ws=ws, with_backward=True)
    gr = ops.logz_bwd(batch, elp64, *tabs, z, grad_logz=t(up), endpen=ep, ws=ws, with_backward=True)
    g_w, g_cst, g_iv = ops.emission_bwd(batch, x, gr['elp'], ws=ws)
    torch.cuda.synchronize()
    # the twin
    xp = np.zeros((b, tmax, d), np.float32)
    cn = None if cons is None else np.zeros((b, tmax, c))
    for i, xi in enumerate(cp['xs']):
        xp[i, :xi.shape[0]] = xi
        if cons is not None:
            cn[i, :xi.shape[0]] = cons[i]
    elp = F.emission(xp, lengths, mu, 1.0 / var, lognorm, cn)
    z_ref, g_ref = F.logz(elp, lengths, cp['trans'], cp['init'], cp['lens'], cp.get('endpen'), grad=True, upstream=up)
    ge = gr['elp'].cpu().numpy()
    if measure:
        # the bench line's figures (bench.py: logz_cpu_baseline): max |got - ref| / max(1, |ref|) over every gradient entry
        rel = lambda got, ref: float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
        errs = {'logz': float(np.max(np.abs(z.cpu().numpy() - z_ref) / np.abs(z_ref))),
                'elp': max(rel(ge[off[i]:off[i] + ti], g_ref['elp'][i, :ti]) for i, ti in enumerate(lengths)),
                'trans': rel(gr['trans'].cpu().numpy()[0], g_ref['trans']), 'init': rel(gr['init'].cpu().numpy()[0], g_ref['init']),
                'len': rel(gr['len'].cpu().numpy()[0, :kp], g_ref['len']),
                'count_max': float(max(np.abs(g_ref['trans']).max(), np.abs(g_ref['len']).max()))}
        return errs
    np.testing.assert_allclose(z.cpu().numpy(), z_ref, rtol=1e-6)
    gw_ref, gc_ref, giv_ref = np.zeros((d, c)), np.zeros(c), np.zeros(d)
    for i, ti in enumerate(lengths):
        gi = ge[off[i]:off[i] + ti]
        np.testing.assert_allclose(gi, g_ref['elp'][i, :ti], rtol=2e-5, atol=2e-5, err_msg='video %d' % i)
        np.testing.assert_allclose(gi.sum(1), up[i], rtol=1e-4)        # posteriors: exactly one state per frame
        xd = cp['xs'][i].astype(np.float64)
        gw_ref += xd.T @ g_ref['elp'][i, :ti]
        gc_ref += g_ref['elp'][i, :ti].sum(0)
        giv_ref += -0.5 * ((xd * xd) * g_ref['elp'][i, :ti].sum(1, keepdims=True)).sum(0)
    np.testing.assert_allclose(gr['trans'].cpu().numpy()[0], g_ref['trans'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gr['init'].cpu().numpy()[0], g_ref['init'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gr['len'].cpu().numpy()[0, :kp], g_ref['len'], rtol=2e-5, atol=2e-5)
    # chain rule through elp = cst + x.w - 0.5 x^2.inv_var: sums over ~T frames of posteriors x features
    scale = max(1.0, float(np.abs(gw_ref).max()))
    np.testing.assert_allclose(g_w.cpu().numpy()[0], gw_ref, rtol=2e-5, atol=2e-5 * scale)
    np.testing.assert_allclose(g_cst.cpu().numpy()[0], gc_ref, rtol=2e-5, atol=2e-5 * scale)
    np.testing.assert_allclose(g_iv.cpu().numpy(), giv_ref, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(giv_ref).max())))
    return z_ref


@pytest.mark.parametrize('c', [7, 13])
def test_cfg4_log_partition_and_gradients_with_constraints(c):
    """BASELINE configs[3] at its own shape: T <= 2048, K = 64, 7..13 states, D = 200, chain-masked transitions at -1e9,
    narration penalties -1e4, end penalties: the fp32 integer-exponent ring of smm_logz_kernel at real magnitudes
    (emission ~ -290 per frame, logZ ~ -6e5)."""
    z = _logz_both(_constrained_corpus(40 + c, [2048, 900, 1500, 200, 640], c, 64))
    assert np.all(z < -1e4) and np.all(np.isfinite(z))


@pytest.mark.parametrize('c', [21, 23])
def test_cfg3_shape_log_partition_and_gradients(c):
    """K = 1024, T = 14 000 (cumE reaches 4e6), 21 and 23 states, next to short videos of the same task."""
    _logz_both(make_corpus(50 + c, [14000, 600, 3000], c, 1024))


def test_cfg3_whole_corpus_one_ragged_launch_equals_the_twin():
    """BASELINE configs[2] as bench.py decodes it: the cfg3 seed-2 corpus (18 tasks x 20 videos, 11..23 states, T up to
    14 000, K = 1024) in ONE ragged launch against the C twin, every frame of every video."""
    import bench
    from action_segmentation_amd import synth
    from action_segmentation_amd.semimarkov import SemiMarkovModel
    dev = torch.device('cuda:0')
    cfg = synth.CONFIGS['cfg3']
    data = synth.SynthDatasplit('cfg3', seed=2, device=dev)
    fitted = SemiMarkovModel.from_args(synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size']), data)
    fitted.fit(data.subset(6), use_labels=True)
    model = SemiMarkovModel.from_args(synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size']), data)
    model.model.load_state_dict(fitted.model.state_dict(), strict=False)
    model.model.to(dev)
    pc = model.prepare(data)
    out = model.model.decode_packed(pc, want_spans=False, want_labels=True)
    torch.cuda.synchronize()
    from action_segmentation_amd import ops
    ops.check_decoded(pc.batch, out)
    labels = out['labels'].cpu().numpy()
    _, par = bench.cpu_factored(pc, model, gpu_labels=labels, budget_s=1e9)
    assert par['videos_checked'] == 360 and par['frames_checked'] == pc.n_frames
    assert par['label_mismatches'] == 0


@pytest.mark.parametrize('k', [1024, 64])
def test_split_decode_equals_single_stream_decode_and_the_twin(k, monkeypatch):
    """smm_decode_f32 splits a launch whose few longest videos set the DP's time: those are scored and decoded on the
    caller's stream, the rest on a second stream beside them (smm_api.hip: choose_split).  Same labels, spans and scores
    as the single-stream decode and as the twin."""
    monkeypatch.setenv('SMM_SPLIT_MIN_US', '0')
    cp = make_corpus(91, [5000, 4800, 4700] + [1200 + 13 * i for i in range(40)], 9, k, rate=(20, 200) if k > 64 else (5, 40))
    res = decode_both(cp)
    check_equivalent(cp, *res)
    monkeypatch.setenv('SMM_NO_SPLIT', '1')
    one = decode_both(cp)
    for key in ('spans', 'labels', 'best', 'n_segs'):
        np.testing.assert_array_equal(res[0][key].cpu().numpy(), one[0][key].cpu().numpy())


def test_cfg5_sharded_decode_at_cfg3_size_through_rccl():
    """BASELINE configs[4] at size on one GPU (the sharded tests elsewhere use the `tiny` corpus): the cfg3 seed-2 corpus
    decoded as shard (r, 4), r = 0..3 -- every shard against the C twin, the union against the unsharded decode, and the
    four shards' evaluation counters summed through a real one-rank RCCL group before they are finalised
    (tests/rccl_shard_fullsize.py; reference src/data/corpus.py:405-604 summed as src/main.py:486-532)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMM_DIST_SINGLE_RANK='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'rccl_shard_fullsize.py')], env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1])
    assert line['backend'] == 'nccl' and line['shards'] == 4 and line['videos'] == 360
    assert line['frames'] == sum(line['frames_per_shard']) and 0.0 < line['mof'] <= 1.0


@pytest.mark.parametrize('t_len', [512, 2048, 8192])
@pytest.mark.parametrize('k', [64, 256, 1024])
def test_logz_gradient_error_does_not_grow_with_the_lattice(t_len, k):
    """The log-partition kernel keeps a ring slot's running sum as an fp32 mantissa with an integer-valued fp32 exponent
    (smm_logz.hip); a candidate's exponent is rounded to fp32 and a slot sums up to K terms.  bench.py's cfg4 corpus --
    chain-masked transitions, narration penalties, CYCLIC label sequences the chain can only explain through penalties --
    sits at grad_max_rel 3.4e-5 with T <= 2048, K = 64.  This test bounds the same figure (max |got - ref| / max(1, |ref|)
    over every gradient entry, against the C twin's exact forward-backward) as T and K grow: <= 5e-5 at T in {512, 2048,
    8192} x K in {64, 256, 1024}, inside the path's 1e-4 (north_star: forward log-marginals within 1e-4 relative)."""
    if k > t_len:
        pytest.skip('K is clipped to the video (reference semimarkov_modules.py:450-452): same lattice as K = T')
    c = 9
    cp = _constrained_corpus(60 + t_len // 512 + k // 64, [t_len, t_len // 2, t_len, max(c + 1, t_len // 3)], c, k,
                             rate=(10, 50) if k <= 64 else (20, min(400, k // 2)), all_cyclic=True)
    errs = _logz_both(cp, measure=True)
    import json
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    with open(os.path.join('gpurun_out', 'logz_error_growth.jsonl'), 'a') as f:
        f.write(json.dumps(dict(T=t_len, K=k, states=c, **errs)) + '\n')
    assert errs['logz'] <= 1e-6, errs
    assert max(errs[key] for key in ('elp', 'trans', 'init', 'len')) <= 5e-5, errs
