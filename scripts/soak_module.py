"""Randomised soak of SemiMarkovModule's training step against the dense reference path (round 5): random modules (3..24 classes,
D in {4, 17, 64, 200}, max span 3..32), random valid-class subsets, batches of 1..5 ragged videos, narration constraints on and
off; mean log Z through the HIP kernels against oracle/dense_ref.py in fp64 (log_hsmm potentials + log-semiring DP), and its
gradients with respect to the four parameter tensors -- i.e. smm_logz_f64, smm_logz_bwd_f64, smm_emission_bwd_f64 and the factor
tables' backward in one chain -- against autograd through the dense path.  Bars: log Z 1e-6 relative + 1e-4; gradients rtol 5e-4 of
the tensor's largest entry (the parameters are fp32 tensors; the unit test's bar).  Each batch is also DECODED (viterbi()) and compared
with the dense path's max-semiring DP: same frame labels and EOS placement, the path re-scored to the dense optimum.
usage: soak_module.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from module_util import make_args
from golden_util import assert_spans_equivalent
from oracle import dense_ref as O
from action_segmentation_amd.semimarkov_modules import SemiMarkovModule

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = np.random.default_rng(seed)
torch.manual_seed(seed)
dev = torch.device('cuda:0')
torch.set_num_threads(8)
names = ['poisson_log_rates', 'gaussian_means', 'transition_logits', 'init_logits']
t0, n, nv, worst_z, worst_g, last = time.time(), 0, 0, 0.0, 0.0, time.time()
while time.time() - t0 < budget:
    nc = int(g.integers(3, 25)); d = int(g.choice([4, 17, 64, 200])); K = int(g.integers(3, 33))
    b = int(g.integers(1, 6)); tmax = int(g.choice([12, 40, 90]))
    m = SemiMarkovModule(make_args(K), nc, d, allow_self_transitions=True).to(dev)
    with torch.no_grad():
        m.poisson_log_rates.copy_(torch.log(torch.tensor(g.uniform(1.5, K * 0.8, nc), dtype=torch.float32)))
        m.gaussian_means.copy_(torch.tensor(g.standard_normal((nc, d)) * 0.7, dtype=torch.float32))
        m.gaussian_cov.copy_(torch.diag(torch.tensor(0.5 + g.random(d), dtype=torch.float32)))
        m.transition_logits.copy_(torch.tensor(g.standard_normal((nc, nc)), dtype=torch.float32))
        m.init_logits.copy_(torch.tensor(g.standard_normal(nc), dtype=torch.float32))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    p = O.RefParams(nc, sd['poisson_log_rates'], sd['gaussian_means'], torch.diagonal(sd['gaussian_cov']).clone(),
                    sd['transition_logits'], sd['init_logits'], K, True).to(torch.float64)
    c = int(g.integers(2, nc + 1))
    valid = torch.tensor(np.sort(g.choice(nc, size=c, replace=False)), dtype=torch.long) if g.random() < 0.7 else None
    cv = c if valid is not None else nc
    lengths = torch.tensor(g.integers(max(2, tmax // 3), tmax + 1, size=b)); lengths[0] = tmax
    feats = torch.tensor(g.standard_normal((b, tmax, d)), dtype=torch.float32)
    cons = torch.tensor((g.random((b, tmax, cv)) < 0.15) * -3.0, dtype=torch.float32) if g.random() < 0.4 else None
    vc = None if valid is None else [valid for _ in range(b)]
    m.zero_grad()
    ll, _ = m.log_likelihood(feats.to(dev), lengths.to(dev), vc, spans=None, constraints=None if cons is None else cons.to(dev))
    ll.backward()
    leaves = {k: getattr(p, k).clone().requires_grad_(True) for k in names}
    for k, v in leaves.items():
        setattr(p, k, v)
    scores, _ = O.score_features(p, feats.double(), lengths, valid, True, None, None if cons is None else cons.double())
    z, _ = O.semimarkov_dp(scores, lengths + 1, O.LogSemiring)
    z.mean().backward()
    ez = abs(ll.item() - z.mean().item()) / (1e-4 + 1e-6 * abs(z.mean().item()))
    worst_z = max(worst_z, ez)
    assert ez <= 1.0, (nc, d, K, b, tmax, 'logz', ll.item(), z.mean().item())
    for k in names:
        got = getattr(m, k).grad.detach().cpu().double().numpy()
        ref = leaves[k].grad.numpy()
        e = float(np.abs(got - ref).max() / (5e-4 * max(1.0, np.abs(ref).max()) ))
        worst_g = max(worst_g, e)
        assert e <= 1.0 + 0.0, (nc, d, K, b, tmax, k, e)
    # the decode of the same batch: frame labels and EOS placement equal to the dense path's, boundaries inside one-class runs
    # certified by re-scoring under the dense potentials (tests/golden_util.py)
    with torch.no_grad():
        spans = m.viterbi(feats.to(dev), lengths.to(dev), vc, add_eos=True, constraints=None if cons is None else cons.to(dev))
    q = O.RefParams(nc, sd['poisson_log_rates'], sd['gaussian_means'], torch.diagonal(sd['gaussian_cov']).clone(),
                    sd['transition_logits'], sd['init_logits'], K, True).to(torch.float64)
    r = O.viterbi_full(q, feats.double(), lengths, valid, True, None, None if cons is None else cons.double())
    assert_spans_equivalent(spans.numpy(), r['spans'].numpy(), lengths, nc)
    local = O.map_spans_to_local(spans, valid, nc)
    np.testing.assert_allclose(O.rescore(r['scores'], local, r['pos_lengths']).numpy(), r['v'].numpy(), rtol=1e-9, atol=1e-7)
    nv += b
    n += 1
    if time.time() - last > 20:
        last = time.time()
        print('  ... %d training steps, worst log Z error %.3f of its bar, worst gradient error %.3f of its bar' % (n, worst_z, worst_g), flush=True)
print('soak ok: %d training steps and decodes of %d videos (random modules, valid-class subsets, constraints), %.0f s; decodes equal to the dense path\'s (labels, EOS, re-scored optimum); worst log Z error %.3f of its bar (1e-4 + 1e-6 |log Z|), '
      'worst gradient error %.3f of its bar (5e-4 of the tensor\'s largest entry)' % (n, nv, time.time() - t0, worst_z, worst_g))
