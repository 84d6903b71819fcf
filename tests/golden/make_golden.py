"""Generate golden vectors by IMPORTING the reference in the build container.

Run once here (``python tests/golden/make_golden.py``); the resulting ``*.npz`` /
``*.json`` files are committed.  The GPU box has no ``/root/reference`` -- tests only
read the fixtures.  What each value is:

* ``elp/init/trans/len/scores`` (fp32 and fp64): outputs of the REFERENCE's own code
  (``SemiMarkovModule.emission_log_probs / initial_log_probs / transition_log_probs /
  length_log_probs / score_features -> log_hsmm``, semimarkov_modules.py:284-595).
* ``fit_*``: the reference's ``fit_supervised`` (modules:195-256, utils:74-126).
* ``kat_scores``: ``log_hsmm`` on the inputs of the reference's known-answer test
  (src/models/test_semimarkov.py:266-323).
* ``ref_spans*``: the reference's ``SemiMarkovModule.viterbi`` host code (class
  un-mapping, EOS conventions, modules:660-696) run with ``torch_struct`` replaced by
  the oracle's restated DP -- labelled as such: they pin the host conventions, not
  torch_struct's numerics.
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference/src')

from oracle import dense_ref as O  # noqa: E402


# ---- torch_struct stand-in built from the oracle's restated DP (only for ref_spans*) ----
class _Struct:
    def __init__(self, semiring=None):
        self.semiring = semiring or O.LogSemiring

    def marginals(self, edge, lengths=None):
        return O.marginals(edge, lengths, self.semiring)[1]

    @staticmethod
    def from_parts(parts):
        return O.from_parts(parts), (parts.shape[-1], parts.shape[2])

    @staticmethod
    def to_parts(seq, extra, lengths=None):
        return O.to_parts(seq, extra[0], extra[1], lengths)

    def score(self, potentials, parts, batch_dims=(0,)):
        return (potentials * parts).flatten(1).sum(-1)


class _CRF:
    struct = _Struct

    def __init__(self, log_potentials, lengths=None):
        self.log_potentials = log_potentials
        self.lengths = lengths
        self.event_shape = log_potentials.shape[1:]

    @property
    def argmax(self):
        return O.marginals(self.log_potentials, self.lengths, O.MaxSemiring)[1]

    @property
    def partition(self):
        return O.semimarkov_dp(self.log_potentials, self.lengths, O.LogSemiring)[0]


ts = types.ModuleType('torch_struct')
ts.SemiMarkovCRF, ts.SemiMarkov, ts.MaxSemiring = _CRF, _Struct, O.MaxSemiring
sys.modules['torch_struct'] = ts
ed = types.ModuleType('editdistance')
ed.eval = lambda a, b: 0
sys.modules['editdistance'] = ed

from models.semimarkov.semimarkov_modules import SemiMarkovModule  # noqa: E402
from models.semimarkov import semimarkov_utils as ref_utils  # noqa: E402


def make_args(max_k):
    p = argparse.ArgumentParser()
    SemiMarkovModule.add_args(p)
    a = p.parse_args([])
    a.sm_train_discriminatively = False
    a.sm_max_span_length = max_k
    return a


CASES = {
    # name: dict(b, T list, n_classes, valid (None or list), K, D, constraints cfg)
    'tiny': dict(T=[12, 9, 12], n_classes=3, valid=None, K=4, D=5),
    'subset_merge': dict(T=[40, 31], n_classes=9, valid=[1, 2, 4, 5, 7, 8], K=8, D=16,
                         merge={0: 0, 1: 1, 2: 2, 3: 3, 4: 1, 5: 5, 6: 6, 7: 1, 8: 8}),
    'k_gt_t': dict(T=[5, 4], n_classes=3, valid=None, K=8, D=4),
    'hmm_k1': dict(T=[7, 6], n_classes=3, valid=None, K=1, D=4),
    'constrained': dict(T=[20, 14, 17], n_classes=7, valid=[0, 1, 2, 3, 4], K=6, D=6,
                        starts={0, 5}, transitions={0: {1}, 1: {2}, 2: {3}, 3: {4}, 5: {6}}, ends={4, 6},
                        additional=[[], [2], [3]], narration=True),
    'no_eos': dict(T=[10, 10], n_classes=3, valid=None, K=4, D=3, add_eos=False),
}


def build(case, seed):
    g = torch.Generator().manual_seed(seed)
    cfg = CASES[case]
    n_classes, d, k = cfg['n_classes'], cfg['D'], cfg['K']
    b, tmax = len(cfg['T']), max(cfg['T'])
    kwargs = {}
    if 'starts' in cfg:
        trans = {s: set(t) | {s} for s, t in cfg['transitions'].items()}
        for s in range(n_classes):
            trans.setdefault(s, set()).add(s)
        kwargs = dict(allowed_starts=cfg['starts'], allowed_transitions=trans, allowed_ends=cfg['ends'])
    m = SemiMarkovModule(make_args(k), n_classes, d, allow_self_transitions=True,
                         merge_classes=cfg.get('merge'), **kwargs)
    with torch.no_grad():
        m.poisson_log_rates.copy_(torch.rand(n_classes, generator=g) * 1.5)
        m.gaussian_means.copy_(torch.randn(n_classes, d, generator=g) * 0.7)
        m.gaussian_cov.copy_(torch.diag(0.5 + torch.rand(d, generator=g)))
        m.transition_logits.copy_(torch.randn(n_classes, n_classes, generator=g))
        m.init_logits.copy_(torch.rand(n_classes, generator=g))
    feats = torch.randn(b, tmax, d, generator=g)
    lengths = torch.tensor(cfg['T'])
    for i, t in enumerate(cfg['T']):
        feats[i, t:] = 0  # padding_colate zero-pads (model.py:59-61)
    valid = None if cfg['valid'] is None else torch.tensor(cfg['valid'])
    c1 = n_classes if valid is None else len(valid)
    cons = None
    if cfg.get('narration'):
        cons = torch.zeros(b, tmax, c1)
        for i in range(b):
            for c in range(1, c1, 2):  # "step" columns get a window, background columns stay 0
                lo = int(torch.randint(0, tmax - 3, (1,), generator=g))
                allowed = torch.zeros(tmax)
                allowed[lo:lo + 6] = 1
                cons[i, :, c] = (1 - allowed) * -1e4
    return m, feats, lengths, valid, cons, cfg


def run_case(case, seed, out):
    m, feats, lengths, valid, cons, cfg = build(case, seed)
    add_eos = cfg.get('add_eos', True)
    addl = cfg.get('additional')
    pre = case + '/'
    out[pre + 'features'] = feats.numpy()
    out[pre + 'lengths'] = lengths.numpy()
    if valid is not None:
        out[pre + 'valid_classes'] = valid.numpy()
    if cons is not None:
        out[pre + 'constraints'] = cons.numpy()
    for name, prm in m.state_dict().items():
        if prm is not None:
            out[pre + 'param/' + name] = prm.numpy()
    for dt, tag in ((torch.float32, 'f32'), (torch.float64, 'f64')):
        torch.set_default_dtype(dt)
        mm = m.double() if dt == torch.float64 else m.float()
        f = feats.to(dt)
        c = None if cons is None else cons.to(dt)
        with torch.no_grad():
            scores, _, elp = mm.score_features(f, lengths, valid, add_eos=add_eos, use_mean_z=True,
                                               additional_allowed_ends_per_instance=addl, constraints=c,
                                               return_elp=True)
            out[pre + tag + '/elp'] = elp.numpy()
            out[pre + tag + '/init'] = mm.initial_log_probs(valid).numpy()
            out[pre + tag + '/trans'] = mm.transition_log_probs(valid).numpy()
            out[pre + tag + '/len'] = mm.length_log_probs(valid).numpy()
            out[pre + tag + '/scores'] = scores.numpy()
            if add_eos:
                vc = None if valid is None else [valid for _ in range(feats.shape[0])]
                spans = mm.viterbi(f, lengths, vc, add_eos=True, additional_allowed_ends_per_instance=addl,
                                   constraints=c)
                out[pre + tag + '/ref_spans'] = spans.numpy()
                ll, _ = mm.log_likelihood(f, lengths, vc, spans=None, add_eos=True,
                                          additional_allowed_ends_per_instance=addl, constraints=c)
                out[pre + tag + '/ref_mean_logz'] = np.array(ll.item())
        torch.set_default_dtype(torch.float32)
        m.float()


def kat(out):
    """Inputs of test_semimarkov.py:266-323 through the reference's log_hsmm."""
    b, c, n, k, step = 10, 4, 100, 5, 4
    padded = n + step * 2
    lengths = torch.full((b,), n).long()
    lengths[0] = padded
    trans = torch.zeros(c, c)
    init = torch.full((c,), -1e9)
    init[0] = 0
    em = torch.full((b, padded, c), -1e9)
    for t in range(padded):
        em[:, t, (t // step) % c] = 1
    ls = torch.full((k, c), -1e9)
    ls[step, :] = 0
    scores = SemiMarkovModule.log_hsmm(trans, em, init, ls, lengths, add_eos=True)
    out['kat/scores'] = scores.numpy()
    out['kat/lengths'] = lengths.numpy()


def fit_case(out):
    g = torch.Generator().manual_seed(7)
    n_classes, d, k = 4, 3, 6
    feats, labels = [], []
    for t in (9, 14, 7, 11, 16):
        lab, cur = [], int(torch.randint(0, n_classes, (1,), generator=g))
        while len(lab) < t:
            lab += [cur] * int(torch.randint(1, 8, (1,), generator=g))
            cur = int(torch.randint(0, n_classes, (1,), generator=g))
        lab = torch.tensor(lab[:t])
        feats.append(torch.randn(t, d, generator=g) + lab[:, None].float())
        labels.append(lab)
    m = SemiMarkovModule(make_args(k), n_classes, d, allow_self_transitions=True)
    m.fit_supervised(feats, labels)
    for i, (f, l) in enumerate(zip(feats, labels)):
        out['fit/features%d' % i] = f.numpy()
        out['fit/labels%d' % i] = l.numpy()
    for name, prm in m.state_dict().items():
        if prm is not None:
            out['fit/param/' + name] = prm.numpy()
    out['fit/max_k'] = np.array(k)
    out['fit/n_classes'] = np.array(n_classes)


def codecs():
    labels = torch.LongTensor([[0, 1, 1, 2, 2, 2], [0, 1, 2, 3, 3, 4]])
    spans = ref_utils.labels_to_spans(labels, max_k=10)
    rnd = torch.randint(0, 3, (5, 20), generator=torch.Generator().manual_seed(3))
    rnd_spans = ref_utils.labels_to_spans(rnd, max_k=5)
    return dict(
        labels=labels.tolist(), spans=spans.tolist(), max_k=10,
        back=ref_utils.spans_to_labels(spans).tolist(),
        rle=ref_utils.rle_spans(spans, torch.LongTensor([6, 6])),
        rle_trunc=ref_utils.rle_spans(spans, torch.LongTensor([5, 6])), trunc_lengths=[5, 6],
        rand_labels=rnd.tolist(), rand_spans=rnd_spans.tolist(), rand_max_k=5,
        rand_back=ref_utils.spans_to_labels(rnd_spans).tolist(),
    )


if __name__ == '__main__':
    out = {}
    for i, case in enumerate(CASES):
        run_case(case, 100 + i, out)
    kat(out)
    fit_case(out)
    np.savez_compressed(os.path.join(HERE, 'reference_vectors.npz'), **out)
    with open(os.path.join(HERE, 'codec_vectors.json'), 'w') as f:
        json.dump(codecs(), f)
    print('wrote', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'reference_vectors.npz')), 'bytes')
