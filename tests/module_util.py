"""Build the product SemiMarkovModule from a golden-fixture case."""
import argparse

import torch

from golden_util import CASES


def make_args(max_k, **kw):
    from action_segmentation_amd.semimarkov_modules import SemiMarkovModule
    p = argparse.ArgumentParser()
    SemiMarkovModule.add_args(p)
    a = p.parse_args([])
    a.sm_max_span_length = max_k
    a.sm_train_discriminatively = False
    for k, v in kw.items():
        setattr(a, k, v)
    return a


CONSTRAINED = dict(starts={0, 5}, transitions={0: {1}, 1: {2}, 2: {3}, 3: {4}, 5: {6}}, ends={4, 6})


def module_from_golden(g, case):
    from action_segmentation_amd.semimarkov_modules import SemiMarkovModule
    cfg = CASES[case]
    pre = case + '/param/'
    n_classes = g[pre + 'init_logits'].shape[0]
    d = g[pre + 'gaussian_means'].shape[1]
    kw = {}
    if case == 'constrained':
        trans = {s: set(t) | {s} for s, t in CONSTRAINED['transitions'].items()}
        for s in range(n_classes):
            trans.setdefault(s, set()).add(s)
        kw = dict(allowed_starts=CONSTRAINED['starts'], allowed_transitions=trans, allowed_ends=CONSTRAINED['ends'])
    m = SemiMarkovModule(make_args(cfg['K']), n_classes, d, allow_self_transitions=True,
                         merge_classes=cfg.get('merge'), **kw)
    sd = {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    return m
