"""Closed-form supervised fit on the device (csrc/smm_fit.hip) against the CPU restatement (oracle/dense_ref.py), which
is itself pinned to the reference's fit_supervised outputs (tests/golden, test_oracle_golden.py).
Integer statistics: exact.  Floating-point sums: fp64 with an unspecified addition order -> rel 1e-12."""
import numpy as np
import pytest
import torch

from oracle import dense_ref as O

pytestmark = pytest.mark.gpu


def make_videos(rng, n_videos, t_lo, t_hi, n_classes, d, mean_len, classes=None):
    feats, labels = [], []
    classes = list(range(n_classes)) if classes is None else classes
    for _ in range(n_videos):
        t = int(rng.integers(t_lo, t_hi + 1))
        y = []
        while len(y) < t:
            y += [int(rng.choice(classes))] * int(max(1, rng.poisson(mean_len)))
        y = np.asarray(y[:t], dtype=np.int64)
        x = (rng.normal(size=(t, d)) + y[:, None] * 0.1).astype(np.float32)
        feats.append(x)
        labels.append(y)
    return feats, labels


@pytest.mark.parametrize('d,n_classes,max_k,t_hi', [(200, 9, 20, 700), (200, 9, 4, 300), (7, 5, 3, 90), (16, 40, 64, 2500),
                                                     (300, 6, None, 400), (1, 3, 2, 50)])
def test_sufficient_stats_match_oracle(d, n_classes, max_k, t_hi):
    from action_segmentation_amd.semimarkov_utils import semimarkov_sufficient_stats_device
    rng = np.random.default_rng(d * 31 + n_classes)
    feats, labels = make_videos(rng, 7, 1, t_hi, n_classes, d, mean_len=11,
                                classes=[c for c in range(n_classes) if c != 2])     # class 2 never occurs
    em, st = semimarkov_sufficient_stats_device([torch.from_numpy(f) for f in feats], [torch.from_numpy(l) for l in labels],
                                                'tied_diag', n_classes, max_k)
    want = O.sufficient_stats(feats, labels, n_classes, max_k)
    for key in ('span_counts', 'span_lengths', 'span_start_counts', 'span_transition_counts'):
        np.testing.assert_array_equal(st[key], want[key], err_msg=key)
    assert st['instance_count'] == want['instance_count']
    np.testing.assert_allclose(em.means_, want['means'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(em.covariances_[0], want['var'], rtol=1e-11, atol=0)
    assert np.all(em.means_[2] == 0)


def test_fit_supervised_on_device_matches_host_module_and_oracle():
    from action_segmentation_amd import synth
    from action_segmentation_amd.semimarkov import SemiMarkovModel
    data = synth.SynthDatasplit('tiny', seed=2)
    host = SemiMarkovModel.from_args(synth.make_args(data.max_k, cuda=False, batch_size=2), data)
    host.fit(data, use_labels=True)
    dev = SemiMarkovModel.from_args(synth.make_args(data.max_k, cuda=True, batch_size=2), data)
    dev.fit(data, use_labels=True)
    feats = [smp['features'].numpy() for smp in data._videos.values()]
    labels = [smp['gt_single'].numpy() for smp in data._videos.values()]
    want = O.fit_supervised(feats, labels, data.corpus.n_classes, data.max_k)
    sd_h, sd_d = host.model.state_dict(), dev.model.state_dict()
    for name in ('poisson_log_rates', 'gaussian_means', 'gaussian_cov', 'transition_logits', 'init_logits'):
        assert sd_d[name].is_cuda
        np.testing.assert_allclose(sd_d[name].cpu().numpy(), sd_h[name].numpy(), rtol=2e-6, atol=1e-7, err_msg=name)
    np.testing.assert_allclose(sd_d['gaussian_means'].cpu().numpy(), want['gaussian_means'], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(sd_d['poisson_log_rates'].cpu().numpy(), want['poisson_log_rates'], rtol=2e-6, atol=1e-7)


def test_out_of_range_label_is_an_error():
    from action_segmentation_amd.semimarkov_utils import semimarkov_sufficient_stats_device
    x = [torch.zeros(5, 8)]
    y = [torch.tensor([0, 1, 9, 1, 0])]
    with pytest.raises(ValueError, match="outside"):
        semimarkov_sufficient_stats_device(x, y, 'tied_diag', 4, 5)
