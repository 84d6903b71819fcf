"""GPU parity of SemiMarkovModel.predict (fused ragged launch and the reference's per-batch call pattern)."""
import numpy as np
import pytest
import torch

from oracle import dense_ref as O
from action_segmentation_amd import synth
from action_segmentation_amd.batching import make_data_loader
from action_segmentation_amd.semimarkov import SemiMarkovModel

pytestmark = pytest.mark.gpu


def build(constrain, narration, seed=3):
    data = synth.SynthDatasplit('tiny', seed=seed)
    fit_args = synth.make_args(data.max_k, cuda=False, batch_size=2)
    fitted = SemiMarkovModel.from_args(fit_args, data)
    fitted.fit(data, use_labels=True)                       # closed form needs an unconstrained model (reference)
    args = synth.make_args(data.max_k, cuda=True, batch_size=2, sm_constrain_transitions=constrain,
                           sm_constrain_with_narration=list(narration))
    model = SemiMarkovModel.from_args(args, data)
    model.model.load_state_dict(fitted.model.state_dict(), strict=False)
    model.model.cuda()
    return data, args, model


def oracle_predictions(data, args, model):
    """The reference path, batch by batch, in fp64: dense potentials + restated pytorch-struct DP."""
    m = model.model
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    p = O.RefParams(m.n_classes, sd['poisson_log_rates'], sd['gaussian_means'], torch.diagonal(sd['gaussian_cov']).clone(),
                    sd['transition_logits'], sd['init_logits'], m.max_k, True, sd.get('init_constraints'),
                    sd.get('transition_constraints'), m.allowed_ends, m.merge_classes).to(torch.float64)
    preds, certs = {}, []
    cons_fn = model._test_constraints(data)
    for b in make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=args.batch_size):
        cons = cons_fn(b) if cons_fn else None
        addl = model.make_additional_allowed_ends(b['task_name'], b['lengths'])
        r = O.viterbi_full(p, b['features'].double(), b['lengths'], b['task_indices'][0], True, addl,
                           None if cons is None else cons.cpu().double())
        lab = O.spans_to_labels(r['spans'].numpy())
        for i, (name, t) in enumerate(zip(b['video_name'], b['lengths'].tolist())):
            preds[name] = lab[i, :t]
    return preds


@pytest.mark.parametrize('constrain,narration', [(False, ()), (True, ()), (True, ('test',))])
def test_predict_matches_reference_path(constrain, narration):
    data, args, model = build(constrain, narration)
    fused = model.predict(data)
    unfused = model.predict(data, fused=False)
    ref = oracle_predictions(data, args, model)
    assert set(fused) == set(unfused) == set(ref) == {n for (_, n) in data._videos}
    for name in ref:
        assert fused[name].dtype == np.int64
        np.testing.assert_array_equal(fused[name], ref[name], err_msg=name)
        np.testing.assert_array_equal(unfused[name], ref[name], err_msg=name)
    acc = np.mean([np.mean(fused[n] == data._videos[(t, n)]['gt_single'].numpy()) for (t, n) in data._videos])
    assert acc > 0.5                                          # it also segments the synthetic videos sensibly


@pytest.mark.parametrize('depth', [1, 3, 8])
@pytest.mark.parametrize('constrain,narration', [(False, ()), (True, ('test',))])
def test_predict_per_batch_on_streams_equals_fused(depth, constrain, narration):
    """predict(fused=False) keeps up to DECODE_DEPTH batches in flight, each on its own stream and pinned result slot once
    its videos are long (STREAM_MIN_FRAMES): forced on here for the tiny corpus, twice in a row (slots and streams reused)."""
    data, args, model = build(constrain, narration)
    args.decode_depth = depth
    model.STREAM_MIN_FRAMES = 0
    ref = model.predict(data)
    for _ in range(2):
        got = model.predict(data, fused=False)
        assert set(got) == set(ref)
        for name in ref:
            np.testing.assert_array_equal(got[name], ref[name], err_msg=name)
    streams = model._decode_streams[(str(model.device), depth)]
    assert len(streams) == depth and all((s is not None) == (depth > 1) for s in streams)


def test_predict_results_stay_valid_while_referenced_and_their_buffers_are_reused_afterwards():
    """predict() hands out views of a pinned label buffer on lease (ops.lease_host_labels): a result the caller still holds
    must survive later decodes -- also when more results are alive than there are leases (the copy path) -- and a
    dropped result's buffer must be the one the next call decodes into."""
    import gc
    from action_segmentation_amd import ops
    data, args, model = build(False, ())
    other = data.subset(1)
    first = model.predict(data)
    keep = {k: v.copy() for k, v in first.items()}
    held = [model.predict(other if i % 2 == 0 else data) for i in range(ops.LABEL_LEASES + 2)]   # more than the pool leases
    for k in keep:
        np.testing.assert_array_equal(first[k], keep[k], err_msg=k)
    for i, r in enumerate(held):
        if i % 2 == 1:
            for k in keep:
                np.testing.assert_array_equal(r[k], keep[k], err_msg='%d %s' % (i, k))
    pool = ops._label_leases[torch.cuda.current_device()]
    assert len(pool) == ops.LABEL_LEASES and all(ops._storage_users(e[0]) > e[1] for e in pool)
    del first, held, r
    gc.collect()
    assert all(ops._storage_users(e[0]) == e[1] for e in pool)   # every lease came back with its last reference
    again = model.predict(data)
    some = next(iter(again.values()))
    base = some
    while getattr(base, 'base', None) is not None and isinstance(base.base, np.ndarray):
        base = base.base
    assert any(base.ctypes.data == e[0].data_ptr() for e in pool)   # ... and the next result lives in one of them
    for k in keep:
        np.testing.assert_array_equal(again[k], keep[k], err_msg=k)


def test_unsupervised_fit_improves_marginal_likelihood():
    """A few epochs of the reference's unsupervised objective (-log Z, Adam) through the HIP forward/backward kernels."""
    data = synth.SynthDatasplit('tiny', seed=9)
    args = synth.make_args(data.max_k, cuda=True, batch_size=2, epochs=6, lr=5e-2, print_every=0)
    model = SemiMarkovModel.from_args(args, data)
    log = []
    model.fit(data, use_labels=False, callback_fn=lambda ep, st: log.append(st['train_loss']))
    assert len(log) == 6 and all(np.isfinite(log))
    assert log[-1] < log[0] - 1.0, log
    preds = model.predict(data)
    assert set(preds) == {n for (_, n) in data._videos}


def test_cli_train_save_load_decode(tmp_path):
    """--classifier semimarkov end to end: closed-form training, pickle, reload, decode (like decode.sh)."""
    from action_segmentation_amd import cli
    out = str(tmp_path / 'model')
    s1 = cli.main(['--classifier', 'semimarkov', '--training', 'supervised', '--cuda', '--dataset', 'synthetic:tiny',
                   '--sm_max_span_length', '12', '--batch_size', '2', '--model_output_path', out])
    s2 = cli.main(['--classifier', 'semimarkov', '--cuda', '--dataset', 'synthetic:tiny', '--sm_max_span_length', '12',
                   '--batch_size', '2', '--model_input_path', out])
    assert s1 == s2 and s1['test_mof'] > 0.5


def test_cli_training_loop_decodes_every_epoch_and_keeps_the_best_snapshot(tmp_path, capsys):
    """main.py:207-264: per-epoch train + dev decode, epoch snapshots every 5 epochs, best model by training loss
    (unsupervised) written to <out>/<split>.pkl."""
    import os
    import pickle
    from action_segmentation_amd import cli
    out = str(tmp_path / 'model')
    stats = cli.main(['--classifier', 'semimarkov', '--training', 'unsupervised', '--cuda', '--dataset', 'synthetic:tiny',
                      '--sm_max_span_length', '12', '--batch_size', '2', '--epochs', '7', '--lr', '5e-2',
                      '--print_every', '0', '--model_output_path', out])
    hist = cli.train.last_history
    assert sorted(hist['stats_by_epoch']) == list(range(7))
    assert sorted(hist['dev_mof_by_epoch']) == list(range(7))                 # --dev_decode_frequency 1
    assert all(0.0 <= v <= 1.0 for v in hist['dev_mof_by_epoch'].values())
    assert sorted(os.listdir(out)) == ['synthetic.pkl', 'synthetic_epoch-0.pkl', 'synthetic_epoch-5.pkl']
    text = capsys.readouterr().out
    assert 'best train loss' in text and text.count('dev_mof') >= 7
    best_epoch = min(hist['stats_by_epoch'].items(), key=lambda t: t[1]['train_loss'])[0]
    with open(os.path.join(out, 'synthetic.pkl'), 'rb') as f:
        best = pickle.load(f)
    assert best.model.n_classes > 0 and 'test_mof' in stats
    assert 'best train loss %.4f in epoch %d' % (hist['stats_by_epoch'][best_epoch]['train_loss'], best_epoch) in text


def test_cli_supervised_gradient_training_stops_on_dev(tmp_path, capsys):
    """Supervised gradient-based training: early stopping on dev MoF (main.py:248-252)."""
    from action_segmentation_amd import cli
    cli.main(['--classifier', 'semimarkov', '--training', 'supervised', '--sm_supervised_method', 'closed-then-gradient',
              '--cuda', '--dataset', 'synthetic:tiny', '--sm_max_span_length', '12', '--batch_size', '2', '--epochs', '3',
              '--print_every', '0', '--dev_decode_frequency', '2'])
    hist = cli.train.last_history
    assert sorted(hist['stats_by_epoch']) == [-1, 0, 1, 2]                   # the closed-form stage calls back with -1
    assert sorted(hist['dev_mof_by_epoch']) == [-1, 0, 2]
    assert 'best dev mof' in capsys.readouterr().out


@pytest.mark.parametrize('constrain,by_index', [(False, False), (True, False), (False, True)])
def test_packed_log_likelihood_equals_per_batch(constrain, by_index, monkeypatch):
    """One launch for many single-task batches (log_likelihood_packed) == the reference's batch-by-batch calls: the
    per-batch values and the gradient of their mean (what --batch_accumulation forms) agree to 1e-9.  by_index: the per-batch
    means the way very large corpora take them (sums by index instead of one product with a dense matrix)."""
    from action_segmentation_amd.batching import pack_batches
    from action_segmentation_amd import semimarkov_modules
    if by_index:
        monkeypatch.setattr(semimarkov_modules, 'BATCH_MEAN_DENSE_MAX', 0)
    data = synth.SynthDatasplit('tiny', seed=12)
    narr = ['train'] if constrain else []
    args = synth.make_args(data.max_k, cuda=True, batch_size=2, sm_constrain_transitions=constrain,
                           sm_constrain_with_narration=narr)
    torch.manual_seed(0)
    model = SemiMarkovModel.from_args(args, data)
    m = model.model
    with torch.no_grad():
        m.gaussian_means.normal_(0, 0.3)
        m.poisson_log_rates.uniform_(1.0, 2.0)
        m.transition_logits.normal_()
    batches = list(make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=2))
    cons_fn = model._train_constraints(data)
    names = ['poisson_log_rates', 'gaussian_means', 'transition_logits', 'init_logits']
    # reference pattern
    m.zero_grad()
    lls = []
    for b in batches:
        ll, _ = m.log_likelihood(b['features'].to(model.device), b['lengths'], b['task_indices'], spans=None,
                                 additional_allowed_ends_per_instance=model.make_additional_allowed_ends(b['task_name'], b['lengths']),
                                 constraints=cons_fn(b) if cons_fn else None)
        lls.append(ll)
    (-(sum(lls) / len(lls))).backward()
    ref_ll = torch.stack(lls).detach().cpu().numpy()
    ref_g = {n: getattr(m, n).grad.detach().cpu().double().numpy().copy() for n in names}
    # packed
    m.zero_grad()
    pc = pack_batches(batches, model.device, m.max_k, constraints_fn=cons_fn,
                      additional_ends_fn=lambda b: model.make_additional_allowed_ends(b['task_name'], b['lengths']))
    ll_p = m.log_likelihood_packed(pc)
    (-ll_p.mean()).backward()
    np.testing.assert_allclose(ll_p.detach().cpu().numpy(), ref_ll, rtol=1e-12, atol=1e-9)
    for n in names:
        got = getattr(m, n).grad.detach().cpu().double().numpy()
        # (the parameters and their .grad are fp32: agreement to fp32 rounding of sums formed in a different order)
        np.testing.assert_allclose(got, ref_g[n], rtol=2e-6, atol=2e-6 * max(1.0, np.abs(ref_g[n]).max()), err_msg=n)


def test_fit_with_batch_accumulation_uses_packed_launches():
    data = synth.SynthDatasplit('tiny', seed=9)
    args = synth.make_args(data.max_k, cuda=True, batch_size=2, epochs=5, lr=5e-2, print_every=0, batch_accumulation=3)
    model = SemiMarkovModel.from_args(args, data)
    log = []
    model.fit(data, use_labels=False, callback_fn=lambda ep, st: log.append(st['train_loss']))
    assert len(log) == 5 and all(np.isfinite(log)) and log[-1] < log[0] - 0.5, log


def test_fit_counts_the_leftover_batches_of_an_epoch():
    """6 batches, --batch_accumulation 4: one optimiser step on four batches, two left over.  Their losses count in
    train_loss / train_nll (reference semimarkov.py:273-310 appends every batch's loss and steps only on full groups).
    lr = 0 keeps the parameters in place, so the epoch's statistics must equal the directly computed per-batch values."""
    data = synth.SynthDatasplit('tiny', seed=9)
    args = synth.make_args(data.max_k, cuda=True, batch_size=2, epochs=1, lr=0.0, print_every=0, batch_accumulation=4)
    model = SemiMarkovModel.from_args(args, data)
    log = []
    model.fit(data, use_labels=False, callback_fn=lambda ep, st: log.append(st))
    batches = list(make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=2))
    assert len(batches) == 6
    m = model.model
    nll, frames, per_batch = 0.0, 0, []
    with torch.no_grad():
        for b in batches:
            ll, _ = m.log_likelihood(b['features'].to(model.device), b['lengths'], b['task_indices'], spans=None)
            per_batch.append(-float(ll))
            nll += -float(ll) * len(b['lengths'])
            frames += int(b['lengths'].sum())
    assert len(log) == 1
    np.testing.assert_allclose(log[0]['train_loss'], np.mean(per_batch), rtol=1e-9)
    np.testing.assert_allclose(log[0]['train_nll_frame_avg'], nll / frames, rtol=1e-9)


@pytest.mark.parametrize('constrain', [False, True])
def test_predict_host_streams_slabs_and_equals_predict(constrain):
    """Features that stay in host memory (pinned slabs, double-buffered upload on a copy stream overlapped with the decode of
    the previous slab) decode to exactly what the device-resident corpus decodes to -- with and without constraints."""
    data, args, model = build(constrain, ('test',) if constrain else ())
    want = model.predict(data)
    host = data.subset(10 ** 9)
    assert all(not smp['features'].is_cuda for smp in host._videos.values())
    for n_slabs in (1, 3, 6):
        got = model.predict_host(host, n_slabs=n_slabs)
        assert set(got) == set(want)
        for name in want:
            np.testing.assert_array_equal(got[name], want[name], err_msg='%s (%d slabs)' % (name, n_slabs))
    slabs = model.prepare_host(host, 3)
    assert 2 <= len(slabs) <= 3 and all(pc.x.is_pinned() and not pc.x.is_cuda for pc in slabs)
    assert sum(pc.n_videos for pc in slabs) == len(want)
