"""Host logic of SemiMarkovModel on the tiny synthetic corpus (CPU only): batching contract, packing, closed-form fit."""
import numpy as np
import pytest
import torch

from action_segmentation_amd import synth
from action_segmentation_amd.batching import make_data_loader, pack_batches
from action_segmentation_amd.semimarkov import SemiMarkovModel


def tiny_model(constrain=False, narration=()):
    data = synth.SynthDatasplit('tiny', seed=3)
    args = synth.make_args(data.max_k, cuda=False, batch_size=2, sm_constrain_transitions=constrain,
                           sm_constrain_with_narration=list(narration))
    model = SemiMarkovModel.from_args(args, data)
    return data, args, model


def test_batch_contract_matches_reference_collate():
    data, args, model = tiny_model()
    loader = make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=2)
    batches = list(loader)
    assert len(batches) == 3 * 2                                  # 3 tasks x 4 videos / 2
    for b in batches:
        assert len(set(b['task_name'])) == 1 and b['video_name'] == sorted(b['video_name'])
        assert b['features'].shape[:2] == (2, int(b['lengths'].max())) and b['features'].dtype == torch.float32
        for i, t in enumerate(b['lengths'].tolist()):
            assert float(b['features'][i, t:].abs().sum()) == 0.0     # zero padding
        assert all(torch.equal(t, b['task_indices'][0]) for t in b['task_indices'])


def test_closed_form_fit_recovers_generator_parameters():
    data, args, model = tiny_model()
    model.fit(data, use_labels=True)
    m = model.model
    x = torch.cat([smp['features'] for smp in data._videos.values()]).double().numpy()
    y = torch.cat([smp['gt_single'] for smp in data._videos.values()]).numpy()
    for c in sorted(set(y.tolist())):
        np.testing.assert_allclose(m.gaussian_means.detach().numpy()[c], x[y == c].mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(torch.diagonal(m.gaussian_cov).numpy(), x.var(0) + 1e-6, rtol=1e-5)
    assert torch.isfinite(m.transition_logits).all() and torch.isfinite(m.init_logits).all()


def test_pack_batches_layout_and_kp():
    data, args, model = tiny_model(constrain=True, narration=('test',))
    pc = model.prepare(data)
    assert pc.n_videos == 12 and pc.x.shape == (data.n_frames, data.feature_dim)
    assert pc.frame_offset == list(np.concatenate([[0], np.cumsum(pc.lengths)[:-1]]))
    loader = make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=2)
    i = 0
    for b in loader:
        tmax = int(b['lengths'].max())
        for j, t in enumerate(b['lengths'].tolist()):
            assert pc.kp[i] == min(args.sm_max_span_length, tmax) and pc.lengths[i] == t
            off = pc.frame_offset[i]
            assert torch.equal(pc.x[off:off + t], b['features'][j, :t])
            i += 1
    assert len(pc.groups) == 3 and pc.c_max == max(pc.n_states)
    for g, grp in enumerate(pc.groups):
        c = pc.n_states[g]
        assert pc.tables['class_map'][g, :c].tolist() == grp['valid_classes'].tolist()
        assert int(pc.tables['class_map'][g, c]) == model.n_classes
        np.testing.assert_allclose(pc.tables['trans'][g, :c, :c].exp().sum(0).numpy(), 1.0, rtol=1e-9)
    # ordering constraints: only the last state of each chain (or the additional one) may end a video
    assert pc.endpen is not None and ((pc.endpen == 0).sum(1) >= 1).all()
    assert pc.cons is not None and pc.cons.shape == (data.n_frames, pc.c_max)
    assert set(np.unique(pc.cons.numpy())) <= {0.0, np.float32(args.sm_constrain_narration_weight)}


def test_predict_without_gpu_fails_loudly():
    from action_segmentation_amd._lib import SmmError
    data, args, model = tiny_model()
    model.fit(data, use_labels=True)
    with pytest.raises(SmmError):
        model.predict(data)


def test_a_rank_without_batches_prepares_and_predicts_nothing():
    """More ranks than single-task batches (a small dev split on a big job): that rank's shard is an EMPTY corpus, its
    predict() returns {} without a launch (it still takes part in the evaluation's reductions), and the other ranks'
    shards partition the batches."""
    data, args, model = tiny_model()
    n_batches = 6
    world = n_batches + 2
    seen = []
    for rank in range(world):
        pc = model.prepare(data, shard=(rank, world))
        seen += list(pc.video_names)
        if pc.n_videos == 0:
            assert pc.x.shape[0] == 0 and pc.n_frames == 0
            assert model.predict(data, shard=(rank, world)) == {}
    assert sorted(seen) == sorted(n for names in data._videos_by_task.values() for n in names)
    assert pack_batches([], torch.device('cpu'), 12).n_videos == 0


def test_prepared_cache_key_covers_the_constraints():
    """The resident PackedCorpus bakes in the narration weight and the allowed ends: changing either is a cache miss."""
    data, args, model = tiny_model(constrain=True, narration=('test',))
    a = model.prepare(data)
    assert model.prepare(data) is a
    args.sm_constrain_narration_weight = -123.0
    b = model.prepare(data)
    assert b is not a and float(b.cons.min()) == -123.0
    model.clear_prepared()
    assert model.prepare(data) is not b


def test_shard_costs_follow_the_kernel_that_will_run():
    """batch_cost balances shards: lattice cells for spans up to 512; for longer spans (the Viterbi kernel's BAND mode) a
    time per frame that hardly depends on the state count -- and either way every batch lands on exactly one rank."""
    from action_segmentation_amd.batching import batch_cost
    from action_segmentation_amd.distributed import shard_batches
    data = synth.SynthDatasplit('tiny', seed=3)
    batches = data.batch_sampler(2, True, False).batches
    for max_k in (12, 1024):
        costs = [batch_cost(data, keys, max_k) for keys in batches]
        assert all(c > 0 for c in costs)
        owned = sorted(i for r in range(3) for i in shard_batches(batches, costs, r, 3))
        assert owned == list(range(len(batches)))
    # spans > 512: cost per frame between the 11- and the 23-state figure, i.e. nearly flat in the state count
    keys = batches[0]
    frames = sum(int(data[k]['features'].shape[0]) for k in keys)
    per_frame = batch_cost(data, keys, 1024) / frames
    # ... and it IS the library's own model of its kernel (include/smmdp.h: smm_band_frame_ns), not a second copy of the constants
    from action_segmentation_amd import _lib
    ns = _lib.load().smm_band_frame_ns
    c = len(data[keys[0]]['task_indices'])
    assert per_frame == pytest.approx(ns(c)) and ns(11) < ns(23) < 1.25 * ns(11)
    assert ns(0) == 0.0 and ns(33) == 0.0


def test_label_lease_pool_follows_the_last_reference(monkeypatch):
    """ops.lease_host_labels (the pinned buffers SemiMarkovModel.predict hands out views of): a buffer is out for as long as
    ANYTHING shares its storage -- a numpy slice made from the tensor counts, the tensor object itself need not survive --,
    at most LABEL_LEASES are out at once, and a dropped one is the next to be handed out.  (Pageable stand-ins for the pinned
    allocations: there is no GPU here; the GPU test covers the real thing.)"""
    from action_segmentation_amd import ops
    real_empty = torch.empty
    monkeypatch.setattr(torch, 'empty', lambda *a, **k: real_empty(*a, **{x: y for x, y in k.items() if x != 'pin_memory'}))
    monkeypatch.setattr(ops, '_label_leases', {})

    class B:
        def __init__(self, n, covered=None):
            self.total_frames, self.lengths = n, np.array([n if covered is None else covered])
    dev = torch.device('cuda', 0)
    if ops._storage_users(real_empty(1)) is None:
        assert ops.lease_host_labels(B(10), dev) is None          # a torch build that cannot tell never leases
        return
    a = ops.lease_host_labels(B(100), dev)
    view = a.numpy()[3:7]                                         # what predict() returns: numpy views
    ptr_a = a.data_ptr()
    del a
    taken = [ops.lease_host_labels(B(100), dev) for _ in range(ops.LABEL_LEASES - 1)]
    assert all(t is not None and t.data_ptr() != ptr_a for t in taken)
    assert ops.lease_host_labels(B(100), dev) is None             # all out
    del view
    again = ops.lease_host_labels(B(50), dev)
    assert again is not None and again.data_ptr() == ptr_a and again.numel() == 50
    del again, taken
    padded = ops.lease_host_labels(B(80, covered=60), dev)        # frames no video covers keep the -1 filler
    assert padded.numel() == 80 and int(padded.min()) == -1 and int(padded.max()) == -1
    bigger = ops.lease_host_labels(B(1000), dev)                  # a free buffer that is too small is replaced, not added
    assert bigger.numel() == 1000 and len(ops._label_leases[0]) <= ops.LABEL_LEASES
