"""Child process of tests/test_gpu_rccl.py: ONE rank, a real RCCL ('nccl') process group on the box's one GPU, and every
collective of the N-rank decode / training path pushed through it (action_segmentation_amd.distributed with
SMM_DIST_SINGLE_RANK=1).  Prints one JSON line; exits non-zero on any mismatch."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SMM_DIST_SINGLE_RANK'] = '1'

import torch.distributed as dist                                  # noqa: E402
from action_segmentation_amd import distributed as D, evaluation, synth   # noqa: E402
from action_segmentation_amd.semimarkov import SemiMarkovModel    # noqa: E402


def main():
    rank, world = D.init('nccl')
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == 'nccl' and D.active()
    dev = torch.device('cuda', torch.cuda.current_device())
    assert D.reduce_device() == dev
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}

    # all_reduce_tensor: SUM and MAX of device tensors go straight through RCCL; a host tensor takes the round trip
    t = torch.arange(1000, dtype=torch.float64, device=dev) * 0.5
    r = D.all_reduce_tensor(t.clone())
    assert r.is_cuda and torch.equal(r, t)
    r = D.all_reduce_tensor(t.clone(), op=dist.ReduceOp.MAX)
    assert torch.equal(r, t)
    i64 = torch.arange(-5, 300, dtype=torch.int64, device=dev)
    assert torch.equal(D.all_reduce_tensor(i64.clone()), i64)
    host = torch.tensor([1.5, 2.5], dtype=torch.float64)
    r = D.all_reduce_tensor(host.clone())
    assert not r.is_cuda and torch.equal(r, host)
    red = D.all_reduce_counters({'mof': [3, 7], 'frames': [11, 2]})
    assert red == {'mof': [3.0, 7.0], 'frames': [11.0, 2.0]}

    # a model: closed-form fit on the host, decode on the device
    data = synth.SynthDatasplit('tiny', seed=4)
    fitted = SemiMarkovModel.from_args(synth.make_args(data.max_k, cuda=False, batch_size=2), data)
    fitted.fit(data, use_labels=True)
    model = SemiMarkovModel.from_args(synth.make_args(data.max_k, cuda=True, batch_size=2), data)
    model.model.load_state_dict(fitted.model.state_dict(), strict=False)
    model.model.cuda()

    # broadcast_parameters: float, and bool (as uint8) buffers through RCCL; values unchanged
    before = {k: v.detach().clone() for k, v in model.model.state_dict().items()}
    D.broadcast_parameters(model.model, src=0)
    for k, v in model.model.state_dict().items():
        assert torch.equal(v, before[k]), k
    cons = SemiMarkovModel.from_args(synth.make_args(data.max_k, cuda=True, batch_size=2, sm_constrain_transitions=True), data)
    b2 = {k: v.detach().clone() for k, v in cons.model.state_dict().items()}
    assert any(v.dtype == torch.bool for v in b2.values())
    D.broadcast_parameters(cons.model, src=0)
    for k, v in cons.model.state_dict().items():
        assert torch.equal(v, b2[k]), k

    # all_reduce_gradients: one flat fp64 buffer, sum and mean (world 1: both are the gradients themselves)
    params = [p for p in model.model.parameters() if p.requires_grad]
    g = torch.Generator(device='cpu').manual_seed(0)
    for p in params[:-1]:
        p.grad = torch.randn(p.shape, generator=g).to(p.device)
    params[-1].grad = None                                         # (a rank that had no batch of the step)
    want = [None if p.grad is None else p.grad.clone() for p in params]
    D.all_reduce_gradients(params, average=False)
    D.all_reduce_gradients(params, average=True)
    for p, w in zip(params, want):
        assert torch.equal(p.grad, torch.zeros_like(p) if w is None else w)
    model.model.zero_grad()

    # sharded predict + evaluation counters reduced over RCCL == the plain evaluation
    preds = model.predict(data)                                    # shard = (0, 1) from the group
    plain = evaluation.accuracy_corpus(data, preds, False, seed=3)
    reduced = evaluation.accuracy_corpus(data, preds, False, seed=3, reduce=D.all_reduce_tensor)
    assert set(plain) == set(reduced)
    for task in plain:
        for key, pair in plain[task].items():
            np.testing.assert_allclose(np.asarray(reduced[task][key], dtype=np.float64), np.asarray(pair, dtype=np.float64),
                                       rtol=0, atol=0, err_msg='%s %s' % (task, key))
    opt_plain = evaluation.accuracy_corpus(data, preds, True, seed=3)
    opt_red = evaluation.accuracy_corpus(data, preds, True, seed=3, reduce=D.all_reduce_tensor)
    for task in opt_plain:
        for key, pair in opt_plain[task].items():
            np.testing.assert_allclose(np.asarray(opt_red[task][key], dtype=np.float64), np.asarray(pair, dtype=np.float64),
                                       rtol=0, atol=0, err_msg='%s %s' % (task, key))
    out["mof"] = float(evaluation.summarise(reduced, evaluation.STAT_KEYS)['mof'])

    # a data-parallel training step (gradient all-reduce inside SemiMarkovModel.fit) over the one-rank group
    targs = synth.make_args(data.max_k, cuda=True, batch_size=2, epochs=1, lr=1e-2, print_every=0, batch_accumulation=2)
    tm = SemiMarkovModel.from_args(targs, data)
    losses = []
    tm.fit(data, use_labels=False, callback_fn=lambda e, s: losses.append(s['train_loss']))
    assert len(losses) == 1 and np.isfinite(losses[0])
    out["train_loss"] = float(losses[0])
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
