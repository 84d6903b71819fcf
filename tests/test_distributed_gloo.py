"""N > 1 path on CPU: two gloo processes shard a corpus and reduce their metric counters (no GPU needed)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_decode(sample):
    """Deterministic stand-in for the GPU decode (this test is about sharding + reduction only)."""
    gt = sample['gt_single'].numpy()
    pred = gt.copy()
    pred[::7] = -5
    return pred


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from action_segmentation_amd import distributed as D, synth
    from action_segmentation_amd.batching import make_data_loader
    r, w = D.init('gloo')
    assert (r, w) == (rank, world)
    data = synth.SynthDatasplit('tiny', seed=5)
    args = synth.make_args(data.max_k, cuda=False, batch_size=2)
    batches = list(data.batch_sampler(2, True, False))
    costs = [sum(data[key]['features'].shape[0] * len(data[key]['task_indices']) for key in b) for b in batches]
    mine = D.shard_batches(batches, costs, rank, world)
    preds, gts = {}, {}
    for i in mine:
        for key in batches[i]:
            preds[key[1]] = _fake_decode(data[key])
            gts[key[1]] = data[key]['gt_single'].numpy()
    red = D.all_reduce_counters(D.frame_accuracy_counters(preds, gts, data.corpus._background_indices))
    out.put((rank, mine, red, float(sum(costs[i] for i in mine))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_and_reduce():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from action_segmentation_amd import distributed as D, synth
    data = synth.SynthDatasplit('tiny', seed=5)
    batches = list(data.batch_sampler(2, True, False))
    (r0, mine0, red0, load0), (r1, mine1, red1, load1) = res
    assert sorted(mine0 + mine1) == list(range(len(batches))) and not set(mine0) & set(mine1)
    assert red0 == red1                                            # every rank ends with the global counters
    assert max(load0, load1) <= 0.75 * (load0 + load1)             # LPT keeps the two shards comparable
    preds = {k[1]: _fake_decode(data[k]) for b in batches for k in b}
    gts = {k[1]: data[k]['gt_single'].numpy() for b in batches for k in b}
    single = D.all_reduce_counters(D.frame_accuracy_counters(preds, gts, data.corpus._background_indices))
    assert single == red0
    assert single['frames'][0] == data.n_frames
