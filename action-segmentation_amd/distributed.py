"""Multi-GPU harness of the decode path: shard videos, reduce metric counters.  (New in this build: the reference
has no distributed code)

One process per GPU (``torchrun``); videos are independent, so each rank decodes its own shard and the only
collectives are one ``all_reduce(SUM)`` of a packed fp64 vector of additive counters (``[numerator, denominator]``
pairs like ``src/evaluation/accuracy.py:460-467``) and, for timing, one ``all_reduce(MAX)``.  Works with the
``nccl`` (= RCCL over xGMI on ROCm) and ``gloo`` backends.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); no-op for one process."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world <= 1 or dist.is_initialized():
        return int(os.environ.get('RANK', 0)), world
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def shard_batches(batches, costs, rank, world):
    """Greedy longest-processing-time assignment of whole (single-task) batches to ranks.

    ``costs[i]`` ~ sum over the batch's videos of T * (K * C + C^2).  Deterministic, so every rank computes the same
    partition without communicating.  Returns the indices owned by ``rank`` in their original order.
    """
    order = sorted(range(len(batches)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    owner = [0] * len(batches)
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += costs[i]
    return [i for i in range(len(batches)) if owner[i] == rank]


def all_reduce_counters(counters, device=None):
    """``{key: [num, den]}`` (python / numpy numbers) -> the same dict summed over all ranks.

    Keys must be identical on every rank (they are sorted before packing).  Non-additive statistics must be carried
    as sums and finalised after the reduce.
    """
    keys = sorted(counters)
    flat = torch.tensor([float(v) for k in keys for v in counters[k]], dtype=torch.float64, device=device or 'cpu')
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat = flat.cpu().tolist()
    out, i = {}, 0
    for k in keys:
        n = len(counters[k])
        out[k] = flat[i:i + n]
        i += n
    return out


def frame_accuracy_counters(predictions, ground_truth, background=()):
    """MoF / MoF-without-background numerators and denominators of one rank's predictions
    (``src/evaluation/accuracy.py:475-579`` restricted to the single-label case)."""
    import numpy as np
    bkg = np.array(sorted(background), dtype=np.int64)
    mof = [0, 0]
    non_bg = [0, 0]
    for name, pred in predictions.items():
        gt = np.asarray(ground_truth[name])
        pred = np.asarray(pred)
        mof[0] += int((pred == gt).sum())
        mof[1] += int(gt.shape[0])
        keep = ~np.isin(gt, bkg)
        non_bg[0] += int((pred[keep] == gt[keep]).sum())
        non_bg[1] += int(keep.sum())
    return {'mof': mof, 'mof_non_bg': non_bg, 'frames': [sum(len(p) for p in predictions.values()), len(predictions)]}
