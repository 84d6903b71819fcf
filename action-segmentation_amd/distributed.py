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


def init(backend=None, timeout_s=180):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); no-op for one process.

    ``backend``: 'nccl' (RCCL; the default whenever a GPU is visible) or 'gloo'.  Nothing is downgraded silently: an
    RCCL group that does not come up raises.  ``timeout_s`` bounds the rendezvous and every collective, so a rank that
    died cannot hang the others forever."""
    import datetime
    world = int(os.environ.get('WORLD_SIZE', 1))
    if (world <= 1 and not _single_rank_group()) or dist.is_initialized():
        return int(os.environ.get('RANK', 0)), world
    if world <= 1:
        # SMM_DIST_SINGLE_RANK=1 (tests): a ONE-rank group, so that the collectives of the N-rank path -- counters,
        # parameter broadcast, gradient all-reduce -- run through RCCL on a box with one GPU (tests/test_gpu_rccl.py)
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('LOCAL_RANK', '0')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as s:
                s.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(s.getsockname()[1])
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        local = int(os.environ.get('LOCAL_RANK', 0))
        if local >= torch.cuda.device_count():
            raise RuntimeError("rank with LOCAL_RANK=%d but only %d GPU(s) visible: RCCL needs one GPU per rank"
                               % (local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s))
    return dist.get_rank(), dist.get_world_size()


def _single_rank_group():
    return os.environ.get('SMM_DIST_SINGLE_RANK') == '1'


def active():
    """A process group is up and there is something to reduce over: more than one rank -- or one rank under the explicit
    test flag SMM_DIST_SINGLE_RANK=1, which sends every collective through the backend anyway."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _single_rank_group())


def reduce_device():
    """Where a tensor has to live to be reduced by the active backend: the current GPU for RCCL ('nccl' cannot reduce
    host tensors), the host for gloo."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def all_reduce_tensor(t, op=None):
    """Sum (or ``op``) of a tensor over the ranks, returned on the tensor's own device.  Device tensors go straight
    through RCCL; under gloo they take a round trip through the host.  No-op for a single process."""
    if not active():
        return t
    op = op or dist.ReduceOp.SUM
    rd = reduce_device()
    if t.device == rd:
        dist.all_reduce(t, op=op)
        return t
    buf = t.to(rd)
    dist.all_reduce(buf, op=op)
    return buf.to(t.device)


def broadcast_tensor(t, src=0):
    """Rank ``src``'s tensor on every rank (returned on the tensor's own device).  No-op for a single process."""
    if not active():
        return t
    rd = reduce_device()
    if t.device == rd:
        dist.broadcast(t, src)
        return t
    buf = t.to(rd)
    dist.broadcast(buf, src)
    return buf.to(t.device)


def broadcast_parameters(module, src=0):
    """Every rank ends with rank ``src``'s parameters and buffers (what loading the same pickle does in the reference
    flow, ``main.py:445-469``).  The closed-form fit sums with fp64 atomics, so two ranks fitting the same data may
    differ in the last bit; decode results are only comparable across world sizes when the parameters are identical."""
    if not active():
        return
    rd = reduce_device()
    with torch.no_grad():
        for _, t in sorted(module.state_dict().items()):
            buf = t.detach().to(rd).contiguous()
            if buf.dtype == torch.bool:
                b8 = buf.to(torch.uint8)
                dist.broadcast(b8, src)
                t.copy_(b8.to(torch.bool).to(t.device))
            else:
                dist.broadcast(buf, src)
                t.copy_(buf.to(t.device))


def all_reduce_gradients(parameters, average=False):
    """Sum (or mean) of the gradients over the ranks, as ONE collective on a flat fp32/fp64 buffer (five small tensors:
    SURVEY.md 8e, "a gradient all-reduce of the 5 small parameter tensors per step").  Parameters without a gradient
    on this rank (it had no batch of that step) count as zeros.  No-op for a single process."""
    if not active():
        return
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return
    with torch.no_grad():
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        flat = torch.cat([p.grad.reshape(-1).to(torch.float64) for p in params])
        flat = all_reduce_tensor(flat)
        if average:
            flat = flat / dist.get_world_size()
        i = 0
        for p in params:
            n = p.numel()
            p.grad.copy_(flat[i:i + n].view_as(p.grad).to(p.grad.dtype))
            i += n


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
    flat = torch.tensor([float(v) for k in keys for v in counters[k]], dtype=torch.float64,
                        device=device or reduce_device())
    flat = all_reduce_tensor(flat).cpu().tolist()
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
