"""Batch contract of the decode path: single-task, zero-padded batches (reference ``src/models/model.py:42-77``,
``src/data/corpus.py:613-644``) and the packed, multi-task form the MI355X path prefers.

The reference feeds ``SemiMarkovModule`` batches of ``--batch_size`` (5) videos of ONE task, zero-padded to the
longest.  That is reproduced here for drop-in use (``padding_colate``, ``BatchSampler``, ``make_data_loader``).
A whole corpus decoded five videos at a time would leave 251 of 256 CUs idle, so ``pack_batches`` folds any number
of such batches into one ragged launch: frames of all videos on one packed axis, one parameter group per task,
and a per-video ``kp`` that keeps the one batch-dependent quantity of the reference (K clipped to the padded
length of the video's own batch, semimarkov_modules.py:450-452).
"""
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Sampler


def add_training_args(parser):
    """reference model.py:7-24"""
    parser.add_argument('--epochs', type=int, default=60)
    parser.add_argument('--batch_accumulation', type=int, default=1)
    parser.add_argument('--lr', type=float, default=5e-3)
    parser.add_argument('--workers', type=int, default=0)
    parser.add_argument('--max_grad_norm', type=float, default=10)
    parser.add_argument('--print_every', type=int, default=100)
    parser.add_argument('--no_reduce_plateau', action='store_true')
    parser.add_argument('--reduce_plateau_factor', type=float, default=0.2)
    parser.add_argument('--reduce_plateau_patience', type=float, default=1)
    parser.add_argument('--reduce_plateau_min_lr', type=float, default=1e-4)
    parser.add_argument('--train_limit', type=int)
    parser.add_argument('--dev_decode_frequency', type=int, default=1)


def make_optimizer(args, parameters):
    opt = torch.optim.Adam(parameters, lr=args.lr)
    sched = None
    if not args.no_reduce_plateau:
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(
            opt, factor=args.reduce_plateau_factor, patience=int(args.reduce_plateau_patience), min_lr=1e-4,
            threshold=1e-5)
    return opt, sched


PAD_KEYS = ('gt_single', 'features', 'constraints')
NOPAD_KEYS = ('task_name', 'video_name', 'task_indices', 'gt', 'gt_with_background')


def padding_colate(samples):
    """list of per-video dicts -> batch dict: ``features`` b x Tmax x D zero-padded, ``lengths`` b, lists for the rest."""
    samples = [s for s in samples if s is not None]
    keys = samples[0].keys()
    batch = {k: [s[k] for s in samples] for k in keys if k in NOPAD_KEYS}
    batch['lengths'] = torch.LongTensor([s['features'].size(0) for s in samples])
    for k in PAD_KEYS:
        if k in keys:
            seqs = [s[k] for s in samples]
            if k == 'constraints' and len({tuple(q.shape[1:]) for q in seqs}) > 1:
                continue          # mixed-task batch: per-task step counts differ, narration constraints do not apply
            batch[k] = torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True, padding_value=0)
    return batch


def ragged_colate(samples):
    """``padding_colate`` without the padding: ``features_list`` (and ``gt_single_list``) keep the per-video tensors as they
    are -- what ``SemiMarkovModel.predict(fused=False)`` hands to ``SemiMarkovModule.decode_ragged_launch`` (the decode
    kernels take a packed frame axis: the zero-padded b x Tmax x D layout is a copy the reference's torch ops needed, not
    the DP).  Single-task batches without narration constraints only."""
    samples = [s for s in samples if s is not None]
    keys = samples[0].keys()
    batch = {k: [s[k] for s in samples] for k in keys if k in NOPAD_KEYS}
    batch['lengths'] = torch.LongTensor([s['features'].size(0) for s in samples])
    batch['features_list'] = [s['features'] for s in samples]
    return batch


class BatchSampler(Sampler):
    """Consecutive chunks of the name-sorted videos of one task (corpus.py:613-644)."""

    def __init__(self, videos_by_task, batch_size, batch_by_task, shuffle, seed=1):
        self.random_state = random.Random(seed) if shuffle else None
        self.batches = []
        if batch_by_task:
            for task in sorted(videos_by_task):
                vids = sorted(videos_by_task[task])
                for i in range(0, len(vids), batch_size):
                    self.batches.append([(task, v) for v in vids[i:i + batch_size]])
        else:
            flat = [(t, v) for t in sorted(videos_by_task) for v in sorted(videos_by_task[t])]
            for i in range(0, len(flat), batch_size):
                self.batches.append(flat[i:i + batch_size])

    def __iter__(self):
        if self.random_state is not None:
            self.random_state.shuffle(self.batches)
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def batch_cost(datasplit, keys, max_k):
    """~ time of the DP over one single-task batch, for balancing shards (any unit; one corpus = one max_k).
    Spans of up to 512: the lattice cells, sum over the videos of T * ((K-1) * C + C^2).  Longer spans run in the Viterbi
    kernel's BAND mode, whose time per frame is its serial chain's and hardly depends on K or C (the library's
    ``smm_band_frame_ns``: 167 ns at 11 states, 191 at 23 with round 4's kernels) -- weighting a 23-state task twice as heavy as an 11-state one would hand its rank half the frames."""
    cost = 0.0
    band_ns = None
    if max_k > 512:
        from . import _lib
        band_ns = _lib.load().smm_band_frame_ns        # the shipped kernel's own model (include/smmdp.h): one constant, one place
    for key in keys:
        smp = datasplit[key]
        if smp is None:
            continue
        t = int(smp['features'].shape[0])
        c = len(smp['task_indices']) if smp.get('task_indices') is not None else datasplit.corpus.n_classes
        if max_k > 512:
            cost += t * band_ns(c)
        else:
            cost += t * ((min(max_k, t + 1) - 1) * c + c * c)
    return cost


def make_data_loader(args, datasplit, shuffle, batch_by_task, batch_size=1, shard=None, ragged=False):
    """``shard=(rank, world)``: keep only this rank's share of the batches (whole single-task batches, so the one
    batch-dependent quantity of the reference -- K clipped to the batch's padded length -- is unchanged; greedy
    longest-processing-time assignment on the DP work, the same on every rank without communication)."""
    sampler = datasplit.batch_sampler(batch_size, batch_by_task, shuffle)
    if shard is not None and shard[1] > 1:
        from .distributed import shard_batches
        assert not shuffle, "shards are cut from the deterministic batch order"
        max_k = getattr(args, 'sm_max_span_length', None) or 1
        costs = [batch_cost(datasplit, keys, max_k) for keys in sampler.batches]
        sampler.batches = [sampler.batches[i] for i in shard_batches(sampler.batches, costs, shard[0], shard[1])]
    return DataLoader(datasplit, num_workers=getattr(args, 'workers', 0), collate_fn=ragged_colate if ragged else padding_colate,
                      batch_sampler=sampler)


class PackedCorpus:
    """Device-resident ragged decode set (see module docstring).  Built by ``pack_batches``."""

    def __init__(self):
        self.x = None              # fp32 [total_frames, D]
        self.cons = None           # fp32 [total_frames, c_max] or None
        self.lengths = []          # per video
        self.frame_offset = []
        self.group = []
        self.kp = []
        self.video_names = []
        self.task_names = []
        self.groups = []           # per group: dict(task, valid_classes (LongTensor or None))
        self.additional_ends = []  # per video (list) or None
        self.batch_index = []      # per video: index of the source batch it came from
        self.k_rows = None
        self.tables = None         # stacked per-group fp64 tables on the device (filled by the module)
        self.device = None         # set when x stays on the host (pack_batches(keep_on_host=True)): the GPU to decode on

    @property
    def n_videos(self):
        return len(self.lengths)

    @property
    def n_frames(self):
        return int(np.sum(self.lengths))


def pack_batches(batches, device, max_k, constraints_fn=None, additional_ends_fn=None, keep_on_host=False):
    """Fold reference-style batches (dicts from ``padding_colate``) into one PackedCorpus on ``device``.
    ``keep_on_host``: the packed features stay on the host in PINNED memory (``pc.x``; ``pc.device`` names the GPU the
    tables and the metadata are built for): the form ``SemiMarkovModel.predict_host`` streams to the device slab by slab.

    Each source batch contributes its videos with kp = min(max_k-rows, Tmax of that batch).
    ``constraints_fn(batch) -> b x Tmax x C tensor or None``; ``additional_ends_fn(batch) -> list or None``.
    """
    pc = PackedCorpus()
    feats, cons_l, any_cons = [], [], False
    group_of = {}
    off = 0
    k_rows = max(max_k, 2)
    for bi, batch in enumerate(batches):
        tasks = batch['task_name']
        assert len(set(tasks)) == 1, "a source batch holds one task"
        task = tasks[0]
        vc = batch.get('task_indices')
        vc0 = None if vc is None else vc[0].detach().cpu().long()
        key = (task, None if vc0 is None else tuple(vc0.tolist()))
        if key not in group_of:
            group_of[key] = len(pc.groups)
            pc.groups.append(dict(task=task, valid_classes=vc0))
        g = group_of[key]
        lengths = batch['lengths'].tolist()
        tmax = int(batch['features'].size(1))
        cons = constraints_fn(batch) if constraints_fn else None
        addl = additional_ends_fn(batch) if additional_ends_fn else None
        for i, t in enumerate(lengths):
            feats.append(batch['features'][i, :t])
            if cons is not None:
                cons_l.append(cons[i, :t])
                any_cons = True
            else:
                cons_l.append(None)
            pc.lengths.append(int(t))
            pc.frame_offset.append(off)
            pc.group.append(g)
            pc.kp.append(min(k_rows, tmax))
            pc.video_names.append(batch['video_name'][i])
            pc.task_names.append(task)
            pc.additional_ends.append(None if addl is None else addl[i])
            pc.batch_index.append(bi)
            off += int(t)
    pc.k_rows = k_rows
    if not feats:                       # an empty shard (more ranks than single-task batches): a corpus without videos
        pc.x = torch.zeros((0, 1), dtype=torch.float32, device=device)
        pc.cons_list = None
        return pc
    if keep_on_host:
        pc.x = torch.empty((off, int(feats[0].shape[1])), dtype=torch.float32, pin_memory=True)
        torch.cat([f.to(device='cpu', dtype=torch.float32) for f in feats], dim=0, out=pc.x)
        pc.device = torch.device(device)
    else:
        pc.x = torch.cat([f.to(device=device, dtype=torch.float32) for f in feats], dim=0).contiguous()
    if any_cons:
        pc.cons_list = cons_l
    else:
        pc.cons_list = None
    return pc
