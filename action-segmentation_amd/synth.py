"""Synthetic CrossTask-shaped corpora for tests and bench.py (there is no dataset in the build environment).

Implements just enough of the reference's ``Datasplit`` / ``Corpus`` protocol (``src/data/corpus.py:253-412, 647``,
``src/data/crosstask.py:328-388``) for ``SemiMarkovModel.from_args / fit / predict``: ``__getitem__((task, video))``
-> the per-video dict ``padding_colate`` expects, ``batch_sampler``, ``get_allowed_starts_and_transitions``,
``get_ordered_indices_no_background``, ``corpus.n_classes / _indices_by_task / _background_indices``.

Generator: every task is a chain BKG_0, step_1, BKG_1, ..., step_S, BKG_S (C1 = 2S+1 task-specific
classes, like ``--task_specific_steps --annotate_background_with_previous``) or, for ``chain=False``, C1 freely
ordered classes.  Ground truth is sampled from the HSMM itself: Poisson(rate_c) segment lengths with
rate_c ~ U(rate_lo, rate_hi) clipped to max_k-1, features x_t = mu_{z_t} + sigma * eps with mu_c ~ N(0, 0.3^2 I),
sigma_d ~ U(0.7, 1.3) -- per-frame emission log-probs of about -290 at D = 200, like PCA-200 CrossTask features.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from .batching import BatchSampler

CONFIGS = {
    # BASELINE.json configs[1]: batch of 64 videos, T=2048, K(states)=16, L=256
    'cfg2': dict(n_tasks=1, videos_per_task=64, steps=None, n_states=16, t_fixed=2048, max_k=256, d=200, chain=False,
                 rate=(20, 200), batch_size=64),
    # configs[0]-like: one long video of one task
    'cfg1': dict(n_tasks=1, videos_per_task=1, steps=None, n_states=20, t_fixed=10000, max_k=1024, d=200, chain=False,
                 rate=(20, 400), batch_size=1),
    # configs[2]: full CrossTask primary, 18 tasks, long videos T up to ~14k, L=1024 (the metric's shape); states per
    # task = 2*steps+1 = 11..23.  Videos with 22-23 states run as two-CU pairs.
    'cfg3': dict(n_tasks=18, videos_per_task=20, steps=(5, 11), t_lognormal=(6000, 0.5, 500, 14000), max_k=1024, d=200,
                 chain=True, rate=(20, 400), batch_size=5),
    # the same capped at 21 states per task (what one 8-wave workgroup holds at K = 1024), for comparison
    # configs[4]: the FULL CrossTask primary set (2750 videos over 18 tasks -> 153 per task) with cfg3's shapes; the
    # corpus of the strong-scaling leg (one corpus sharded by video over the ranks)
    'cfg5': dict(n_tasks=18, videos_per_task=153, steps=(5, 11), t_lognormal=(6000, 0.5, 500, 14000), max_k=1024, d=200,
                 chain=True, rate=(20, 400), batch_size=5),
    'cfg3c': dict(n_tasks=18, videos_per_task=20, steps=(5, 10), t_lognormal=(6000, 0.5, 500, 14000), max_k=1024, d=200,
                  chain=True, rate=(20, 400), batch_size=5),
    # configs[3]: ordering constraints + narration constraints, small shapes
    'cfg4': dict(n_tasks=6, videos_per_task=10, steps=(3, 7), t_lognormal=(900, 0.4, 200, 2048), max_k=64, d=200,
                 chain=True, rate=(10, 50), batch_size=5, narration=True),
    # the reference's DEFAULT shapes: --sm_max_span_length 20 (modules:55), --batch_size 5 (main.py:70), one
    # feature vector per second of CrossTask video (T of a few hundred), PCA-200 features
    'refdef': dict(n_tasks=18, videos_per_task=5, steps=(5, 11), t_lognormal=(300, 0.3, 100, 600), max_k=20, d=200,
                   chain=True, rate=(4, 16), batch_size=5),
    # tiny CPU-test corpus
    'tiny': dict(n_tasks=3, videos_per_task=4, steps=(2, 4), t_lognormal=(60, 0.3, 20, 120), max_k=12, d=8,
                 chain=True, rate=(3, 9), batch_size=2, narration=True),
}


class SynthCorpus:
    def __init__(self, n_classes, indices_by_task, background_indices):
        self.n_classes = n_classes
        self._indices_by_task = indices_by_task
        self._background_indices = background_indices


class SynthDatasplit(Dataset):
    def __init__(self, cfg, seed=0, device='cpu', scale=1.0, video_seed=None, keep=None):
        """``seed`` fixes the label space (tasks, steps, class means, rates); ``video_seed`` (optional) draws a
        different set of videos over the SAME label space -- a held-out split for a model fitted on ``seed``.
        ``keep``: None = every video gets features; otherwise a set of video names (possibly empty): the others are
        STRUCTURE ONLY (labels, lengths, ``features`` a storage-less 'meta' tensor of the right shape) -- what a rank of
        a sharded job holds for the videos of other ranks; the random streams advance identically either way, so the
        kept videos are bit-identical to those of the full corpus."""
        c = dict(CONFIGS[cfg]) if isinstance(cfg, str) else dict(cfg)
        self.cfg = c
        rng = np.random.default_rng(seed)
        vrng = None if video_seed is None else np.random.default_rng([seed, video_seed])
        self.feature_dim = c['d']
        self.max_k = c['max_k']
        self.remove_background = False
        d = c['d']
        self.sigma = rng.uniform(0.7, 1.3, size=d).astype(np.float32)
        n_classes = 0
        indices_by_task, background, self._videos_by_task = {}, [], {}
        self._ordered, self._steps = {}, {}
        self._videos = {}
        means, rates = [], []
        n_videos = max(1, int(round(c['videos_per_task'] * scale)))
        gen = torch.Generator(device=device).manual_seed(seed if video_seed is None else seed + 7919 * (video_seed + 1))
        for ti in range(c['n_tasks']):
            task = 'task%02d' % ti
            if c['steps'] is not None:
                s = int(rng.integers(c['steps'][0], c['steps'][1] + 1))
                c1 = 2 * s + 1
            else:
                s, c1 = 0, c['n_states']
            ids = list(range(n_classes, n_classes + c1))
            n_classes += c1
            indices_by_task[task] = ids
            if c['chain']:
                background += ids[0::2]
                self._steps[task] = ids[1::2]
            else:
                self._steps[task] = ids
            self._ordered[task] = ids
            mu = rng.normal(0, 0.3, size=(c1, d)).astype(np.float32)
            rt = rng.uniform(c['rate'][0], c['rate'][1], size=c1)
            means.append(mu)
            rates.append(rt)
            names = []
            mu_t = sigma_t = None
            for vi in range(n_videos):
                if 't_fixed' in c:
                    t = int(c['t_fixed'])
                else:
                    m, sg, lo, hi = c['t_lognormal']
                    t = int(np.clip(rng.lognormal(np.log(m), sg), lo, hi))
                labels_local = self._sample_labels(rng, t, c1, rt, c['chain'], c['max_k'])
                cons = self._narration(rng, labels_local, c1, t) if c.get('narration') else None
                if vrng is not None:            # the structural stream above stays in step with the seed-only split
                    if 't_fixed' not in c:
                        t = int(np.clip(vrng.lognormal(np.log(m), sg), lo, hi))
                    labels_local = self._sample_labels(vrng, t, c1, rt, c['chain'], c['max_k'])
                    cons = self._narration(vrng, labels_local, c1, t) if c.get('narration') else None
                name = '%s_v%03d' % (task, vi)
                names.append(name)
                lab = torch.from_numpy(labels_local)
                if keep is not None and name not in keep:
                    if len(keep):                      # keep the feature stream in step with the full corpus
                        torch.randn(t, d, generator=gen, device=device)   # (drawn and dropped)
                    x = torch.empty(t, d, device='meta')
                else:
                    if mu_t is None:
                        mu_t = torch.from_numpy(mu).to(device)
                        sigma_t = torch.from_numpy(self.sigma).to(device)
                    x = mu_t[lab.to(device)] + sigma_t * torch.randn(t, d, generator=gen, device=device)
                sample = dict(features=x, gt_single=lab + ids[0], task_name=task, video_name=name,
                              task_indices=torch.tensor(ids, dtype=torch.long))
                if cons is not None:
                    sample['constraints'] = cons
                self._videos[(task, name)] = sample
            self._videos_by_task[task] = names
        self.corpus = SynthCorpus(n_classes, indices_by_task, background)
        self.true_means = np.concatenate(means, 0)
        self.true_rates = np.concatenate(rates, 0)

    @staticmethod
    def _sample_labels(rng, t, c1, rates, chain, max_k):
        out, cur = [], 0 if chain else int(rng.integers(0, c1))
        total = 0
        while total < t:
            ln = int(np.clip(rng.poisson(rates[cur]), 1, max(max_k - 1, 1)))
            out.append(np.full(ln, cur, dtype=np.int64))
            total += ln
            if chain:
                cur = (cur + 1) % c1
            else:
                nxt = int(rng.integers(0, c1 - 1)) if c1 > 1 else 0
                cur = nxt + (nxt >= cur) if c1 > 1 else 0
        return np.concatenate(out)[:t]

    @staticmethod
    def _narration(rng, labels_local, c1, t):
        """T x S, 1 where a step may occur: a window around the frames where it really occurs (0 elsewhere)."""
        s = (c1 - 1) // 2
        cons = torch.zeros(t, s)
        for j in range(s):
            pos = np.flatnonzero(labels_local == 2 * j + 1)
            if len(pos) == 0:
                lo, hi = 0, t
            else:
                lo = max(0, int(pos.min()) - int(rng.integers(0, 20)))
                hi = min(t, int(pos.max()) + 1 + int(rng.integers(0, 20)))
            cons[lo:hi, j] = 1
        return cons

    # ---- Dataset / Datasplit protocol
    def __len__(self):
        return len(self._videos)

    def __getitem__(self, key):
        return self._videos[key]

    def batch_sampler(self, batch_size, batch_by_task, shuffle):
        return BatchSampler(self._videos_by_task, batch_size, batch_by_task, shuffle)

    def get_ordered_indices_no_background(self):
        return dict(self._steps)

    def get_allowed_starts_and_transitions(self):
        """Chain BKG_0 -> step_1 -> BKG_1 -> ... -> BKG_S per task (reference crosstask.py:349-386)."""
        starts, transitions, ends = set(), {}, set()
        for task, ids in self._ordered.items():
            for src, tgt in zip(ids, ids[1:]):
                transitions.setdefault(src, set()).add(tgt)
            starts.add(ids[0])
            ends.add(ids[-1])
        return starts, transitions, ends, dict(self._ordered)

    def subset(self, videos_per_task, max_frames=None):
        """Shallow copy holding the first ``videos_per_task`` videos of each task (optionally truncated): the
        'training split' the closed-form fit runs on."""
        other = object.__new__(SynthDatasplit)
        other.__dict__.update(self.__dict__)
        other._videos_by_task = {t: v[:videos_per_task] for t, v in self._videos_by_task.items()}
        other._videos = {}
        for t, names in other._videos_by_task.items():
            for n in names:
                smp = dict(self._videos[(t, n)])
                if max_frames is not None:
                    for k in ('features', 'gt_single', 'constraints'):
                        if k in smp:
                            smp[k] = smp[k][:max_frames]
                smp['features'] = smp['features'].cpu()
                other._videos[(t, n)] = smp
        return other

    @property
    def n_frames(self):
        return sum(int(v['features'].shape[0]) for v in self._videos.values())


def make_args(max_k, cuda=True, batch_size=5, **kw):
    """An ``args`` namespace with every flag SemiMarkovModel / SemiMarkovModule read (reference main.py flags)."""
    import argparse
    from .semimarkov import SemiMarkovModel
    from .batching import add_training_args
    p = argparse.ArgumentParser()
    SemiMarkovModel.add_args(p)
    add_training_args(p)
    p.add_argument('--cuda', action='store_true')
    p.add_argument('--batch_size', type=int, default=5)
    p.add_argument('--annotate_background_with_previous', action='store_true')
    p.add_argument('--no_merge_classes', action='store_true')
    a = p.parse_args([])
    a.sm_max_span_length = max_k
    a.cuda = cuda
    a.batch_size = batch_size
    for k, v in kw.items():
        setattr(a, k, v)
    return a
