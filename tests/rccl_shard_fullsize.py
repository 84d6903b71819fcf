"""Child process of tests/test_gpu_fullsize.py::test_cfg5_sharded_decode_at_cfg3_size_through_rccl: BASELINE config 5
(one corpus sharded by video, evaluation counters all-reduced) AT SIZE on the box's one GPU.  The cfg3 seed-2 corpus
(18 tasks x 20 videos, 11..23 states, T up to 14 000, K = 1024) is decoded as shard (r, 4) for r = 0..3; every shard's
labels are compared with the C twin's, the union with the unsharded decode, and the evaluation counters of the four
shards are summed and pushed through a real one-rank RCCL ('nccl') group (SMM_DIST_SINGLE_RANK=1) before they are
finalised -- what N ranks do (reference src/data/corpus.py:405-604 summed as src/main.py:486-532).  Prints one JSON line;
exits non-zero on any mismatch."""
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

WORLD = 4


class _Abort(Exception):
    pass


def sharded_counters(data, parts, through):
    """The three passes of a sharded evaluation (the second reduction depends on the first, as in a real job): each
    shard's tensors are summed over the shards here and the sum goes through ``through`` (the RCCL all-reduce)."""
    grab = {}

    def run(r, reduce):
        return evaluation.accuracy_corpus(data, parts[r], False, seed=3, reduce=reduce)

    def first(r):
        def reduce(t):
            grab[('conf', r)] = t.clone()
            raise _Abort
        return reduce
    for r in range(len(parts)):
        try:
            run(r, first(r))
        except _Abort:
            pass
    conf = through(sum(grab[('conf', r)] for r in range(len(parts))))

    def second(r):
        calls = []

        def reduce(t):
            calls.append(1)
            if len(calls) == 1:
                return conf.clone()
            grab[('sums', r)] = t.clone()
            raise _Abort
        return reduce
    for r in range(len(parts)):
        try:
            run(r, second(r))
        except _Abort:
            pass
    sums = through(sum(grab[('sums', r)] for r in range(len(parts))))

    def third():
        calls = []

        def reduce(t):
            calls.append(1)
            return conf.clone() if len(calls) == 1 else sums.clone()
        return reduce
    return [run(r, third()) for r in range(len(parts))]


def main():
    import bench
    rank, world = D.init('nccl')
    assert (rank, world) == (0, 1) and dist.get_backend() == 'nccl' and D.active()
    dev = torch.device('cuda', torch.cuda.current_device())
    cfg = synth.CONFIGS['cfg3']
    data = synth.SynthDatasplit('cfg3', seed=2, device=dev)
    mk = lambda: SemiMarkovModel.from_args(synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size']), data)
    fitted = mk()
    fitted.fit(data.subset(6), use_labels=True)
    model = mk()
    model.model.load_state_dict(fitted.model.state_dict(), strict=False)
    model.model.to(dev)

    full = model.predict(data, shard=(0, 1))
    assert len(full) == 360
    parts, frames = [], []
    seen = {}
    for r in range(WORLD):
        pc = model.prepare(data, shard=(r, WORLD))
        out = model.model.decode_packed(pc, want_spans=False, want_labels=True)
        torch.cuda.synchronize()
        labels = out['labels'].cpu().numpy()
        # the twin on this shard's videos, every frame
        _, par = bench.cpu_factored(pc, model, gpu_labels=labels, budget_s=1e9)
        assert par['videos_checked'] == pc.n_videos and par['frames_checked'] == pc.n_frames, par
        assert par['label_mismatches'] == 0, (r, par)
        part = model.predict(data, shard=(r, WORLD))
        assert len(part) == pc.n_videos and not set(part) & set(seen)
        seen.update(part)
        parts.append(part)
        frames.append(pc.n_frames)
        model.clear_prepared()
    assert set(seen) == set(full)
    for name in full:
        np.testing.assert_array_equal(seen[name], full[name], err_msg=name)
    assert max(frames) < 0.30 * sum(frames), frames                # the shards are balanced (4 ranks: 25 % each)

    single = evaluation.accuracy_corpus(data, full, False, seed=3)
    n_reduced = []

    def through(t):
        n_reduced.append(t.numel())
        return D.all_reduce_tensor(t.clone())
    for got in sharded_counters(data, parts, through):
        assert set(got) == set(single)
        for task in single:
            for key, pair in single[task].items():
                np.testing.assert_allclose(np.asarray(got[task][key], dtype=np.float64), np.asarray(pair, dtype=np.float64),
                                           rtol=1e-13, atol=0, err_msg='%s %s' % (task, key))
    assert len(n_reduced) == 2
    mof = float(evaluation.summarise(single, evaluation.STAT_KEYS)['mof'])
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"backend": "nccl", "world": 1, "shards": WORLD, "videos": len(full), "frames": int(sum(frames)),
                      "frames_per_shard": [int(f) for f in frames], "mof": mof, "reduced_elements": n_reduced}), flush=True)


if __name__ == '__main__':
    main()
