"""Multi-GPU path on ONE GPU: a corpus sharded by video decodes to the same labels, and the evaluation counters summed
over the shards give the statistics of the whole corpus (reference src/data/corpus.py:405-604 summed as
src/main.py:486-532)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from action_segmentation_amd import evaluation, synth
from action_segmentation_amd.semimarkov import SemiMarkovModel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(seed=4):
    data = synth.SynthDatasplit('tiny', seed=seed)
    fit_args = synth.make_args(data.max_k, cuda=False, batch_size=2)
    fitted = SemiMarkovModel.from_args(fit_args, data)
    fitted.fit(data, use_labels=True)
    args = synth.make_args(data.max_k, cuda=True, batch_size=2)
    model = SemiMarkovModel.from_args(args, data)
    model.model.load_state_dict(fitted.model.state_dict(), strict=False)
    model.model.cuda()
    return data, model


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_predict_is_a_partition_of_predict(world):
    data, model = build()
    full = model.predict(data)
    seen = {}
    for rank in range(world):
        part = model.predict(data, shard=(rank, world))
        assert not set(part) & set(seen)
        seen.update(part)
    assert set(seen) == set(full)
    for name in full:
        np.testing.assert_array_equal(seen[name], full[name], err_msg=name)


class _Abort(Exception):
    pass


def _sharded_stats(data, preds_by_rank, optimal):
    """evaluate every shard with a ``reduce`` that sums over the shards (three passes: the second reduction depends on
    the result of the first, exactly as in a real job where both are collectives)."""
    world = len(preds_by_rank)
    grab = {}

    def run(rank, reduce):
        return evaluation.accuracy_corpus(data, preds_by_rank[rank], optimal, seed=3, reduce=reduce)

    def first(rank):
        def reduce(t):
            grab[('conf', rank)] = t.clone()
            raise _Abort
        return reduce
    for r in range(world):
        with pytest.raises(_Abort):
            run(r, first(r))
    conf = sum(grab[('conf', r)] for r in range(world))

    def second(rank):
        calls = []

        def reduce(t):
            calls.append(1)
            if len(calls) == 1:
                return conf.clone()
            grab[('sums', rank)] = t.clone()
            raise _Abort
        return reduce
    for r in range(world):
        with pytest.raises(_Abort):
            run(r, second(r))
    sums = sum(grab[('sums', r)] for r in range(world))

    def third():
        calls = []

        def reduce(t):
            calls.append(1)
            return conf.clone() if len(calls) == 1 else sums.clone()
        return reduce
    return [run(r, third()) for r in range(world)]


@pytest.mark.parametrize('optimal', [False, True])
def test_reduced_counters_equal_single_process(optimal):
    data, model = build(seed=6)
    full = model.predict(data)
    rng = np.random.default_rng(0)
    noisy = {}
    for (task, name), smp in data._videos.items():          # imperfect predictions: every counter gets exercised
        p = full[name].copy()
        ids = smp['task_indices'].numpy()
        flip = rng.random(p.shape[0]) < 0.15
        p[flip] = rng.choice(ids, size=int(flip.sum()))
        noisy[name] = p
    single = evaluation.accuracy_corpus(data, noisy, optimal, seed=3)
    names = sorted(noisy)
    shards = [{n: noisy[n] for n in names[0::3]}, {n: noisy[n] for n in names[1::3]}, {n: noisy[n] for n in names[2::3]},
              {}]                                            # (a rank without videos takes part too)
    for got in _sharded_stats(data, shards, optimal):
        assert set(got) == set(single)
        for task in single:
            for key, pair in single[task].items():
                np.testing.assert_allclose(np.asarray(got[task][key], dtype=np.float64),
                                           np.asarray(pair, dtype=np.float64), rtol=1e-13, atol=0, err_msg='%s %s' % (task, key))


def _bench(args, timeout=900):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_two_ranks_one_corpus_same_stats_as_one_rank():
    """`python bench.py --gpus 2` spawns two ranks (here both on the one GPU of the box, gloo for the reductions --
    RCCL refuses two ranks on one device) and the strong-scaling leg's reduced statistics equal the one-rank ones."""
    common = ['--workload', 'tiny', '--strong-workload', 'tiny', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
              '--no-predict-e2e', '--seed', '5']
    one = _bench(['--gpus', '1', '--strong-leg'] + common)
    two = _bench(['--gpus', '2', '--backend', 'gloo', '--share-gpus'] + common)
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2
    # ONE headline leg for every N (round 5): weak, the same workload string at N = 1 and N = 2
    assert two['backend'] == 'gloo' and two['scaling'] == 'weak' == one['scaling']
    assert two['config']['workload'] == one['config']['workload']
    s1, s2 = one['strong_scaling'], two['strong_scaling']
    assert s2['n_gpus'] == 2 and s1['frames'] == s2['frames'] and s1['videos'] == s2['videos']
    assert s2['max_frames_on_a_rank'] < s2['frames']
    assert s1['stats'] == s2['stats']
    assert two['strong_scaling']['value'] > 0 and two['value'] == two['weak_scaling']['value']
    assert s1['workload'] == s2['workload']
    strong_head = _bench(['--gpus', '2', '--backend', 'gloo', '--share-gpus', '--scaling', 'strong'] + common)
    assert strong_head['scaling'] == 'strong' and strong_head['value'] == strong_head['strong_scaling']['value']
    assert strong_head['config']['workload'] == s2['workload']
    assert strong_head['strong_scaling']['stats'] == s1['stats']


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_bench_two_ranks_rccl():
    common = ['--workload', 'tiny', '--strong-workload', 'tiny', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
              '--no-predict-e2e', '--seed', '5']
    one = _bench(['--gpus', '1', '--strong-leg'] + common)
    two = _bench(['--gpus', '2'] + common)
    assert two['n_gpus'] == 2 and two['backend'] == 'nccl'
    assert one['strong_scaling']['stats'] == two['strong_scaling']['stats']


def test_bench_under_torchrun_launcher():
    """The driver's launch line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (ranks from the environment; here two ranks on the box's one GPU over gloo)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--share-gpus',
           '--workload', 'tiny', '--strong-workload', 'tiny', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
           '--no-predict-e2e', '--seed', '5']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['strong_scaling']['n_gpus'] == 2 and line['scaling'] == 'weak'
    one = _bench(['--gpus', '1', '--strong-leg', '--workload', 'tiny', '--strong-workload', 'tiny', '--steps', '2', '--warmup', '1',
                  '--no-cpu-baseline', '--no-predict-e2e', '--seed', '5'])
    assert one['strong_scaling']['stats'] == line['strong_scaling']['stats']


def _train(world, accum, tmp_path):
    import socket
    out = str(tmp_path / ('train_w%d_a%d.json' % (world, accum)))
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    script = os.path.join(ROOT, 'scripts', 'dp_train_check.py')
    if world == 1:
        cmd = [sys.executable, script, out, str(accum)]
    else:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
               '--master-addr', '127.0.0.1', '--master-port', str(port), script, out, str(accum)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out))


@pytest.mark.parametrize('accum', [4, 1])
def test_data_parallel_training_reproduces_the_single_process_parameters(accum, tmp_path):
    """unsupervised training shards over the ranks with one gradient all-reduce per optimiser step.  Two
    ranks (gloo, both on the box's GPU) must end where one process ends: with --batch_accumulation 4 each rank
    back-propagates two of the four batches of a step (packed launch); with 1 the ranks repeat the batch and average.
    (The kernels' fp64 atomics round differently from run to run: equal to 1e-6, not bit for bit.)"""
    one = _train(1, accum, tmp_path)
    two = _train(2, accum, tmp_path)
    assert one['world'] == 1 and two['world'] == 2
    for (e1, l1), (e2, l2) in zip(one['logs'], two['logs']):
        assert e1 == e2 and abs(l1 - l2) <= 1e-6 * max(1.0, abs(l1)), (one['logs'], two['logs'])
    moved = 0
    for name, v1 in one['state'].items():
        a, b = np.asarray(v1), np.asarray(two['state'][name])
        np.testing.assert_allclose(b, a, rtol=1e-5, atol=1e-6, err_msg=name)
        moved += float(np.abs(a).sum())
    assert len(one['logs']) == 2 and moved > 0


def test_cli_trains_and_evaluates_under_torchrun(tmp_path):
    """`python -m torch.distributed.run ... -m action_segmentation_amd.cli --training unsupervised`: two ranks (gloo
    rehearsal on one GPU) train data-parallel, decode sharded train / dev sets every epoch, reduce the counters, and
    rank 0 writes the snapshots; the per-epoch training losses equal those of the one-process run."""
    import socket
    out1, out2 = str(tmp_path / 'm1'), str(tmp_path / 'm2')
    base = ['--classifier', 'semimarkov', '--training', 'unsupervised', '--cuda', '--dataset', 'synthetic:tiny',
            '--sm_max_span_length', '12', '--batch_size', '2', '--epochs', '2', '--lr', '5e-2', '--print_every', '0',
            '--batch_accumulation', '2']
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r1 = subprocess.run([sys.executable, '-m', 'action_segmentation_amd.cli'] + base + ['--model_output_path', out1],
                        env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-3000:]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env2 = dict(env, SMM_DIST_BACKEND='gloo')
    r2 = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                         '--master-addr', '127.0.0.1', '--master-port', str(port), '-m', 'action_segmentation_amd.cli']
                        + base + ['--model_output_path', out2], env=env2, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert sorted(os.listdir(out1)) == sorted(os.listdir(out2)) == ['synthetic.pkl', 'synthetic_epoch-0.pkl']

    # Training is reproduced (same per-epoch losses); the decode statistics of this two-epoch model are not compared:
    # its states are still nearly symmetric (all means start at the data mean), so parameters that differ in the 7th
    # digit -- the kernels' atomics round differently from run to run -- already flip near-tied decodes.
    import re
    loss = lambda text: [float(v) for v in re.findall(r'train_loss ([0-9.]+)', text)]
    a, b = loss(r1.stdout), loss(r2.stdout)
    assert len(a) == 2 and len(b) == 2, (r1.stdout[-1500:], r2.stdout[-1500:])
    np.testing.assert_allclose(b, a, rtol=1e-6)
    rows = [l for l in r2.stdout.splitlines() if l.strip() and all(_isfloat(t) for t in l.replace(',', ' ').split())]
    assert len(rows) >= 2 and len(rows[-1].split(',')) > 3           # rank 0 printed the reduced train / test statistics


def _isfloat(t):
    try:
        float(t)
        return True
    except ValueError:
        return False
