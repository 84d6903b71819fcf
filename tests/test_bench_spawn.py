"""bench.py's N-rank launch, rehearsed on the host: ``python bench.py --gpus N`` must start N ranks itself (before any
GPU call), fail loudly on a world-size mismatch, and shard ONE corpus so that the reduced counters are the corpus's."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_spawn_two_ranks_dry_run():
    r = run(['--gpus', '2', '--dry-run', '--strong-workload', 'tiny'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['backend'] == 'gloo'
    assert line['frames'] == line['corpus_frames'] and line['batches'] == line['n_batches']   # every video exactly once


def test_spawn_three_ranks_dry_run():
    r = run(['--gpus', '3', '--dry-run', '--strong-workload', 'tiny', '--seed', '7'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 3 and line['ranks_seen'] == 3 and line['frames'] == line['corpus_frames']


def test_world_size_mismatch_fails():
    # a launcher that produced one rank for --gpus 2 (what `python bench.py --gpus 2` used to do silently)
    r = run(['--gpus', '2', '--dry-run'], env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0
    assert 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_more_ranks_than_gpus_fails():
    """The spawning parent does not touch the GPU runtime; the rank without a GPU of its own fails and takes the job down."""
    import torch
    n = torch.cuda.device_count()
    r = run(['--gpus', str(max(n, 1) + 1)])
    assert r.returncode != 0 and 'has no GPU of its own' in r.stderr


def test_strong_scaling_is_the_headline_for_more_than_one_rank():
    """BASELINE config 5 (one corpus sharded by video) is what N > 1 headlines; N = 1 stays the weak (cfg3) workload."""
    r = run(['--gpus', '2', '--dry-run', '--strong-workload', 'tiny'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])['scaling'] == 'strong'
    r = run(['--gpus', '1', '--dry-run', '--strong-workload', 'tiny'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])['scaling'] == 'weak'


def test_share_gpus_needs_gloo():
    r = run(['--gpus', '2', '--share-gpus'])
    assert r.returncode != 0 and 'gloo' in r.stderr
