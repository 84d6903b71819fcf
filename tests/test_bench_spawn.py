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


def test_one_headline_workload_for_every_rank_count():
    """VERDICT r4 (weak #11): ``value`` must not change workload with N -- a 1 -> 8 curve computed from the per-N lines would
    otherwise divide cfg5 numbers by a cfg3 anchor.  The headline leg (``scaling``) and its ``config.workload`` string are
    the same for --gpus 1, 2 and 3; the strong leg is named beside it at every N; ``--scaling strong`` flips the headline
    for every N alike."""
    lines = {}
    for n in (1, 2, 3):
        r = run(['--gpus', str(n), '--dry-run', '--workload', 'tiny', '--strong-workload', 'tiny'])
        assert r.returncode == 0, r.stderr[-2000:]
        lines[n] = json.loads(r.stdout.strip().splitlines()[-1])
    assert {l['scaling'] for l in lines.values()} == {'weak'}
    assert len({l['config']['workload'] for l in lines.values()}) == 1
    assert lines[1]['config']['workload'].startswith('tiny seed 2:')
    assert len({l['strong_scaling']['workload'] for l in lines.values()}) == 1
    strong = {}
    for n in (1, 2):
        r = run(['--gpus', str(n), '--dry-run', '--workload', 'tiny', '--strong-workload', 'tiny', '--scaling', 'strong'])
        assert r.returncode == 0, r.stderr[-2000:]
        strong[n] = json.loads(r.stdout.strip().splitlines()[-1])
    assert {l['scaling'] for l in strong.values()} == {'strong'}
    assert strong[1]['config']['workload'] == strong[2]['config']['workload'] == lines[1]['strong_scaling']['workload']


def test_share_gpus_needs_gloo():
    r = run(['--gpus', '2', '--share-gpus'])
    assert r.returncode != 0 and 'gloo' in r.stderr
