"""The RCCL ('nccl' backend) code path on the ONE GPU of the test box (it had only ever run over gloo): a
fresh child process brings up a one-rank nccl group and pushes every collective of the N-rank path through it -- counter
all-reduce (SUM / MAX, device and host tensors), parameter broadcast (float and bool), the flat gradient all-reduce, the
reduced evaluation of a sharded predict, a data-parallel training epoch -- and bench.py's strong-scaling leg (BASELINE
config 5: one corpus sharded by video, evaluation counters all-reduced) runs over the same group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, SMM_DIST_SINGLE_RANK='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_every_collective_runs_through_rccl_on_one_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_single_rank.py')], env=_env(), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1])
    assert line['backend'] == 'nccl' and line['world'] == 1 and 0.0 < line['mof'] <= 1.0


def test_bench_strong_leg_reduces_over_rccl():
    common = ['--workload', 'tiny', '--strong-workload', 'tiny', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
              '--no-predict-e2e', '--seed', '5', '--gpus', '1', '--strong-leg']
    bench = os.path.join(ROOT, 'bench.py')
    r = subprocess.run([sys.executable, bench] + common, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    over = json.loads(r.stdout.strip().splitlines()[-1])
    assert over['backend'] == 'nccl' and over['strong_scaling']['stats_reduced_over'].startswith('RCCL all-reduce')
    env = _env()
    env.pop('SMM_DIST_SINGLE_RANK')
    r = subprocess.run([sys.executable, bench] + common, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    plain = json.loads(r.stdout.strip().splitlines()[-1])
    assert plain['backend'] is None and plain['strong_scaling']['stats_reduced_over'] == '1 rank'
    assert over['strong_scaling']['stats'] == plain['strong_scaling']['stats']
    assert over['strong_scaling']['frames'] == plain['strong_scaling']['frames']
