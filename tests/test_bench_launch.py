"""`python bench.py --gpus N` must launch its own ranks (the driver's SCALE run passes no launcher): the self-launch,
barrier / max-over-ranks timing and single JSON line, exercised on CPU (gloo) with the stub step (MTTS_BENCH_STUB=1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, argv):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(MTTS_BENCH_STUB='1', MTTS_DIST_BACKEND='gloo', **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout          # exactly ONE JSON line (rank 0)
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    line = _run({}, ['--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '4', '--frames', '10'])
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1
    assert line['config']['global_batch'] == 8 and line['config']['parallelism'] == 'dp2'
    assert line['value'] > 0 and line['scaling'] == 'weak'


def test_bench_single_rank_needs_no_launcher():
    line = _run({}, ['--gpus', '1', '--steps', '2', '--warmup', '0', '--batch', '4', '--frames', '10'])
    assert line['n_gpus'] == 1


def test_bench_rejects_mismatched_world_size():
    env = {k: v for k, v in os.environ.items()}
    env.update(MTTS_BENCH_STUB='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


# ---- SCALE dry run on ONE GPU -----------------------------------------------------------------------------------------------------
import pytest


def _run_real(argv, extra_env, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'MTTS_BENCH_STUB')}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-2500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_scale_dry_run_two_ranks_on_one_gpu_with_the_real_train_step():
    """What the driver's SCALE run does (`python bench.py --gpus N --steps K --warmup W`, no launcher), with the REAL train step:
    self-launch under torch.distributed.run, rendezvous on 127.0.0.1, per-rank model / shard / gradient buckets with the overlapped
    all-reduce, fused clip + Adam, barrier + max-over-ranks timing, ONE JSON line from rank 0.  Both ranks share cuda:0
    (MTTS_SINGLE_DEVICE=1), so the collective backend is gloo (RCCL refuses two ranks on one device) and the persistent kernels are
    off (two processes cannot both own every CU); on the 8-GPU node the same code path runs over RCCL with one device per rank.
    (The two-rank step time itself says nothing here: gloo stages the 114 MB of gradients through the host - seconds per step.)"""
    argv = ['--steps', '2', '--warmup', '1', '--batch', '8', '--frames', '60', '--no-secondary', '--no-cpu-baseline']
    env = dict(MTTS_SINGLE_DEVICE='1', MTTS_DIST_BACKEND='gloo', MTTS_PERSIST='0')
    two = _run_real(['--gpus', '2', *argv], env)
    assert two['n_gpus'] == 2 and two['steps'] == 2 and two['warmup'] == 1 and two['scaling'] == 'weak'
    assert two['config']['global_batch'] == 16 and two['config']['parallelism'] == 'dp2'
    assert two['unit'] == 'mel-frames/s' and two['value'] > 0 and 'roofline' in two
    assert abs(two['value'] - 16 * 60 * 1e3 / two['ms_per_step']) <= 0.02 * two['value']          # whole-job frames / max-over-ranks time
    one = _run_real(['--gpus', '1', *argv], dict(MTTS_PERSIST='0'))
    assert one['n_gpus'] == 1 and one['config']['global_batch'] == 8 and one['value'] > 0
    assert two['ms_per_step'] >= 0.5 * one['ms_per_step']          # two ranks time-share one GPU: never faster than half the single-rank step
