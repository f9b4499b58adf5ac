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
