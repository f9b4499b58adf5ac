"""Which (batch, length) combinations of the long-input train step lose gradients against the oracle?  (round 5 debugging aid)
    python scripts/dbg_long_inputs.py B,L,T [B,L,T ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tests.test_gpu_more import run_train_step_case

for spec in sys.argv[1:]:
    parts = spec.split(',')
    B, L, T = (int(x) for x in parts[:3])
    seed = int(parts[3]) if len(parts) > 3 else 9
    try:
        run_train_step_case('shared_training', B, L, T, {}, seed=seed)
        print(f'B={B} L={L} T={T} seed={seed}: ok', flush=True)
    except AssertionError as e:
        print(f'B={B} L={L} T={T} seed={seed}: FAIL {str(e)[:700]}', flush=True)
