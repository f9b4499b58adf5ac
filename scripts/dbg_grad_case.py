"""Debug helper: run one gradient-parity case of tests.test_gpu_more.run_train_step_case and print the verdict (no pytest):
python scripts/dbg_grad_case.py PRESET B L T"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
from tests.test_gpu_more import run_train_step_case
preset, B, L, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 9
try:
    run_train_step_case(preset, B, L, T, {}, seed=seed)
    print(f'{preset} B={B} L={L} T={T} seed={seed} env={ {k: v for k, v in os.environ.items() if k.startswith("MTTS_")} }: OK')
except AssertionError as e:
    print(f'{preset} B={B} L={L} T={T} seed={seed} env={ {k: v for k, v in os.environ.items() if k.startswith("MTTS_")} }: FAIL {str(e)[:1500]}')
