"""Code generation of the latency-bound step kernels (no GPU needed: hipcc cross-compiles gfx950 assembly).

DESIGN.md 3.10: a select or branch on a loaded value makes the compiler wait for the load right behind it, i.e. one memory round
trip per operand; these kernels are written so that their entry requests go out as ONE burst.  The property lives in the compiler's
output, not in the source - so it is pinned here with scripts/asm_wait_scan.py: the number of vector-memory requests issued before
the first `s_waitcnt vmcnt`, and no "short round trip" (a wait that drains the queue after fewer than three requests)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))

HIPCC = shutil.which('hipcc') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else None)

# kernel-name substring -> (source file, minimum requests in the entry burst, maximum short round trips)
KERNELS = {
    'prenet2_kernel': ('lstm_step.hip', 30, 0),                      # 40: x / weight fragments of both layers, biases, keep flags
    'skinny_proj_kernel': ('skinny.hip', 28, 1),                     # 33: bias + 16 chunks x (x, w); the one short trip is the mask byte
    'attn_step_big_kernelILi8': ('attention.hip', 40, 0),            # 47: length, query partials, v / bias / cum, Mt tiles, filter bank
    'attn_step_kernelILi32ELi16ELi4': ('attention.hip', 50, 0),      # 59: + the processed memory
}


@pytest.fixture(scope='module')
def assembly(tmp_path_factory):
    if HIPCC is None:
        pytest.skip('hipcc not available')
    out = {}
    d = tmp_path_factory.mktemp('asm')
    for src in sorted({v[0] for v in KERNELS.values()}):
        dst = os.path.join(d, src + '.s')
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-x', 'hip', '--cuda-device-only', '-S',
                            '-o', dst, os.path.join(ROOT, 'multilingual_text_to_speech_amd', 'csrc', src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[src] = dst
    return out


@pytest.mark.parametrize('kernel', sorted(KERNELS))
def test_entry_requests_go_out_as_one_burst(assembly, kernel):
    import asm_wait_scan as scan
    src, min_burst, max_short = KERNELS[kernel]
    found = [(name, body) for name, body in scan.kernels(assembly[src]) if kernel in name]
    assert len(found) == 1, [n for n, _ in found]
    n_loads, waits0, short, seq = scan.scan(found[0][1])
    assert seq, 'no vector-memory wait at all?'
    assert seq[0][0] >= min_burst, (kernel, 'requests before the first wait', seq[:4])
    assert short <= max_short, (kernel, 'short round trips', short, seq)
