"""Software-pipelined GEMM core (gemm_pipe_kernel, csrc/gemm.hip): plain fp32 GEMMs with whole 32-wide K blocks run on it by default.
It must (a) stay at fp32-accumulation error against fp64 on every storage variant, ragged M / N, split-K and batched launches and
every epilogue, for plain products and for the three implicit-GEMM forms of a convolution, and (b) return the SAME BITS as
gemm_split_kernel (same products, same accumulation order), which a child process with MTTS_GEMM_PIPE=0 computes."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (M, N, K): K a multiple of 32; ragged M / N; one-tile / long-K shapes that take the split-K path; a single K block
SHAPES = [(333, 517, 1248), (128, 128, 32), (64, 40, 64), (1000, 81, 1024), (256, 384, 4096), (100, 2052, 96), (4, 8, 4096),
          # more than 256 tiles (two rounds of workgroups); odd / even / single K block counts
          (2300, 2052, 96), (2304, 2048, 128), (4096, 1152, 32)]


def _operands(M, N, K, variant, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    A = torch.randn(M, K, device='cuda', generator=g) * torch.exp2(torch.randint(-6, 6, (M, K), device='cuda', generator=g).float())
    B = torch.randn(N, K, device='cuda', generator=g) * torch.exp2(torch.randint(-6, 6, (N, K), device='cuda', generator=g).float())
    tA, tB = variant[0] == 't', variant[1] == 'n'
    return A, B, (A.t().contiguous() if tA else A), (B.t().contiguous() if tB else B), tA, tB


def _run(M, N, K, variant, seed, **epilogue):
    from multilingual_text_to_speech_amd import kernels as Kn
    A, B, a_st, b_st, tA, tB = _operands(M, N, K, variant, seed)
    C = torch.full((M, N), 0.5, device='cuda')
    Kn.gemm(a_st, b_st, C, M, N, K, M if tA else K, N if tB else K, N, transA=tA, transB=tB, **epilogue)
    return A, B, C


def _transposable(M, N, variant):
    # a transposed operand is handed over with its row count as the leading dimension: float4 loads need it to be a multiple of 4
    return (variant[0] != 't' or M % 4 == 0) and (variant[1] != 'n' or N % 4 == 0)


@pytest.mark.parametrize('variant', ['nt', 'nn', 'tt', 'tn'])
@pytest.mark.parametrize('shape', SHAPES)
def test_pipelined_core_is_fp32_accurate(shape, variant):
    M, N, K = shape
    if not _transposable(M, N, variant):
        pytest.skip('leading dimension of the transposed operand is not a multiple of 4: not a shape the library accepts vectorised')
    A, B, C = _run(M, N, K, variant, seed=11)
    ref = A.double() @ B.double().t()
    scale = A.double().abs() @ B.double().abs().t()
    err = ((C.double() - ref).abs() / scale).max().item()
    err_torch = (((A @ B.t()).double() - ref).abs() / scale).max().item()
    # the maximum over a few hundred thousand outputs fluctuates by tens of percent between two correct fp32 summation orders
    assert err <= 2.0 * err_torch + 2.0 ** -24, f'{variant} {shape}: {err:.3e} (torch fp32: {err_torch:.3e})'


def test_pipelined_core_epilogue_and_batch():
    """alpha / beta / bias / activation / keep-mask epilogue and a grid.z batch (b_z, c_z strides) on the pipelined core."""
    from multilingual_text_to_speech_amd import kernels as Kn
    M, N, K = 200, 136, 256
    A, B, a_st, b_st, _, _ = _operands(M, N, K, 'nt', 5)
    bias = torch.randn(N, device='cuda')
    mask = (torch.rand(M, N, device='cuda') > 0.3).to(torch.uint8)
    C0 = torch.randn(M, N, device='cuda')
    C = C0.clone()
    Kn.gemm(a_st, b_st, C, M, N, K, K, K, N, alpha=0.5, beta=2.0, bias=bias, act=1, mask=mask, mask_scale=1.25)
    ref = torch.relu(0.5 * (A.double() @ B.double().t()) + bias.double() + 2.0 * C0.double()) * mask.double() * 1.25
    assert (C.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    # batch of 3 independent products through grid.z
    G = 3
    Ab = torch.randn(G, M, K, device='cuda')
    Bb = torch.randn(G, N, K, device='cuda')
    Cb = torch.empty(G, M, N, device='cuda')
    Kn.gemm(Ab, Bb, Cb, M, N, K, K, K, N, batch=G, a_z=M * K, b_z=N * K, c_z=M * N)
    refb = torch.matmul(Ab.double(), Bb.double().transpose(1, 2))
    assert (Cb.double() - refb).abs().max().item() <= 1e-4 * refb.abs().max().item()


# (groups, Cg, Og, k, dilation, N, L): channel counts multiples of 32 and N * L a multiple of 32 put the forward, input-gradient and
# weight-gradient GEMMs of a convolution on the pipelined core (CONV 1, 2, 3); L < 32 keeps the weight gradient on the old core
CONV_CASES = [(1, 64, 96, 5, 1, 4, 40), (2, 32, 64, 3, 3, 2, 48), (1, 128, 32, 4, 1, 3, 32), (1, 64, 64, 31, 1, 2, 64),
              (5, 32, 32, 5, 1, 1, 96), (1, 32, 32, 5, 1, 8, 20), (1, 512, 512, 5, 1, 2, 80)]


def _conv_case(case, seed):
    from multilingual_text_to_speech_amd import kernels as Kn
    groups, Cg, Og, k, dil, N_, L = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N_, L, Cg * groups, generator=g).cuda()
    w = (torch.randn(Og * groups, Cg, k, generator=g) / (Cg * k) ** 0.5).cuda()
    dy = torch.randn(N_, L, Og * groups, generator=g).cuda()
    wp = Kn.pack_conv_weight(w)
    y = Kn.conv1d_fwd(x, wp, k, dil, groups)
    dx, dwp = Kn.conv1d_bwd(x, wp, dy, k, dil, groups)
    return x, w, dy, y, dx, Kn.unpack_conv_weight(dwp, Og * groups, Cg, k)


@pytest.mark.parametrize('case', CONV_CASES)
def test_pipelined_core_convolutions_match_torch(case):
    groups, Cg, Og, k, dil, N_, L = case
    x, w, dy, y, dx, dw = _conv_case(case, 7)
    xr = x.double().transpose(1, 2).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    pl = (k - 1) * dil // 2
    yr = torch.nn.functional.conv1d(torch.nn.functional.pad(xr, (pl, (k - 1) * dil - pl)), wr, dilation=dil, groups=groups)
    yr.backward(dy.double().transpose(1, 2))
    tol = lambda ref: 2e-5 * max(1.0, ref.abs().max().item())
    assert (y.double() - yr.transpose(1, 2)).abs().max().item() <= tol(yr), case
    assert (dx.double() - xr.grad.transpose(1, 2)).abs().max().item() <= tol(xr.grad), case
    assert (dw.double() - wr.grad).abs().max().item() <= tol(wr.grad) * (N_ * L) ** 0.5, case


_CHILD = r'''
import sys, torch
sys.path.insert(0, %(root)r)
from tests.test_gpu_gemm_pipe import SHAPES, CONV_CASES, _run, _transposable, _conv_case
out = {}
for shape in SHAPES:
    for variant in ('nt', 'nn', 'tt', 'tn'):
        if _transposable(shape[0], shape[1], variant):
            out[(shape, variant)] = _run(*shape, variant, seed=23)[2].cpu()
for case in CONV_CASES:
    for name, t in zip(('y', 'dx', 'dw'), _conv_case(case, 29)[3:]):
        out[(case, name)] = t.cpu()
from multilingual_text_to_speech_amd import _C
out['planes_launches'] = int(_C.lib().mtts_gemm_planes_count())
torch.save(out, sys.argv[1])
'''


def test_pipelined_core_returns_the_bits_of_the_phase_alternating_core(tmp_path):
    """... and so does the core on PRE-SPLIT operands (round 5, csrc/gemm_planes.h: pack pass + gemm_planes_kernel), which a third
    child runs on every plain shape (MTTS_PLANES_MIN_GFLOP=0 lifts its size threshold): the same exact split, the same six terms in
    the same order, the same K order and split-K partition."""
    outs = {}
    for mode, env_add in (('pipe', {'MTTS_GEMM_PIPE': '1', 'MTTS_GEMM_PLANES': '0'}), ('split', {'MTTS_GEMM_PIPE': '0', 'MTTS_GEMM_PLANES': '0'}),
                          ('planes', {'MTTS_PLANES_MIN_GFLOP': '0', 'MTTS_GEMM_PLANES': '2'})):
        path = str(tmp_path / f'gemm_{mode}.pt')
        env = dict(os.environ, **env_add)
        r = subprocess.run([sys.executable, '-c', _CHILD % {'root': ROOT}, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[mode] = torch.load(path)
    assert outs['pipe'].keys() == outs['split'].keys() == outs['planes'].keys() and len(outs['pipe']) >= 40
    assert outs['planes']['planes_launches'] >= 24 and outs['pipe']['planes_launches'] == 0
    for key, c in outs['split'].items():
        if key == 'planes_launches':
            continue
        assert torch.equal(outs['pipe'][key], c), f'{key}: pipelined and phase-alternating cores differ'
        assert torch.equal(outs['planes'][key], c), f'{key}: pre-split and phase-alternating cores differ'


# production shapes at the DEFAULT size threshold (the hoisted decoder projections, a per-chunk weight gradient, an input gradient), K
# that is not a multiple of 32 (zero-filled tail of the last record), an odd row count, every storage variant
PLANES_SHAPES = [((38400, 4096, 1568), 'nt'), ((4096, 1568, 3072), 'tn'), ((3072, 1568, 4096), 'nn'), ((4096, 1024, 3072), 'tt'),
                 ((3077, 2052, 1000), 'nt'), ((2052, 1000, 3100), 'tn'), ((38400, 81, 1568), 'nt')]

_CHILD_PLANES = r'''
import sys, torch
sys.path.insert(0, %(root)r)
from tests.test_gpu_gemm_pipe import PLANES_SHAPES, _run
from multilingual_text_to_speech_amd import _C
out = {}
for shape, variant in PLANES_SHAPES:
    bias = torch.randn(shape[1], device='cuda', generator=torch.Generator(device='cuda').manual_seed(3))
    out[(shape, variant)] = _run(*shape, variant, seed=31, alpha=0.75, beta=0.5, bias=bias, act=2)[2].cpu()
out['planes_launches'] = int(_C.lib().mtts_gemm_planes_count())
torch.save(out, sys.argv[1])
'''


def test_presplit_core_at_production_shapes_returns_the_bits_of_the_pipelined_core(tmp_path):
    outs = {}
    for mode, env_add in (('planes', {'MTTS_GEMM_PLANES': '2'}), ('pipe', {'MTTS_GEMM_PLANES': '0'})):
        path = str(tmp_path / f'gemm_{mode}.pt')
        r = subprocess.run([sys.executable, '-c', _CHILD_PLANES % {'root': ROOT}, path], env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[mode] = torch.load(path)
    assert outs['planes']['planes_launches'] >= len(PLANES_SHAPES) - 1 and outs['pipe']['planes_launches'] == 0, outs['planes']['planes_launches']
    for key, c in outs['pipe'].items():
        if key != 'planes_launches':
            assert torch.equal(outs['planes'][key], c), f'{key}: pre-split core differs from the pipelined / phase-alternating core'


@pytest.mark.parametrize('shape,variant', [((4096, 4096, 1568), 'nt'), ((4096, 1568, 3072), 'tn'), ((1000, 2052, 3100), 'tt'), ((3077, 1028, 1000), 'nn')])
def test_presplit_core_bf16_equals_the_fp64_product_of_the_rounded_operands(shape, variant, monkeypatch):
    """bf16 mode of the pre-split core (one RNE-rounded plane, records of three K blocks): fp32 accumulation of exact bf16 x bf16
    products - as close to the fp64 product of the rounded operands as an fp32 matmul of the same rounded operands is (the error left
    is the fp32 summation's, which grows with K: 6e-7 * sum |a b| at K = 1568)."""
    from multilingual_text_to_speech_amd import _C
    M, N, K = shape
    before = int(_C.lib().mtts_gemm_planes_count())
    _C.set_precision('bf16')
    try:
        A, B, C = _run(M, N, K, variant, seed=17)
    finally:
        _C.set_precision('fp32')
    assert int(_C.lib().mtts_gemm_planes_count()) == before + 1
    Ar, Br = A.to(torch.bfloat16).double(), B.to(torch.bfloat16).double()
    ref = Ar @ Br.t()
    scale = Ar.abs() @ Br.abs().t()
    err = ((C.double() - ref).abs() / scale).max().item()
    err_torch = (((A.to(torch.bfloat16).float() @ B.to(torch.bfloat16).float().t()).double() - ref).abs() / scale).max().item()
    assert err <= 2.0 * err_torch + 2.0 ** -24, f'{variant} {shape}: {err:.3e} (torch fp32 on the rounded operands: {err_torch:.3e})'
    unrounded = ((C.double() - A.double() @ B.double().t()).abs() / scale).max().item()
    assert unrounded > 1e-4, 'operands were not rounded to bf16: this is not the bf16 path'


