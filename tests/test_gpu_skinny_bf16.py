"""bf16 pair tiles of the per-step backward products (round 5; csrc/skinny_body.h PK = 3, mtts_pack_weight_bf16,
SkinnyArgs.dg_pack_bf16) through the C ABI: the product kernel against the fp64 product of the RNE-rounded operands, and the cell
backward's bf16 copy of the gate gradients against its own fp32 output (reference autograd of LSTMCell, modules/layers.py:18-47,
call sites modules/tacotron2.py:185,188)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pack_bf16(x, rows, K):
    from multilingual_text_to_speech_amd._C import check, lib, ptr, stream_ptr
    rp = (rows + 15) & ~15
    dst = torch.zeros(rp * K // 2, dtype=torch.float32, device='cuda')       # rp * K bf16
    check(lib().mtts_pack_weight_bf16(ptr(x), K, rows, K, ptr(dst), stream_ptr()), 'pack_weight_bf16')
    return dst


@pytest.mark.parametrize('B,N,K,ks', [(64, 1024, 4096, 4), (40, 544, 4096, 7), (16, 1024, 4096, 4), (80, 288, 4096, 2), (5, 48, 64, 1)])
def test_bf16_pair_tile_product_equals_the_fp64_product_of_the_rounded_operands(B, N, K, ks):
    from multilingual_text_to_speech_amd import _C
    from multilingual_text_to_speech_amd._C import check, lib, ptr, stream_ptr
    g = torch.Generator(device='cuda').manual_seed(B + N)
    X = torch.randn(B, K, device='cuda', generator=g) * torch.exp2(torch.randint(-4, 4, (B, K), device='cuda', generator=g).float())
    W = torch.randn(N, K, device='cuda', generator=g) * 0.1
    xp, wp = _pack_bf16(X, B, K), _pack_bf16(W, N, K)
    a = _C.SkinnyArgs()
    a.nseg, a.B, a.N, a.ksplit = 1, B, N, ks
    a.seg[0].x, a.seg[0].w, a.seg[0].K, a.seg[0].ldx, a.seg[0].ldw, a.seg[0].xpack, a.seg[0].wpack = ptr(xp), ptr(wp), K, K, K, 2, 2
    out = torch.zeros(ks, B, N, device='cuda')
    a.out, a.ldo, a.out_ks = ptr(out), N, (B * N if ks > 1 else 0)
    check(lib().mtts_skinny_gemm(ctypes.byref(a), stream_ptr()), 'skinny')
    got = out.sum(0).double()
    Xr, Wr = X.to(torch.bfloat16).double(), W.to(torch.bfloat16).double()
    ref = Xr @ Wr.t()
    scale = Xr.abs() @ Wr.abs().t()
    err = ((got - ref).abs() / scale).max().item()
    assert err <= 2e-6, f'{err:.3e}'                                        # fp32 accumulation of exact bf16 x bf16 products
    assert ((got - X.double() @ W.double().t()).abs() / scale).max().item() > 1e-4      # (the operands WERE rounded)


def test_mixing_bf16_pair_tiles_with_other_operand_forms_is_refused():
    from multilingual_text_to_speech_amd import _C
    from multilingual_text_to_speech_amd._C import lib, ptr, stream_ptr
    X, W = torch.randn(16, 64, device='cuda'), torch.randn(16, 64, device='cuda')
    out = torch.zeros(16, 16, device='cuda')
    a = _C.SkinnyArgs()
    a.nseg, a.B, a.N, a.ksplit = 1, 16, 16, 1
    a.seg[0].x, a.seg[0].w, a.seg[0].K, a.seg[0].ldx, a.seg[0].ldw, a.seg[0].xpack, a.seg[0].wpack = ptr(X), ptr(W), 64, 64, 64, 2, 0
    a.out, a.ldo = ptr(out), 16
    assert lib().mtts_skinny_gemm(ctypes.byref(a), stream_ptr()) != 0


@pytest.mark.parametrize('B', [64, 40, 7])
def test_cell_backward_bf16_copy_of_the_gate_gradients(B):
    """lstm == 2 with dg_pack_bf16: the packed copy holds the RNE-rounded values of dgates_out in the pair-tile order the product kernel
    reads - checked by multiplying it with an identity-like weight (tile order in == tile order out)."""
    from multilingual_text_to_speech_amd import _C
    from multilingual_text_to_speech_amd._C import check, lib, ptr, stream_ptr
    H = 128
    g = torch.Generator(device='cuda').manual_seed(B)
    R = lambda *s: torch.randn(*s, device='cuda', generator=g)
    bufs = dict(dh_a=R(B, H), gates=torch.rand(B, 4 * H, device='cuda', generator=g), c_prev=R(B, H), dc_in=R(B, H), dc_out=R(B, H),
                dgates_out=R(B, 4 * H), pack=torch.zeros(((B + 15) & ~15) * 4 * H, device='cuda'))
    a = _C.SkinnyArgs()
    a.B, a.H, a.lstm, a.ksplit, a.N = B, H, 2, 1, H
    a.dh_a, a.ld_dh_a = ptr(bufs['dh_a']), H
    a.gates, a.c_prev, a.dc_in, a.dc_out = ptr(bufs['gates']), ptr(bufs['c_prev']), ptr(bufs['dc_in']), ptr(bufs['dc_out'])
    a.dgates_out, a.ld_dgates = ptr(bufs['dgates_out']), 4 * H
    a.dg_pack_out, a.dg_pack_bf16 = ptr(bufs['pack']), 1
    check(lib().mtts_skinny_gemm(ctypes.byref(a), stream_ptr()), 'cell backward')
    # product of the packed copy with the identity [4H, 4H] (exact in bf16) returns the rounded gate gradients
    eye = torch.eye(4 * H, device='cuda')
    wp = _pack_bf16(eye, 4 * H, 4 * H)
    q = _C.SkinnyArgs()
    q.nseg, q.B, q.N, q.ksplit = 1, B, 4 * H, 1
    q.seg[0].x, q.seg[0].w, q.seg[0].K, q.seg[0].ldx, q.seg[0].ldw, q.seg[0].xpack, q.seg[0].wpack = ptr(bufs['pack']), ptr(wp), 4 * H, 4 * H, 4 * H, 2, 2
    out = torch.zeros(B, 4 * H, device='cuda')
    q.out, q.ldo = ptr(out), 4 * H
    check(lib().mtts_skinny_gemm(ctypes.byref(q), stream_ptr()), 'skinny')
    assert torch.equal(out, bufs['dgates_out'].to(torch.bfloat16).float())
