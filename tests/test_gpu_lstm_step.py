"""mtts_lstm_step_fwd (csrc/lstm_step.hip): K-split gate GEMM + (partial sum, LSTM cell, query partials) against a plain
torch fp64 / fp32 restatement of torch.nn.LSTMCell + dropout / zoneout (reference modules/layers.py:18-47) and of the
attention query projection (modules/attention.py:68)."""
import ctypes

import pytest
import torch

from multilingual_text_to_speech_amd import _C
from multilingual_text_to_speech_amd._C import check, lib, ptr, stream_ptr

pytestmark = pytest.mark.gpu


def run_step(B, H, Ks, A, precision, zone=0, seed=0, with_pre=True, with_q=True, nb_max=0, inference=False):
    g = torch.Generator().manual_seed(seed)
    dev = 'cuda'
    xs = [torch.randn(B, K, generator=g).to(dev) for K in Ks]
    ws = [(torch.randn(4 * H, K, generator=g) / sum(Ks) ** 0.5).to(dev) for K in Ks]
    b_ih, b_hh = torch.randn(4 * H, generator=g).to(dev), torch.randn(4 * H, generator=g).to(dev)
    pre = torch.randn(B, 4 * H, generator=g).to(dev) if with_pre else None       # gate-major, like the reference's gates
    h_prev, c_prev = torch.randn(B, H, generator=g).to(dev), torch.randn(B, H, generator=g).to(dev)
    wq = (torch.randn(A, H, generator=g) / H ** 0.5).to(dev)
    hmask = (torch.rand(B, H, generator=g) >= 0.1).to(torch.uint8).to(dev)
    cmask = (torch.rand(B, H, generator=g) >= 0.1).to(torch.uint8).to(dev)
    Kt = sum(Ks)
    L = lib()
    packed = torch.empty(int(L.mtts_lstm_packed_weight_bytes(H, Kt, precision)), dtype=torch.uint8, device=dev)
    bias_u = torch.empty(4 * H, device=dev)
    pk = _C.LstmPackArgs()
    for i, w in enumerate(ws):
        pk.w[i], pk.K[i], pk.ldw[i] = w.data_ptr(), Ks[i], Ks[i]
    pk.nseg, pk.H, pk.precision, pk.dst = len(Ks), H, precision, ptr(packed)
    pk.b_ih, pk.b_hh, pk.bias_u = ptr(b_ih), ptr(b_hh), ptr(bias_u)
    check(L.mtts_lstm_pack_weights(ctypes.byref(pk), stream_ptr()), 'pack')
    pre_u = None
    if pre is not None:      # unit-major columns 4u + g
        pre_u = pre.view(B, 4, H).permute(0, 2, 1).reshape(B, 4 * H).contiguous()
    a = _C.LstmStepArgs()
    for i, x in enumerate(xs):
        a.x[i], a.K[i], a.ldx[i] = x.data_ptr(), Ks[i], Ks[i]
    a.nseg, a.w_packed, a.precision, a.B, a.H, a.nb_max = len(Ks), ptr(packed), precision, B, H, nb_max
    part = torch.full((int(L.mtts_lstm_step_partial_floats(B, H, Kt)),), float('nan'), device=dev)
    h_out, c_out = torch.full((B, H), float('nan'), device=dev), torch.full((B, H), float('nan'), device=dev)
    gates = torch.full((B, 4 * H), float('nan'), device=dev)
    qpart = torch.full((H // 16, B, A), float('nan'), device=dev)
    a.partials, a.pre, a.ldpre, a.bias_u = ptr(part), ptr(pre_u), 4 * H, ptr(bias_u)
    a.h_prev, a.c_prev, a.h_out, a.c_out, a.gates_out = ptr(h_prev), ptr(c_prev), ptr(h_out), ptr(c_out), (None if inference else ptr(gates))
    a.zone = zone
    if zone == 0 and inference:
        pass                                     # eval mode of the dropout cell: no mask, no saved gates
    elif zone == 0:
        a.hmask, a.hscale = ptr(hmask), 1.0 / 0.9
    elif zone == 1:
        a.hmask, a.cmask = ptr(hmask), ptr(cmask)
    else:
        a.zh, a.zc = 0.1, 0.15
    if with_q:
        a.w_query, a.A, a.qpart = ptr(wq), A, ptr(qpart)
    check(L.mtts_lstm_step_fwd(ctypes.byref(a), stream_ptr()), 'lstm_step')
    torch.cuda.synchronize()

    # ---- reference (fp64)
    d = lambda t: t.double()
    if precision == 1:
        rb = lambda t: t.to(torch.bfloat16).double()
        z = sum(rb(x) @ rb(w).t() for x, w in zip(xs, ws))
    else:
        z = sum(d(x) @ d(w).t() for x, w in zip(xs, ws))
    z = z + d(b_ih) + d(b_hh) + (d(pre) if pre is not None else 0)
    i_, f_, g_, o_ = z.view(B, 4, H).unbind(1)
    i_, f_, g_, o_ = torch.sigmoid(i_), torch.sigmoid(f_), torch.tanh(g_), torch.sigmoid(o_)
    cn = f_ * d(c_prev) + i_ * g_
    hn = o_ * torch.tanh(cn)
    if zone == 0 and inference:
        ho, co = hn, cn
    elif zone == 0:
        ho, co = hn * d(hmask) / 0.9, cn
    elif zone == 1:
        ho = torch.where(hmask.bool(), hn, d(h_prev)); co = torch.where(cmask.bool(), cn, d(c_prev))
    else:
        ho, co = 0.1 * d(h_prev) + 0.9 * hn, 0.15 * d(c_prev) + 0.85 * cn
    tol = 2e-5
    assert (h_out.double() - ho).abs().max().item() <= tol, ('h', (h_out.double() - ho).abs().max().item())
    assert (c_out.double() - co).abs().max().item() <= tol * max(1.0, co.abs().max().item())
    ref_gates = torch.stack((i_, f_, g_, o_), 1).reshape(B, 4 * H)
    if not inference:
        assert (gates.double() - ref_gates).abs().max().item() <= tol
    if with_q:
        q = qpart.double().sum(0)
        ref_q = h_out.double() @ d(wq).t()
        assert (q - ref_q).abs().max().item() <= 2e-5 * max(1.0, ref_q.abs().max().item())
    return h_out


@pytest.mark.parametrize('B', [1, 7, 16, 40, 64, 65, 100, 240])
def test_lstm_step_matches_lstm_cell(B):
    """The decoder's shapes (H = 1024, [context | h] operand of both presets) over every row-tile regime: one ragged 16-row tile,
    one 64-row tile, several row tiles with the weights held in registers."""
    run_step(B, 1024, [544, 1024], 128, 0, seed=B)
    run_step(B, 1024, [288, 1024], 128, 0, seed=B + 1)


@pytest.mark.parametrize('Ks,H,A', [([32], 32, 16), ([256, 544, 1024], 1024, 128), ([1024, 288, 1024], 1024, 64), ([96, 64], 64, 48),
                                    ([2592], 128, 128)])
def test_lstm_step_shapes(Ks, H, A):
    """1-3 K segments, k-slices that straddle segment boundaries, K larger than 8 x 7 k-blocks (KS > 8), small H / A."""
    run_step(24, H, Ks, A, 0, seed=3)


@pytest.mark.parametrize('zone', [1, 2])
def test_lstm_step_zoneout(zone):
    run_step(33, 256, [64, 256], 64, 0, zone=zone, seed=11)


@pytest.mark.parametrize('B', [1, 2])
@pytest.mark.parametrize('zone', [0, 2])
def test_lstm_step_one_or_two_rows_at_inference(B, zone):
    """lstm_gemv_kernel: the step of one / two rows without saved gates and training masks (single-utterance synthesis) - both decoder
    LSTMs' operand shapes (three K segments: [prenet | context | h] and [h | context | h]), eval-mode dropout cell and eval-mode zoneout,
    with and without the hoisted addend / the query partials."""
    run_step(B, 1024, [256, 288, 1024], 128, 0, zone=zone, seed=40 + B, inference=True)
    run_step(B, 1024, [1024, 544, 1024], 128, 0, zone=zone, seed=50 + B, inference=True, with_pre=False)
    run_step(B, 256, [64, 256], 64, 0, zone=zone, seed=60 + B, inference=True, with_q=False)


def test_lstm_step_without_addend_and_query():
    run_step(20, 128, [128], 128, 0, with_pre=False, with_q=False, seed=5)


@pytest.mark.parametrize('B', [16, 64, 240])
def test_lstm_step_bf16_operands(B):
    """precision 1: operands rounded to bf16 (RNE), exact products, fp32 accumulation -> equals the fp64 product of the rounded
    operands to fp32 accumulation error."""
    run_step(B, 1024, [288, 1024], 128, 1, seed=B)


def test_fp32_split_products_are_fp32_accurate():
    """Six-term bf16 split vs fp64 on wide-dynamic-range data: error at the level of an fp32 accumulation (cf. the GEMM core test)."""
    g = torch.Generator().manual_seed(1)
    B, H, K = 64, 256, 1024
    dev = 'cuda'
    x = (torch.randn(B, K, generator=g) * torch.exp2(torch.randint(-6, 6, (B, K), generator=g).float())).to(dev)
    w = (torch.randn(4 * H, K, generator=g) * torch.exp2(torch.randint(-6, 6, (4 * H, K), generator=g).float())).to(dev)
    L = lib()
    packed = torch.empty(int(L.mtts_lstm_packed_weight_bytes(H, K, 0)), dtype=torch.uint8, device=dev)
    pk = _C.LstmPackArgs()
    pk.w[0], pk.K[0], pk.ldw[0], pk.nseg, pk.H, pk.precision, pk.dst = w.data_ptr(), K, K, 1, H, 0, ptr(packed)
    check(L.mtts_lstm_pack_weights(ctypes.byref(pk), stream_ptr()), 'pack')
    a = _C.LstmStepArgs()
    a.x[0], a.K[0], a.ldx[0], a.nseg, a.w_packed, a.B, a.H = x.data_ptr(), K, K, 1, ptr(packed), B, H
    KS = int(L.mtts_lstm_step_ksplit(K))
    part = torch.empty(KS, B, 4 * H, device=dev)
    c_prev, h_out, c_out = torch.zeros(B, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    a.partials, a.c_prev, a.h_out, a.c_out = ptr(part), ptr(c_prev), ptr(h_out), ptr(c_out)
    check(L.mtts_lstm_step_fwd(ctypes.byref(a), stream_ptr()), 'lstm_step')
    torch.cuda.synchronize()
    z = part.double().sum(0).view(B, H, 4).permute(0, 2, 1).reshape(B, 4 * H)        # unit-major -> gate-major
    ref = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    err = ((z - ref).abs() / scale).max().item()
    err_torch = (((x @ w.t()).double() - ref).abs() / scale).max().item()
    assert err <= 1.25 * err_torch + 2.0 ** -24, (err, err_torch)


# ---- batches above 64 rows: the fused kernel (lstm_fused_kernel: full-K gate products, cell and query partials in ONE launch) ----------
@pytest.mark.parametrize('B', [96, 128, 129, 200, 256])
def test_fused_large_batch_step_matches_lstm_cell(B):
    """32-row workgroups (B <= 128) and 64-row workgroups (B > 128), ragged last row group, the training operands [context | h] and
    the three-segment operands of the free-running schedule ([prenet | context | h], [h_att | context | h_gen])."""
    run_step(B, 1024, [288, 1024], 128, 0, seed=B)
    run_step(B, 1024, [256, 288, 1024], 128, 0, seed=B + 1)
    run_step(B, 1024, [1024, 288, 1024], 128, 0, seed=B + 2, with_pre=False)


@pytest.mark.parametrize('B,H,Ks,A', [(70, 128, [128], 128), (100, 64, [96, 64], 48), (130, 256, [64, 256], 256), (65, 32, [32], 16)])
def test_fused_large_batch_step_small_widths(B, H, Ks, A):
    """H / 16 not a multiple of 8 (plain workgroup order instead of the XCD-aware one), A up to 256 (two query tiles per wave),
    a single k-block, without the hoisted addend / the query."""
    run_step(B, H, Ks, A, 0, seed=B)
    run_step(B, H, Ks, A, 0, seed=B + 1, with_pre=False, with_q=False)


@pytest.mark.parametrize('zone', [1, 2])
def test_fused_large_batch_step_zoneout(zone):
    run_step(100, 256, [64, 256], 64, 0, zone=zone, seed=13)
    run_step(150, 256, [64, 256], 64, 0, zone=zone, seed=14)


@pytest.mark.parametrize('nb_max', [0, 4])
@pytest.mark.parametrize('B', [100, 128, 200])
def test_fused_large_batch_step_bf16_operands(B, nb_max):
    """nb_max 4: lstm_fused_kernel (two workgroups per CU, what the two concurrent chains of the teacher-forced schedule launch);
    nb_max 0: lstm_fused2_kernel (a lone chain: 128-k blocks, k-quarter x column-pair waves, partial tiles joined through LDS)."""
    run_step(B, 1024, [288, 1024], 128, 1, seed=B, nb_max=nb_max)
    run_step(B, 1024, [256, 288, 1024], 128, 1, seed=B + 1, with_pre=False, nb_max=nb_max)


@pytest.mark.parametrize('B,H,Ks,A', [(96, 1024, [288, 1024], 128), (128, 1024, [256, 288, 1024], 128), (240, 1024, [544, 1024], 128),
                                      (200, 1024, [1024, 288, 1024], 128), (70, 128, [128], 128), (130, 64, [96, 64], 48),
                                      (256, 1024, [32], 128), (129, 256, [160], 64), (65, 32, [32, 32, 32], 16)])
@pytest.mark.parametrize('nb_max', [0, 4])
def test_fused_large_batch_step_presplit_weight_planes(B, H, Ks, A, nb_max):
    """precision 2: fp32 operands with the weights stored as three pre-split bf16 planes and the activations split on their way into
    LDS - the six-term products of the GEMM core (fp32-accurate); what the decoder uses for every fp32 batch above 64 rows.  Both
    kernels (nb_max 4: lstm_fused_kernel, 0: lstm_fused2_kernel); K of one, five (a block with one 32-k quarter past the end) and three
    32-k blocks from three segments; odd and even 128-k block counts."""
    run_step(B, H, Ks, A, 2, seed=B, nb_max=nb_max)
    run_step(B, H, Ks, A, 2, seed=B + 1, with_pre=False, with_q=False, zone=2, nb_max=nb_max)


@pytest.mark.parametrize('precision', [1, 2])
def test_fused2_matches_fused_to_rounding(precision):
    """The two fused kernels on the same operands: same products, different summation trees (k quarters) - equal to fp32 rounding."""
    h0 = run_step(240, 1024, [544, 1024], 128, precision, seed=77, nb_max=0)
    h4 = run_step(240, 1024, [544, 1024], 128, precision, seed=77, nb_max=4)
    assert (h0 - h4).abs().max().item() <= 1e-5


def test_presplit_planes_are_fp32_accurate():
    """The plane form against fp64 on wide-dynamic-range data, next to torch's fp32 product (cf. test_fp32_split_products_are_fp32_accurate)."""
    g = torch.Generator().manual_seed(1)
    B, H, K = 128, 256, 1024
    dev = 'cuda'
    x = (torch.randn(B, K, generator=g) * torch.exp2(torch.randint(-6, 6, (B, K), generator=g).float())).to(dev)
    w = (torch.randn(4 * H, K, generator=g) * torch.exp2(torch.randint(-6, 6, (4 * H, K), generator=g).float())).to(dev)
    L = lib()
    packed = torch.empty(int(L.mtts_lstm_packed_weight_bytes(H, K, 2)), dtype=torch.uint8, device=dev)
    pk = _C.LstmPackArgs()
    pk.w[0], pk.K[0], pk.ldw[0], pk.nseg, pk.H, pk.precision, pk.dst = w.data_ptr(), K, K, 1, H, 2, ptr(packed)
    check(L.mtts_lstm_pack_weights(ctypes.byref(pk), stream_ptr()), 'pack')
    a = _C.LstmStepArgs()
    a.x[0], a.K[0], a.ldx[0], a.nseg, a.w_packed, a.B, a.H, a.precision = x.data_ptr(), K, K, 1, ptr(packed), B, H, 2
    part = torch.empty(int(L.mtts_lstm_step_partial_floats(B, H, K)), device=dev)
    c_prev, h_out, c_out = torch.zeros(B, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    gates = torch.empty(B, 4 * H, device=dev)
    a.partials, a.c_prev, a.h_out, a.c_out, a.gates_out = ptr(part), ptr(c_prev), ptr(h_out), ptr(c_out), ptr(gates)
    check(L.mtts_lstm_step_fwd(ctypes.byref(a), stream_ptr()), 'lstm_step')
    torch.cuda.synchronize()
    # pre-activations back from the saved (activated) input gate: logit(sigmoid(z)) - only where the sigmoid is not saturated
    ref = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    ig, zi = gates[:, :H].double(), ref[:, :H]
    ok = (zi.abs() < 6)
    z_back = torch.log(ig / (1 - ig))
    err = (((z_back - zi).abs() / scale[:, :H])[ok]).max().item()
    err_torch = ((((x @ w.t()).double()[:, :H] - zi).abs() / scale[:, :H])[ok]).max().item()
    assert err <= 4.0 * err_torch + 2.0 ** -20, (err, err_torch)
