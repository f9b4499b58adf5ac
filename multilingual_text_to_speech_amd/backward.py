"""Backward passes of the recurrent blocks: argument packing for mtts_decoder_bwd / mtts_bilstm_bwd."""
import ctypes

import torch

from . import _C
from ._C import check, lib, ptr, stream_ptr
from .decoder_ops import WEIGHT_ORDER, fill_decoder_args


def _z(*shape, device):
    return torch.zeros(*shape, dtype=torch.float32, device=device)


def _e(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def decoder_bwd(ctx, dspec, dstop, dalign):
    st, w, masks, cfg = ctx.st, ctx.w, ctx.masks, ctx.cfg
    M, P, H, A, Dm, ksz, C = st.dims
    B, L, T, Mo, n = st.B, st.L, st.T, st.Mo, ctx.n_prenet
    dev = ctx.memory.device

    # gradient of (frame, stop) per step, time-major with the forward's row stride
    dout = _z(T + 1, B, Mo, device=dev)
    if dspec is not None:
        dout[1:, :, :M] = dspec.transpose(0, 1)
    if dstop is not None:
        dout[1:, :, M] = dstop.transpose(0, 1)
    dal = dalign.transpose(0, 1).contiguous() if dalign is not None else None

    a = _C.DecoderArgs()
    th = (ctypes.c_uint8 * T)(*[int(x) for x in ctx.teacher])
    fill_decoder_args(a, st, w, ctx.memory, ctx.lengths32, ctx.frames_in, th, masks, cfg)
    a.t0, a.t1 = 0, T

    g = _C.DecoderGradArgs()
    keep = []

    def buf(name, t):
        keep.append(t)
        setattr(g, name, ptr(t))
        return t

    import os
    ksb = int(os.environ.get('MTTS_KSB', cfg.get('ksb', 4)))                 # tuning knobs (scripts/sweep_bwd.sh)
    nch = int(os.environ.get('MTTS_NCH_BWD', cfg.get('nch_bwd', 4)))
    buf('dout', dout)
    if dal is not None:
        buf('dalign', dal)
    buf('att_w_rec_T', _e(Dm + H, 4 * H, device=dev))
    buf('gen_w_hh_T', _e(H, 4 * H, device=dev))
    buf('w_query_T', _e(H, A, device=dev))
    buf('dG_att', _e(T, B, 4 * H, device=dev))
    buf('dG_gen', _e(T, B, 4 * H, device=dev))
    if H % 16 == 0 and st.h_att_p is not None:      # MFMA-tile-order copies for the per-step input-gradient GEMMs
        Bp = (B + 15) & ~15
        zp = _z if Bp != B else _e                                 # padded batch rows of the packed copies stay zero
        buf('dG_att_p', zp(T, Bp * 4 * H, device=dev))
        buf('dG_gen_p', zp(T, Bp * 4 * H, device=dev))
        buf('att_w_rec_Tp', _e(((Dm + H + 15) & ~15) * 4 * H, device=dev))
        buf('gen_w_hh_Tp', _e(H * 4 * H, device=dev))
    if st.fast and H % 32 == 0 and Dm % 4 == 0 and B <= 64 and os.environ.get('MTTS_GBWD', '0') == '1':
        # experiment (off by default, see csrc/decoder_bwd.hip): K-split input-gradient product of chain A
        buf('att_w_rec_T2p', torch.empty(int(lib().mtts_ksplit_packed_weight_bytes(Dm + H, 4 * H, 0)), dtype=torch.uint8, device=dev))
        buf('part_rec', _e(24 * B * (Dm + H), device=dev))
        buf('dh_rec_sum', _e(B, H, device=dev))
    if not st.fast:      # general schedule (teacher forcing < 1): per-step chain with transposed full weights
        buf('att_w_ih_T', _e(P + Dm + H, 4 * H, device=dev))
        buf('gen_w_ih_T', _e(2 * H + Dm, 4 * H, device=dev))
        buf('w_out_T', _e(H + Dm, Mo, device=dev))
        pwt = [_e(M if i == 0 else P, P, device=dev) for i in range(n)]
        keep.extend(pwt)
        for i in range(n):
            g.prenet_w_T[i] = pwt[i].data_ptr()
        buf('step_ws', _z(B * ((P + Dm + H) + (2 * H + Dm) + (H + Dm) + M), device=dev))
        buf('frames_fed', _e(T, B, M, device=dev))
    buf('dHG', _e(T, B, H, device=dev))
    buf('dHA', _e(T, B, H, device=dev))
    # written slot by slot before they are read; only the boundary slots need clearing
    buf('dctx_all', _e(T + 1, B, Dm, device=dev))[0].zero_()
    buf('dctx_tot', _e(T + 1, B, Dm, device=dev))[0].zero_()
    buf('dcum_all', _z(T + 1, B, L, device=dev))                   # accumulated with atomics by the chunk workgroups
    buf('dq_all', _z(T, B, A, device=dev))
    buf('part_gen', _z(ksb, B, H, device=dev))
    # K-split of the ctx-column input gradient: as many slabs as keep the launch within ONE wave of workgroups (256 CUs)
    ksc = int(os.environ.get('MTTS_KSC', cfg.get('ksb_ctx', max(1, min(8, 256 // ((Dm + 15) // 16))))))
    buf('part_att', _z(ksc * B * Dm + ksb * B * H, device=dev))
    g.ksb, g.nch, g.ksb_ctx = ksb, nch, ksc
    buf('dc_att', _z(2, B, H, device=dev))
    buf('dc_gen', _z(2, B, H, device=dev))
    buf('dh_carry_att', _z(2, B, H, device=dev))
    buf('dh_carry_gen', _z(2, B, H, device=dev))
    buf('dMt', _z(B, L, A, device=dev))
    buf('dU_slab', _z(B * nch, A * ksz, device=dev))
    buf('dv_slab', _z(B * nch, A, device=dev))
    buf('dbias_slab', _z(B * nch, A, device=dev))
    buf('dU', _e(A, ksz, device=dev))
    buf('dpren', _e(n, T, B, P, device=dev))
    buf('colsum_ws', _e(int(lib().mtts_colsum_workspace_floats(max(4 * H, A * ksz, P, M + 1))), device=dev))
    dmemory = buf('dmemory', _e(B, L, Dm, device=dev))
    dpw = [_e(*w['prenet_w'][i].shape, device=dev) for i in range(n)]
    dpb = [_e(*w['prenet_b'][i].shape, device=dev) for i in range(n)]
    for i in range(n):
        g.d_prenet_w[i], g.d_prenet_b[i] = dpw[i].data_ptr(), dpb[i].data_ptr()
    grads = {}
    for name in WEIGHT_ORDER:
        grads[name] = buf('d_' + name, _e(*w[name].shape, device=dev))

    check(lib().mtts_decoder_bwd(ctypes.byref(a), ctypes.byref(g), stream_ptr()), 'mtts_decoder_bwd')

    flat = []
    for i in range(n):
        flat += [dpw[i], dpb[i]]
    flat += [grads[k] for k in WEIGHT_ORDER]
    # forward signature: (memory, target, lengths, teacher, masks, cfg, n_prenet, *flat)
    return (dmemory, None, None, None, None, None, None, *flat)


def bilstm_bwd(ctx, dy):
    (x_tm, lengths32, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, h0, h1, c0, c1, g0, g1) = ctx.saved_tensors
    L, B, Cin = x_tm.shape
    H = w_hh.shape[1]
    dev = x_tm.device
    dy = dy.contiguous()
    a = _C.BiLstmArgs()
    a.B, a.L, a.Cin, a.H = B, L, Cin, H
    a.x, a.lengths = ptr(x_tm), ptr(lengths32)
    ws = [(w_ih, w_hh, b_ih, b_hh), (w_ih_r, w_hh_r, b_ih_r, b_hh_r)]
    hs, cs, gs = [h0, h1], [c0, c1], [g0, g1]
    g = _C.BiLstmGradArgs()
    ksb = 4
    whT = [_e(H, 4 * H, device=dev) for _ in range(2)]
    dxp = [_e(L, B, 4 * H, device=dev) for _ in range(2)]
    part, dc, dhc = _z(2, ksb, B, H, device=dev), _z(2, 2, B, H, device=dev), _z(2, 2, B, H, device=dev)
    cws = _e(int(lib().mtts_colsum_workspace_floats(4 * H)), device=dev)
    dx = _e(L, B, Cin, device=dev)
    dw = [[_e(*t.shape, device=dev) for t in ws[d]] for d in range(2)]
    for d in range(2):
        a.w_ih[d], a.w_hh[d], a.b_ih[d], a.b_hh[d] = (t.data_ptr() for t in ws[d])
        a.h[d], a.c[d], a.gates[d] = hs[d].data_ptr(), cs[d].data_ptr(), gs[d].data_ptr()
        g.w_hh_T[d], g.dxproj[d] = whT[d].data_ptr(), dxp[d].data_ptr()
        g.d_w_ih[d], g.d_w_hh[d], g.d_b_ih[d], g.d_b_hh[d] = (t.data_ptr() for t in dw[d])
    g.dy, g.part, g.ksb, g.dc, g.dh_carry, g.colsum_ws, g.dx = ptr(dy), ptr(part), ksb, ptr(dc), ptr(dhc), ptr(cws), ptr(dx)
    check(lib().mtts_bilstm_bwd(ctypes.byref(a), ctypes.byref(g), stream_ptr()), 'mtts_bilstm_bwd')
    return (dx.transpose(0, 1), None, *dw[0], *dw[1])
