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
    # workgroups per sample of the attention-step backward: 4 fill the chip at batch 64; inputs above 128 characters get one per 32
    # positions so that they stay on the MFMA kernel (attn_bwd_fast_ok: <= 32 own rows per workgroup) instead of the generic one,
    # whose LDS request also ends at L ~ 310
    nch = int(os.environ.get('MTTS_NCH_BWD', cfg.get('nch_bwd', max(4, (L + 31) // 32))))
    # K-split of the ctx-column input gradient: as many slabs as keep the launch within ONE wave of workgroups (256 CUs)
    ksc = int(os.environ.get('MTTS_KSC', cfg.get('ksb_ctx', max(1, min(8, 256 // ((Dm + 15) // 16))))))
    g.ksb, g.nch, g.ksb_ctx = ksb, nch, ksc
    # every workspace size comes from the library (mtts_decoder_grad_buffer_elems)
    lib().mtts_decoder_grad_buffer_elems.restype = ctypes.c_long

    def n_of(field):
        v = int(lib().mtts_decoder_grad_buffer_elems(ctypes.byref(a), ctypes.byref(g), field.encode()))
        if v < 0:
            raise _C.MttsError(f'mtts_decoder_grad_buffer_elems: unknown field {field}')
        return v
    E = lambda field, *shape: torch.empty(n_of(field), dtype=torch.float32, device=dev).view(*shape) if shape else \
        torch.empty(n_of(field), dtype=torch.float32, device=dev)
    Z = lambda field, *shape: torch.zeros(n_of(field), dtype=torch.float32, device=dev).view(*shape) if shape else \
        torch.zeros(n_of(field), dtype=torch.float32, device=dev)
    buf('dout', dout)
    if dal is not None:
        buf('dalign', dal)
    buf('att_w_rec_T', E('att_w_rec_T'))
    buf('gen_w_hh_T', E('gen_w_hh_T'))
    buf('w_query_T', E('w_query_T'))
    buf('dG_att', E('dG_att'))
    buf('dG_gen', E('dG_gen'))
    if H % 16 == 0 and st.h_att_p is not None:      # MFMA-tile-order copies for the per-step input-gradient GEMMs
        Bp = (B + 15) & ~15
        zp = Z if Bp != B else E                                   # padded batch rows of the packed copies stay zero
        buf('dG_att_p', zp('dG_att_p'))
        buf('dG_gen_p', zp('dG_gen_p'))
        buf('att_w_rec_Tp', E('att_w_rec_Tp'))
        buf('gen_w_hh_Tp', E('gen_w_hh_Tp'))
    if not st.fast:      # general schedule (teacher forcing < 1): per-step chain with transposed full weights
        buf('att_w_ih_T', E('att_w_ih_T'))
        buf('gen_w_ih_T', E('gen_w_ih_T'))
        buf('w_out_T', E('w_out_T'))
        pwt = [E('prenet_w_T0' if i == 0 else 'prenet_w_T') for i in range(n)]
        keep.extend(pwt)
        for i in range(n):
            g.prenet_w_T[i] = pwt[i].data_ptr()
        buf('step_ws', Z('step_ws'))
        buf('frames_fed', E('frames_fed'))
    buf('dHG', E('dHG'))
    buf('dHA', E('dHA'))
    # written slot by slot before they are read; only the boundary slots need clearing
    buf('dctx_all', E('dctx_all', T + 1, B, Dm))[0].zero_()
    buf('dctx_tot', E('dctx_tot', T + 1, B, Dm))[0].zero_()
    buf('dcum_all', Z('dcum_all'))                   # accumulated with atomics by the chunk workgroups
    buf('dq_all', Z('dq_all'))
    buf('part_gen', Z('part_gen'))
    buf('part_att', Z('part_att'))
    if st.fast and 32 < B <= 64 and st.h_att_p is not None and H % 16 == 0 and os.environ.get('MTTS_PBWD', '0') == '1':
        # persistent backward of chain A (csrc/pbwd.hip; opt-in, measured slower than the per-step launches: profiles/r06_pbwd_ab.txt):
        # partial slabs in a ring indexed by the step; the library falls back to the per-step launches for shapes it does not take
        g.part_ring_slots = int(lib().mtts_decoder_bwd_ring_slots())
        buf('part_ring', E('part_ring'))
    for name in ('dc_att', 'dc_gen', 'dh_carry_att', 'dh_carry_gen', 'dMt', 'dU_slab', 'dv_slab', 'dbias_slab'):
        buf(name, Z(name))
    buf('dU', E('dU'))
    buf('dpren', E('dpren'))
    buf('colsum_ws', E('colsum_ws'))
    dmemory = buf('dmemory', E('dmemory', B, L, Dm))
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
    lib().mtts_bilstm_buffer_elems.restype = ctypes.c_long
    n = lambda field: int(lib().mtts_bilstm_buffer_elems(ctypes.byref(a), ksb, field.encode()))          # sizes come from the library
    whT = [_e(n('w_hh_T'), device=dev) for _ in range(2)]
    dxp = [_e(n('dxproj'), device=dev).view(L, B, 4 * H) for _ in range(2)]
    part, dc, dhc = _z(n('part'), device=dev), _z(n('dc'), device=dev), _z(n('dh_carry'), device=dev)
    cws = _e(n('colsum_ws'), device=dev)
    dx = _e(n('dx'), device=dev).view(L, B, Cin)
    dw = [[_e(*t.shape, device=dev) for t in ws[d]] for d in range(2)]
    for d in range(2):
        a.w_ih[d], a.w_hh[d], a.b_ih[d], a.b_hh[d] = (t.data_ptr() for t in ws[d])
        a.h[d], a.c[d], a.gates[d] = hs[d].data_ptr(), cs[d].data_ptr(), gs[d].data_ptr()
        g.w_hh_T[d], g.dxproj[d] = whT[d].data_ptr(), dxp[d].data_ptr()
        g.d_w_ih[d], g.d_w_hh[d], g.d_b_ih[d], g.d_b_hh[d] = (t.data_ptr() for t in dw[d])
    g.dy, g.part, g.ksb, g.dc, g.dh_carry, g.colsum_ws, g.dx = ptr(dy), ptr(part), ksb, ptr(dc), ptr(dhc), ptr(cws), ptr(dx)
    check(lib().mtts_bilstm_bwd(ctypes.byref(a), ctypes.byref(g), stream_ptr()), 'mtts_bilstm_bwd')
    return (dx.transpose(0, 1), None, *dw[0], *dw[1])
