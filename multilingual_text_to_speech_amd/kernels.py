"""Thin Python wrappers over the C-ABI: argument packing + torch.autograd glue.

Every compute step goes through libmtts_hip.so; torch is used for device memory, streams and
autograd bookkeeping only.  Activations are channel-last ([N, L, C]).
"""
import ctypes
import os

import torch

from . import _C
from ._C import check, lib, ptr, require_gpu, stream_ptr

ACT = {'identity': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3}


def _f32(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def keep_mask(shape, p, device, generator=None):
    """uint8 keep flags for dropout probability p (1 = keep)."""
    return (torch.rand(shape, device=device, generator=generator) >= p).to(torch.uint8)


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------

def gemm(A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, alpha=1.0, beta=0.0, bias=None, act=0,
         mask=None, mask_scale=1.0, **kw):
    """C = epilogue(alpha * A(m,k) B(n,k) + beta C).  Extra GemmArgs fields via **kw (conv / batch modes)."""
    _C.ensure_workspace(C.device)
    g = _C.GemmArgs()
    g.A, g.B, g.C, g.bias, g.mask = ptr(A), ptr(B), ptr(C), ptr(bias), ptr(mask)
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, lda, ldb, ldc
    g.ldmask = kw.pop('ldmask', N)
    g.transA, g.transB = int(transA), int(transB)
    g.taps, g.Kc, g.batch, g.zt = 1, K, 1, 1
    g.alpha, g.beta, g.act, g.mask_scale = alpha, beta, act, mask_scale
    for k, v in kw.items():
        setattr(g, k, v)
    check(lib().mtts_gemm_ex(ctypes.byref(g), stream_ptr()), 'mtts_gemm_ex')
    return C


def colsum(x2):
    """Column sums of a contiguous [R, C] matrix (deterministic two-stage reduction, mtts_colsum): bias gradients."""
    R, C = x2.shape
    out = _f32(C, device=x2.device)
    ws = _f32(int(lib().mtts_colsum_workspace_floats(C)), device=x2.device)
    check(lib().mtts_colsum(ptr(x2), ptr(out), R, C, C, ptr(ws), stream_ptr()), 'mtts_colsum')
    return out


def linear_fwd(x, weight, bias=None, act=0, mask=None, mask_scale=1.0):
    """y = act(x W^T + b) [* dropout]; x [..., K] contiguous, weight [N, K] (torch Linear layout)."""
    require_gpu(x, weight)
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K)
    y = _f32(x2.shape[0], N, device=x.device)
    gemm(x2, weight, y, x2.shape[0], N, K, K, K, N, bias=bias, act=act, mask=mask, mask_scale=mask_scale)
    return y.reshape(*x.shape[:-1], N)


def linear_bwd(x, weight, dy, need_dx=True):
    """Gradients of y = x W^T + b given dy (pre-activation gradient)."""
    K = x.shape[-1]
    N = weight.shape[0]
    x2, dy2 = x.reshape(-1, K), dy.reshape(-1, N)
    R = x2.shape[0]
    dW = _f32(N, K, device=x.device)
    gemm(dy2, x2, dW, N, K, R, N, K, K, transA=True, transB=True)
    db = colsum(dy2.contiguous())
    dx = None
    if need_dx:
        dx = _f32(R, K, device=x.device)
        gemm(dy2, weight, dx, R, K, N, N, K, K, transB=True)
        dx = dx.reshape(x.shape)
    return dx, dW, db


class LinearFn(torch.autograd.Function):
    """Linear (+ReLU +dropout) through the MFMA GEMM.  mask: uint8 keep flags or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, mask, mask_scale):
        x = x.contiguous()
        y = linear_fwd(x, weight, bias, act, mask, mask_scale)
        ctx.save_for_backward(x, weight, y, mask)
        ctx.act, ctx.mask_scale, ctx.has_bias = act, mask_scale, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y, mask = ctx.saved_tensors
        dz = dy.contiguous()
        if mask is not None:
            dz = dz * mask.view_as(dz) * ctx.mask_scale
        if ctx.act == 1:
            dz = dz * (y > 0)
        elif ctx.act != 0:
            raise NotImplementedError('LinearFn backward supports identity/relu')
        dx, dW, db = linear_bwd(x, weight, dz, ctx.needs_input_grad[0])
        return dx, dW, (db if ctx.has_bias else None), None, None, None


def linear(x, weight, bias=None, act='identity', mask=None, mask_scale=1.0):
    return LinearFn.apply(x, weight, bias, ACT[act], mask, mask_scale)


# ------------------------------------------------------------------------------------------------
# Embedding
# ------------------------------------------------------------------------------------------------

_ERR_FLAGS = {}


class _PinnedRing:
    """Small host -> device copies WITHOUT a stream synchronisation.  `cpu_tensor.to(device)` of pageable memory is hipMemcpyAsync +
    hipStreamSynchronize: the host waits until everything queued on the stream has finished - a train step did that nine times (sequence
    lengths in the encoder, the decoder, the output mask and the loss; the optimizer's pointer table), each one ending the host's run-ahead
    and leaving the GPU idle until the next launch arrives (scripts/dbg_sync_points.py).  Here the values go through a ring of pinned
    staging buffers: a CPU copy into the slot, an asynchronous copy to the device, an event that guards the slot's reuse."""
    SLOTS, BYTES = 32, 1 << 16

    def __init__(self):
        import threading
        self.slots, self.next, self.lock = [], 0, threading.Lock()

    def copy(self, src, device, dtype=None):
        dtype = dtype or src.dtype
        src = src.detach().to(dtype=dtype).contiguous().reshape(-1)
        nbytes = src.numel() * src.element_size()
        if nbytes > self.BYTES or nbytes == 0:
            return src.to(device)
        with self.lock:                                  # host threads driving different streams share the ring
            return self._copy_locked(src, device, dtype, nbytes)

    def _copy_locked(self, src, device, dtype, nbytes):
        if len(self.slots) < self.SLOTS:
            self.slots.append(dict(buf=torch.empty(self.BYTES, dtype=torch.uint8).pin_memory(), event=None))
            slot = self.slots[-1]
        else:
            slot = self.slots[self.next]
            self.next = (self.next + 1) % self.SLOTS
            if slot['event'] is not None:
                slot['event'].synchronize()              # the copy out of this slot was queued 32 copies ago: long done
        view = slot['buf'][:nbytes].view(dtype)
        view.copy_(src)
        out = torch.empty(src.numel(), dtype=dtype, device=device)
        with torch.cuda.device(device):
            out.copy_(view, non_blocking=True)
            if slot['event'] is None:
                slot['event'] = torch.cuda.Event()
            slot['event'].record()
        return out


_H2D = _PinnedRing()
_H2D_BLOCKING = os.environ.get('MTTS_H2D_BLOCKING', '0') == '1'      # A/B: the synchronising copies of round 5


def to_device_async(t, device, dtype=None):
    """`t.to(device=device, dtype=dtype)` for the small host tensors of a step (lengths, ids, pointer tables) without synchronising the
    stream; tensors that already live on the device are converted in place of a copy."""
    if t is None:
        return None
    if t.is_cuda or _H2D_BLOCKING:
        return t.to(device=device, dtype=dtype or t.dtype)
    shape = t.shape
    return _H2D.copy(t, torch.device(device), dtype).reshape(shape)


def _err_flag(device):
    """Two device ints per GPU: [0] raised by kernels that met invalid input (an embedding id outside its table), [1] by a persistent
    decoder kernel whose grid barrier gave up (DecoderArgs.persist_err)."""
    key = torch.device(device).index or 0
    if key not in _ERR_FLAGS:
        _ERR_FLAGS[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return _ERR_FLAGS[key]


def check_device_errors(device=None):
    """Raise if a kernel flagged invalid input since the last check, or if a persistent decoder kernel gave up on a grid barrier
    (synchronises; call once per step / synthesis call)."""
    for key, flag in list(_ERR_FLAGS.items()):
        if device is not None and (torch.device(device).index or 0) != key:
            continue
        v = flag.tolist()
        if v[1] != 0:
            flag.zero_()
            raise _C.MttsError(f'a persistent decoder kernel reported error {v[1]} (2 = a grid barrier timed out: the decode of that call '
                               'is invalid); set MTTS_PERSIST=0 to run the per-step launch schedule')
        if v[0] != 0:
            flag.zero_()
            raise _C.MttsError('an embedding id was outside its table (symbol id >= symbols_count()+3, speaker id >= '
                               'hp.speaker_number or language id >= hp.language_number); the row was read as zeros')


_ERR_POLL = {}


def poll_device_errors(device=None):
    """check_device_errors without stalling the stream: enqueues a copy of this GPU's error words into pinned host memory and raises
    for what the PREVIOUS poll fetched (if that copy has completed - after a whole train step it has).  Called once per optimizer step
    by bench.train_step: a persistent decoder kernel that gave up is reported one step later, and the guarded optimizer step
    (AdamArgs.guard) has kept the invalid step away from the weights in the meantime."""
    for key, flag in list(_ERR_FLAGS.items()):
        if device is not None and (torch.device(device).index or 0) != key:
            continue
        st = _ERR_POLL.get(key)
        if st is None:
            st = _ERR_POLL[key] = dict(host=torch.zeros(2, dtype=torch.int32).pin_memory(), event=None)
        if st['event'] is not None:
            if not st['event'].query():
                continue                               # the previous copy is still in flight: look again at the next call
            st['event'] = None
            if int(st['host'][0]) != 0 or int(st['host'][1]) != 0:
                check_device_errors(flag.device)       # synchronises, clears the words and raises with the full message
        with torch.cuda.device(flag.device):
            st['host'].copy_(flag, non_blocking=True)
            st['event'] = torch.cuda.Event()
            st['event'].record()


class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, padding_idx):
        require_gpu(table, ids)
        ids = ids.contiguous()
        D = table.shape[1]
        out = _f32(*ids.shape, D, device=table.device)
        check(lib().mtts_embedding_fwd(ptr(table), ptr(ids), ptr(out), ids.numel(), D, D, 0, ctypes.c_long(table.shape[0]),
                                       ptr(_err_flag(table.device)), stream_ptr()), 'embedding_fwd')
        ctx.save_for_backward(ids)
        ctx.shape, ctx.padding_idx = table.shape, padding_idx
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dout = dout.contiguous()
        D = ctx.shape[1]
        dtable = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        check(lib().mtts_embedding_bwd(ptr(dout), ptr(ids), ptr(dtable), ids.numel(), D, D, 0,
                                       -1 if ctx.padding_idx is None else ctx.padding_idx, ctypes.c_long(ctx.shape[0]),
                                       stream_ptr()), 'embedding_bwd')
        return dtable, None, None


def embedding(table, ids, padding_idx=None):
    return EmbeddingFn.apply(table, ids, padding_idx)


# ------------------------------------------------------------------------------------------------
# Conv1d (+BN +act +dropout +highway), channel-last
# ------------------------------------------------------------------------------------------------

def pack_conv_weight(weight):
    """torch [O, I, k] -> implicit-GEMM layout [O, k, I]."""
    O, I, k = weight.shape
    out = _f32(O, k, I, device=weight.device)
    check(lib().mtts_conv_weight_pack(ptr(weight.contiguous()), ptr(out), O, I, k, 1, stream_ptr()), 'conv_weight_pack')
    return out


def unpack_conv_weight(packed, O, I, k):
    out = _f32(O, I, k, device=packed.device)
    check(lib().mtts_conv_weight_pack(ptr(packed), ptr(out), O, I, k, 0, stream_ptr()), 'conv_weight_pack')
    return out


def conv1d_fwd(x, wp, k, dilation, groups):
    """x [N, L, Cin] channel-last, wp packed [O, k, Cin/G] -> [N, L, O] ('same' zero padding)."""
    N_, L, Cin = x.shape
    O = wp.shape[0]
    Cg, Og = Cin // groups, O // groups
    y = _f32(N_, L, O, device=x.device)
    p = (k - 1) * dilation // 2
    gemm(x, wp, y, N_ * L, Og, k * Cg, Cin, k * Cg, O, taps=k, Kc=Cg, seq_len=L, shift_mode=1, shift0=-p, dshift=dilation,
         batch=groups, a_z=Cg, b_z=Og * k * Cg, c_z=Og)
    return y


def conv1d_bwd(x, wp, dy, k, dilation, groups, need_dx=True):
    """Gradients of conv1d_fwd: dx [N,L,Cin], dwp packed [O,k,Cin/G]."""
    N_, L, Cin = x.shape
    O = wp.shape[0]
    Cg, Og = Cin // groups, O // groups
    R = N_ * L
    p = (k - 1) * dilation // 2
    dwp = _f32(O, k, Cg, device=x.device)
    # dW[o, t, c] = sum_r dy[r, o] * x[r + shift_t, c]   (one TN GEMM per tap through grid.z)
    gemm(dy, x, dwp, Og, Cg, R, O, Cin, k * Cg, transA=True, transB=True, seq_len=L, shift_mode=2, shift0=-p,
         dshift=dilation, batch=groups, zt=k, a_z=Og, b_z=Cg, c_z=Og * k * Cg, c_ztap=Cg)
    dx = None
    if need_dx:
        dx = _f32(N_, L, Cin, device=x.device)
        # dx[r, c] = sum_t sum_o dy[r - shift_t, o] * w[o, t, c]
        gemm(dy, wp, dx, R, Cg, k * Og, O, k * Cg, Cin, transB=True, taps=k, Kc=Og, seq_len=L, shift_mode=1, shift0=p,
             dshift=-dilation, b_tap=Cg, batch=groups, a_z=Og, b_z=Og * k * Cg, c_z=Cg)
    return dx, dwp


def _bn_args(x2, gamma, beta, rmean, rvar, smean, srstd, mask, resid, y, ws, training, momentum, eps, act, mask_scale,
             hw_groups):
    a = _C.BnArgs()
    a.x, a.gamma, a.beta = ptr(x2), ptr(gamma), ptr(beta)
    a.running_mean, a.running_var, a.save_mean, a.save_rstd = ptr(rmean), ptr(rvar), ptr(smean), ptr(srstd)
    a.mask, a.resid, a.y, a.ws = ptr(mask), ptr(resid), ptr(y), ptr(ws)
    a.R, a.C = x2.shape[0], x2.shape[1]
    a.training, a.momentum, a.eps, a.act, a.mask_scale, a.hw_groups = int(training), momentum, eps, act, mask_scale, hw_groups
    return a


class ConvBnActFn(torch.autograd.Function):
    """pad -> conv1d(groups, dilation, no bias) -> BatchNorm -> act -> dropout [-> highway gate], channel-last.

    reference: ConvBlock modules/layers.py:50-86, HighwayConvBlock :134-153 and the generated variants :89-178
    (the generated kernel / affine tensors are passed in as `weight`, `gamma`, `beta`)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, rmean, rvar, mask, cfg):
        k, dilation, groups, act, training, momentum, eps, mask_scale, highway = cfg[:9]
        packed = len(cfg) > 9 and cfg[9]          # weight already in the implicit-GEMM layout [O, k, I/G] (generated kernels)
        require_gpu(x, weight)
        x = x.contiguous()
        N_, L, Cin = x.shape
        O = weight.shape[0]
        wp = weight.contiguous() if packed else pack_conv_weight(weight)
        conv = conv1d_fwd(x, wp, k, dilation, groups)
        conv2 = conv.view(N_ * L, O)
        dev = x.device
        smean, srstd = _f32(O, device=dev), _f32(O, device=dev)
        ws = _f32(int(lib().mtts_bn_workspace_floats(O)), device=dev)
        Cy = O // 2 if highway else O
        y = _f32(N_, L, Cy, device=dev)
        a = _bn_args(conv2, gamma.contiguous(), beta.contiguous(), rmean, rvar, smean, srstd, mask, x if highway else None, y, ws,
                     training, momentum, eps, act, mask_scale, groups if highway else 0)
        check(lib().mtts_bn_act_fwd(ctypes.byref(a), stream_ptr()), 'bn_act_fwd')
        ctx.save_for_backward(x, wp, conv, gamma, beta, smean, srstd, mask)
        ctx.cfg, ctx.wshape = cfg, tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp, conv, gamma, beta, smean, srstd, mask = ctx.saved_tensors
        k, dilation, groups, act, training, momentum, eps, mask_scale, highway = ctx.cfg[:9]
        packed = len(ctx.cfg) > 9 and ctx.cfg[9]
        N_, L, Cin = x.shape
        O = conv.shape[2]
        dev = x.device
        dy = dy.contiguous()
        dconv = _f32(N_, L, O, device=dev)
        dgamma, dbeta = _f32(O, device=dev), _f32(O, device=dev)
        dresid = _f32(N_, L, O // 2, device=dev) if highway else None
        ws = _f32(int(lib().mtts_bn_workspace_floats(O)), device=dev)
        a = _bn_args(conv.view(N_ * L, O), gamma.contiguous(), beta.contiguous(), None, None, smean, srstd, mask,
                     x if highway else None, None, ws, training, momentum, eps, act, mask_scale, groups if highway else 0)
        a.dy, a.dx, a.dgamma, a.dbeta, a.dresid = ptr(dy), ptr(dconv), ptr(dgamma), ptr(dbeta), ptr(dresid)
        check(lib().mtts_bn_act_bwd(ctypes.byref(a), stream_ptr()), 'bn_act_bwd')
        dx, dwp = conv1d_bwd(x, wp, dconv, k, dilation, groups, ctx.needs_input_grad[0] or highway)
        if highway:
            dx = dx + dresid
        dw = dwp if packed else unpack_conv_weight(dwp, *ctx.wshape)
        return dx, dw, dgamma, dbeta, None, None, None, None


def conv_bn_act(x, weight, gamma, beta, rmean, rvar, mask, *, kernel, dilation=1, groups=1, act='identity', training=True,
                momentum=0.1, eps=1e-5, mask_scale=1.0, highway=False, packed=False):
    cfg = (kernel, dilation, groups, ACT[act], training, momentum, eps, mask_scale, highway, packed)
    return ConvBnActFn.apply(x, weight, gamma, beta, rmean, rvar, mask, cfg)


class GenKernelFn(torch.autograd.Function):
    """Generated convolution kernel straight into the implicit-GEMM layout (reference Conv1dGenerated.forward,
    modules/generated.py:34-42): hidden [G, bott] x w_kernel [(O/G)(I/G)k, bott]^T + b_kernel -> packed [O, k, I/G].
    One bandwidth-bound kernel each way (mtts_gen_params_fwd / _bwd) instead of a padded MFMA GEMM + view + repack."""

    @staticmethod
    def forward(ctx, hidden, w_kernel, b_kernel, dims):
        Og, Cg, k = dims
        require_gpu(hidden, w_kernel)
        hidden, w_kernel = hidden.contiguous(), w_kernel.contiguous()
        G, bott = hidden.shape
        wp = _f32(G * Og, k, Cg, device=hidden.device)
        a = _C.GenParamsArgs()
        a.hidden, a.w_kernel, a.b_kernel, a.w_packed = ptr(hidden), ptr(w_kernel), ptr(b_kernel), ptr(wp)
        a.G, a.bott, a.Og, a.Cg, a.k = G, bott, Og, Cg, k
        check(lib().mtts_gen_params_fwd(ctypes.byref(a), stream_ptr()), 'mtts_gen_params_fwd')
        ctx.save_for_backward(hidden, w_kernel)
        ctx.dims, ctx.has_bias = dims, b_kernel is not None
        return wp

    @staticmethod
    def backward(ctx, dwp):
        hidden, w_kernel = ctx.saved_tensors
        Og, Cg, k = ctx.dims
        G, bott = hidden.shape
        dev = hidden.device
        dwp = dwp.contiguous()
        nslab = int(lib().mtts_gen_params_slabs(Og, Cg))
        dwk = _f32(*w_kernel.shape, device=dev)
        dbk = _f32(w_kernel.shape[0], device=dev) if ctx.has_bias else None
        slab = _f32(nslab, G * bott, device=dev)
        a = _C.GenParamsArgs()
        a.hidden, a.w_kernel, a.d_w_packed, a.d_w_kernel, a.d_b_kernel, a.d_hidden_slab = ptr(hidden), ptr(w_kernel), ptr(dwp), ptr(dwk), ptr(dbk), ptr(slab)
        a.G, a.bott, a.Og, a.Cg, a.k = G, bott, Og, Cg, k
        check(lib().mtts_gen_params_bwd(ctypes.byref(a), stream_ptr()), 'mtts_gen_params_bwd')
        dhid = _f32(G * bott, device=dev)
        ws = _f32(int(lib().mtts_colsum_workspace_floats(G * bott)), device=dev)
        check(lib().mtts_colsum(ptr(slab), ptr(dhid), nslab, G * bott, G * bott, ptr(ws), stream_ptr()), 'mtts_colsum')
        return dhid.view(G, bott), dwk, dbk, None


def generated_kernel(hidden, w_kernel, b_kernel, Og, Cg, k):
    return GenKernelFn.apply(hidden, w_kernel, b_kernel, (Og, Cg, k))


# ------------------------------------------------------------------------------------------------
# BiLSTM (packed-sequence semantics)
# ------------------------------------------------------------------------------------------------

class BiLstmFn(torch.autograd.Function):
    """Bidirectional LSTM with packed-sequence semantics (reference modules/encoder.py:41-44); x [B, L, C]."""

    @staticmethod
    def forward(ctx, x, lengths, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        require_gpu(x, w_ih)
        B, L, Cin = x.shape
        x_tm = x.transpose(0, 1).contiguous()            # time-major [L, B, C]
        H = w_hh.shape[1]
        dev = x.device
        a = _C.BiLstmArgs()
        a.B, a.L, a.Cin, a.H = B, L, Cin, H
        lengths32 = to_device_async(lengths, dev, torch.int32).contiguous()
        a.x, a.lengths = ptr(x_tm), ptr(lengths32)
        ws = [(w_ih.contiguous(), w_hh.contiguous(), b_ih.contiguous(), b_hh.contiguous()),
              (w_ih_r.contiguous(), w_hh_r.contiguous(), b_ih_r.contiguous(), b_hh_r.contiguous())]
        lib().mtts_bilstm_buffer_elems.restype = ctypes.c_long
        n = lambda field: int(lib().mtts_bilstm_buffer_elems(ctypes.byref(a), 0, field.encode()))      # sizes come from the library
        xproj = [_f32(n('xproj'), device=dev).view(L, B, 4 * H) for _ in range(2)]
        h = [torch.zeros(n('h'), device=dev).view(L + 1, B, H) for _ in range(2)]
        c = [torch.zeros(n('c'), device=dev).view(L + 1, B, H) for _ in range(2)]
        gates = [_f32(n('gates'), device=dev).view(L, B, 4 * H) for _ in range(2)]
        y = _f32(n('y'), device=dev).view(B, L, 2 * H)
        for d in range(2):
            a.w_ih[d], a.w_hh[d], a.b_ih[d], a.b_hh[d] = (t.data_ptr() for t in ws[d])
            a.xproj[d], a.h[d], a.c[d], a.gates[d] = xproj[d].data_ptr(), h[d].data_ptr(), c[d].data_ptr(), gates[d].data_ptr()
        a.y = ptr(y)
        check(lib().mtts_bilstm_fwd(ctypes.byref(a), stream_ptr()), 'bilstm_fwd')
        ctx.save_for_backward(x_tm, lengths32, *ws[0], *ws[1], *h, *c, *gates)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .backward import bilstm_bwd
        return bilstm_bwd(ctx, dy)


def bilstm(x, lengths, params):
    return BiLstmFn.apply(x, lengths, *params)


def grad_reverse_clamp(g, l, c):
    """clamp(g, -c, c) * (-l): gradient-reversal backward (reference modules/classifier.py:16-18)."""
    require_gpu(g)
    g = g.contiguous()
    out = torch.empty_like(g)
    check(lib().mtts_grad_reverse_clamp(ptr(g), ptr(out), ctypes.c_long(g.numel()), ctypes.c_float(l), ctypes.c_float(c),
                                        stream_ptr()), 'grad_reverse_clamp')
    return out
