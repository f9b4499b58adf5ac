"""Decoder loop (attention + 2 LSTM cells + projections) through mtts_decoder_fwd / mtts_decoder_bwd.

reference: Decoder._decode, modules/tacotron2.py:148-209.
"""
import ctypes
import os

import torch

from . import _C
from . import kernels as K
from ._C import check, lib, ptr, require_gpu, stream_ptr


def _round4(x):
    return (x + 3) & ~3


class DecoderState:
    """Device buffers of one decode (time-major, see DecoderArgs in include/mtts.h)."""

    def __init__(self, B, L, T, dims, device, n_prenet, save_gates=True, fast=False, kq=8, precision=0):
        M, P, H, A, Dm, ksz, C = dims
        self.B, self.L, self.T, self.dims, self.n_prenet, self.kq, self.fast = B, L, T, dims, n_prenet, kq, fast
        self.Mo = _round4(M + 1)
        self.precision = int(precision)
        # every buffer size comes from the library (mtts_decoder_buffer_elems): shape-only argument block first
        sh = _C.DecoderArgs()
        sh.B, sh.L, sh.T, sh.M, sh.P, sh.H, sh.A, sh.Dm, sh.ksz, sh.C = B, L, T, M, P, H, A, Dm, ksz, C
        sh.n_prenet, sh.kq, sh.fast, sh.precision = n_prenet, kq, int(bool(fast)), self.precision
        lib().mtts_decoder_buffer_elems.restype = ctypes.c_long

        def n(field):
            v = int(lib().mtts_decoder_buffer_elems(ctypes.byref(sh), field.encode()))
            if v < 0:
                raise _C.MttsError(f'mtts_decoder_buffer_elems: unknown field {field}')
            return v
        e = lambda field: torch.empty(n(field), dtype=torch.float32, device=device)
        z = lambda field: torch.zeros(n(field), dtype=torch.float32, device=device)
        raw = lambda field: torch.empty(n(field), dtype=torch.uint8, device=device)

        def z0(field, *shape):              # only slot 0 (the initial state) is read before it is written: no need to clear ~1 GB per decode
            t = e(field).view(*shape)
            t[0].zero_()
            return t
        self.prenet_act = [e('prenet_act').view(T, B, P) for _ in range(n_prenet)]
        self.U, self.Mt, self.PL = e('U').view(A, ksz), e('Mt').view(B, L, A), e('PL').view(2, B, L, A)
        # K-split step kernels (csrc/lstm_step.hip): query partials come in H/16 slabs
        off = os.environ.get('MTTS_NO_LSTEP', '0')
        ok = H % 32 == 0 and Dm % 32 == 0 and A % 16 == 0 and A <= 256
        self.use_lstep = bool(fast and off != '1' and ok)                       # teacher-forced schedule: hoisted projections
        use_lgen = bool(not fast and off != '1' and ok and P % 32 == 0)          # general schedule: un-hoisted 3-segment operands
        self.qpart = e('qpart').view(-1, B, A) if (self.use_lstep or use_lgen) else torch.empty(kq, B, A, dtype=torch.float32, device=device)
        self.att_w2p = self.att_bias_u = self.att_w_pre_u = self.gate_part = None
        self.gen_w2p = self.gen_bias_u = self.gen_w_ih_u = self.gate_part_gen = None
        if fast and off not in ('1', 'gen') and H % 32 == 0:
            self.gen_w2p, self.gen_bias_u, self.gen_w_ih_u, self.gate_part_gen = raw('gen_w2p'), e('gen_bias_u'), e('gen_w_ih_u'), e('gate_part_gen')
        if self.use_lstep:
            self.att_w2p, self.att_bias_u, self.att_w_pre_u, self.gate_part = raw('att_w2p'), e('att_bias_u'), e('att_w_pre_u'), e('gate_part')
        self.prenet_wp = None
        if not fast and n_prenet == 2 and M % 16 == 0 and P % 16 == 0:        # free-running steps: fused two-layer prenet kernel
            self.prenet_wp = [e('prenet_wp0'), e('prenet_wp1')]
        if use_lgen:
            self.att_w2p, self.att_bias_u, self.gate_part = raw('att_w2p'), e('att_bias_u'), e('gate_part')
            self.gen_w2p, self.gen_bias_u, self.gate_part_gen = raw('gen_w2p'), e('gen_bias_u'), e('gate_part_gen')
        self.h_att, self.c_att = z0('h_att', T + 1, B, H), z0('c_att', T + 1, B, H)
        self.h_gen, self.c_gen = z0('h_gen', T + 1, B, H), z0('c_gen', T + 1, B, H)
        self.ctx, self.cum = z0('ctx', T + 1, B, Dm), z0('cum', T + 1, B, L)
        self.align = e('align').view(T, B, L)
        self.gates_att = e('gates_att').view(T, B, 4 * H) if save_gates else None
        self.gates_gen = e('gates_gen').view(T, B, 4 * H) if save_gates else None
        self.out = z('out').view(T + 1, B, self.Mo)
        self.pre_att = e('pre_att').view(T, B, 4 * H) if fast else None
        self.pre_gen = e('pre_gen').view(T, B, 4 * H) if fast else None
        self.q_all = e('q_all').view(T, B, A) if save_gates else None
        # MFMA-tile-order copies of the recurrent operands (only when the widths are multiples of 16)
        Bp = (B + 15) & ~15
        use_pack = os.environ.get('MTTS_NO_PACK', '0') != '1'      # debugging / A-B switch: row-major operands only
        hp_ok, dp_ok = use_pack and H % 16 == 0, use_pack and Dm % 16 == 0
        zp = (lambda f, *sh_: z0(f, *sh_)) if Bp == B else (lambda f, *sh_: z(f).view(*sh_))      # padded batch rows of the packed copies stay zero
        self.h_att_p = zp('h_att_p', T + 1, Bp * H) if hp_ok else None
        self.h_gen_p = zp('h_gen_p', T + 1, Bp * H) if hp_ok else None
        self.ctx_p = zp('ctx_p', T + 1, Bp * Dm) if dp_ok else None
        self.att_w_ctx_p = e('att_w_ctx_p') if (dp_ok and H % 4 == 0) else None
        self.att_w_hh_p = e('att_w_hh_p') if hp_ok else None
        self.gen_w_hh_p = e('gen_w_hh_p') if hp_ok else None
        self.w_query_p = e('w_query_p') if hp_ok else None
        # exchange / barrier workspace of the persistent recurrence kernels (csrc/persist.hip), zero-filled once
        self.persist_ws = torch.zeros(n('persist_ws'), dtype=torch.uint8, device=device) if fast else None
        self._args = (B, L, dims, device, n_prenet, save_gates, fast, kq, precision)

    _PER_STEP = ('prenet_act', 'h_att', 'c_att', 'h_gen', 'c_gen', 'ctx', 'cum', 'align', 'gates_att', 'gates_gen', 'out', 'pre_att',
                 'pre_gen', 'q_all', 'h_att_p', 'h_gen_p', 'ctx_p')

    def grown(self, T):
        """A state with room for T steps that continues this one: every per-step array keeps its first slots (free-running
        synthesis allocates geometrically instead of hp.max_output_length = 5000 frames up front)."""
        B, L, dims, device, n_prenet, save_gates, fast, kq, precision = self._args
        new = DecoderState(B, L, T, dims, device, n_prenet, save_gates, fast, kq, precision)
        for name in self._PER_STEP:
            old, cur = getattr(self, name), getattr(new, name)
            if old is None:
                continue
            if isinstance(old, list):
                for o, c in zip(old, cur):
                    c[:o.shape[0]].copy_(o)
            else:
                cur[:old.shape[0]].copy_(old)
        for name in ('U', 'Mt', 'PL', 'qpart', 'att_w_ctx_p', 'att_w_hh_p', 'gen_w_hh_p', 'w_query_p', 'att_w2p', 'att_bias_u', 'att_w_pre_u',
                     'gate_part', 'gen_w2p', 'gen_bias_u', 'gen_w_ih_u', 'gate_part_gen', 'prenet_wp', 'persist_ws'):      # per-call constants
            setattr(new, name, getattr(self, name))
        return new


def fill_decoder_args(a, st, w, memory, lengths32, frames_in, teacher_host, masks, cfg):
    """Populate a DecoderArgs struct.  `w` maps weight names to contiguous tensors, `masks` to uint8 tensors."""
    M, P, H, A, Dm, ksz, C = st.dims
    a.B, a.L, a.T, a.M, a.P, a.H, a.A, a.Dm, a.ksz, a.C = st.B, st.L, st.T, M, P, H, A, Dm, ksz, C
    a.n_prenet = st.n_prenet
    a.training, a.zone = int(cfg['training']), int(cfg['zone'])
    a.p_prenet, a.p_hidden, a.p_cell = cfg['p_prenet'], cfg['p_hidden'], cfg['p_cell']
    a.memory, a.lengths, a.frames_in = ptr(memory), ptr(lengths32), ptr(frames_in)
    a.teacher = ctypes.cast(teacher_host, ctypes.c_void_p) if teacher_host is not None else None
    for i in range(st.n_prenet):
        a.prenet_w[i], a.prenet_b[i] = w['prenet_w'][i].data_ptr(), w['prenet_b'][i].data_ptr()
        pm = masks.get(f'prenet.{i}')
        a.prenet_mask[i] = pm.data_ptr() if pm is not None else None
        a.prenet_act[i] = st.prenet_act[i].data_ptr()
    if getattr(st, 'prenet_wp', None) is not None:
        a.prenet_wp[0], a.prenet_wp[1] = st.prenet_wp[0].data_ptr(), st.prenet_wp[1].data_ptr()
    for name in ('att_w_ih', 'att_w_hh', 'att_b_ih', 'att_b_hh', 'gen_w_ih', 'gen_w_hh', 'gen_b_ih', 'gen_b_hh', 'w_query',
                 'w_memory', 'w_loc', 'w_conv', 'att_bias', 'w_energy', 'w_out', 'b_out'):
        setattr(a, name, ptr(w[name]))
    a.att_hmask, a.att_cmask = ptr(masks.get('att_h')), ptr(masks.get('att_c'))
    a.gen_hmask, a.gen_cmask = ptr(masks.get('gen_h')), ptr(masks.get('gen_c'))
    for name in ('U', 'Mt', 'PL', 'qpart', 'h_att', 'c_att', 'h_gen', 'c_gen', 'ctx', 'cum', 'align', 'gates_att', 'gates_gen',
                 'out', 'pre_att', 'pre_gen', 'q_all', 'h_att_p', 'h_gen_p', 'ctx_p', 'att_w_ctx_p', 'att_w_hh_p', 'gen_w_hh_p',
                 'w_query_p', 'att_w2p', 'att_bias_u', 'att_w_pre_u', 'gate_part', 'gen_w2p', 'gen_bias_u', 'gen_w_ih_u', 'gate_part_gen'):
        setattr(a, name, ptr(getattr(st, name)))
    a.kq, a.fast, a.precision = st.kq, int(st.fast), st.precision
    ws = getattr(st, 'persist_ws', None)
    a.persist_ws, a.persist_ws_bytes = ptr(ws), (ws.numel() if ws is not None else 0)
    if ws is not None:
        from .kernels import _err_flag
        a.persist_err = ctypes.c_void_p(_err_flag(ws.device).data_ptr() + 4)      # kernels.check_device_errors() reads it
    return a


def decoder_weights(dec, attention, prenet):
    """Collect contiguous weight tensors from the module tree (names follow DecoderArgs)."""
    w = {
        'prenet_w': [l.weight.contiguous() for l in prenet._layers],
        'prenet_b': [l.bias.contiguous() for l in prenet._layers],
        'att_w_ih': dec._attention_lstm.weight_ih, 'att_w_hh': dec._attention_lstm.weight_hh,
        'att_b_ih': dec._attention_lstm.bias_ih, 'att_b_hh': dec._attention_lstm.bias_hh,
        'gen_w_ih': dec._generator_lstm.weight_ih, 'gen_w_hh': dec._generator_lstm.weight_hh,
        'gen_b_ih': dec._generator_lstm.bias_ih, 'gen_b_hh': dec._generator_lstm.bias_hh,
        'w_query': attention._query.weight, 'w_memory': attention._memory.weight,
        'w_loc': attention._location.weight, 'w_conv': attention._loc_features.weight.view(attention._loc_features.weight.shape[0], -1),
        'att_bias': attention._bias.view(-1), 'w_energy': attention._energy.weight.view(-1),
        'w_out': torch.cat((dec._frame_prediction.weight, dec._stop_prediction.weight), 0),
        'b_out': torch.cat((dec._frame_prediction.bias, dec._stop_prediction.bias), 0),
    }
    return {k: (v if isinstance(v, list) else v.contiguous()) for k, v in w.items()}


WEIGHT_ORDER = ['att_w_ih', 'att_w_hh', 'att_b_ih', 'att_b_hh', 'gen_w_ih', 'gen_w_hh', 'gen_b_ih', 'gen_b_hh', 'w_query',
                'w_memory', 'w_loc', 'w_conv', 'att_bias', 'w_energy', 'w_out', 'b_out']


def run_decoder(st, w, memory, lengths32, frames_in, teacher, masks, cfg, t0, t1):
    _C.ensure_workspace(memory.device)
    a = _C.DecoderArgs()
    th = None
    if teacher is not None:
        th = (ctypes.c_uint8 * st.T)(*[int(bool(x)) for x in teacher])
    fill_decoder_args(a, st, w, memory, lengths32, frames_in, th, masks, cfg)
    a.t0, a.t1 = t0, t1
    check(lib().mtts_decoder_fwd(ctypes.byref(a), stream_ptr()), 'mtts_decoder_fwd')
    return a


class DecoderFn(torch.autograd.Function):
    """Teacher-forced (or mixed) training decode.  Inputs: memory [B,L,Dm], target frames [B,T,M] channel-last,
    then the flat weight list (prenet w/b pairs, WEIGHT_ORDER)."""

    @staticmethod
    def forward(ctx, memory, target, lengths, teacher, masks, cfg, n_prenet, *flat):
        require_gpu(memory, target)
        memory = memory.contiguous()
        B, L, Dm = memory.shape
        T, M = target.shape[1], target.shape[2]
        w = {'prenet_w': list(flat[0:2 * n_prenet:2]), 'prenet_b': list(flat[1:2 * n_prenet:2])}
        for i, name in enumerate(WEIGHT_ORDER):
            w[name] = flat[2 * n_prenet + i]
        w = {k: ([t.contiguous() for t in v] if isinstance(v, list) else v.contiguous()) for k, v in w.items()}
        P, H = w['prenet_w'][0].shape[0], w['att_w_hh'].shape[1]
        A, C, ksz = w['w_query'].shape[0], w['w_conv'].shape[0], w['w_conv'].shape[1]
        dev = memory.device
        teacher = [bool(x) for x in teacher]
        fast = all(teacher) and cfg.get('allow_fast', True)
        st = DecoderState(B, L, T, (M, P, H, A, Dm, ksz, C), dev, n_prenet, save_gates=True, fast=fast, kq=cfg.get('kq', 8),
                          precision=cfg.get('precision', _C.get_precision()))
        # frame fed at step t: zero frame at t=0, target[t-1] afterwards (tacotron2.py:129-131), time-major
        frames_in = torch.zeros(T, B, M, dtype=torch.float32, device=dev)
        frames_in[1:] = target[:, :T - 1].transpose(0, 1)
        lengths32 = K.to_device_async(lengths, dev, torch.int32).contiguous()
        run_decoder(st, w, memory, lengths32, frames_in, teacher, masks, cfg, 0, T)
        ctx.st, ctx.w, ctx.masks, ctx.cfg, ctx.teacher, ctx.n_prenet = st, w, masks, cfg, teacher, n_prenet
        ctx.memory, ctx.lengths32, ctx.frames_in = memory, lengths32, frames_in
        spec = st.out[1:, :, :M].transpose(0, 1).contiguous()          # [B,T,M]
        stop = st.out[1:, :, M].transpose(0, 1).contiguous()           # [B,T]
        align = st.align.transpose(0, 1).contiguous()                  # [B,T,L]
        return spec, stop, align

    @staticmethod
    def backward(ctx, dspec, dstop, dalign):
        from .backward import decoder_bwd
        return decoder_bwd(ctx, dspec, dstop, dalign)


def decode_train(memory, target, lengths, teacher, masks, cfg, w):
    n = len(w['prenet_w'])
    flat = []
    for i in range(n):
        flat += [w['prenet_w'][i], w['prenet_b'][i]]
    flat += [w[k] for k in WEIGHT_ORDER]
    return DecoderFn.apply(memory, target, lengths, teacher, masks, cfg, n, *flat)


class GraphedDecode:
    """Persistent buffers + a private stream for hipGraph replay of free-running decode chunks (mtts_decoder_fwd_graphed).

    A captured graph bakes in every pointer, so everything the decoder kernels touch must live at a fixed address between calls:
    the DecoderState, the memory / lengths / keep-flag inputs (copied or regenerated IN PLACE per call) and the concatenated
    frame+stop projection.  One session per (device, B, L, dims, frames, precision); the library keys its graphs by the exact
    argument block, runs a new block eagerly once, captures it the second time and replays it afterwards."""
    _cache = {}
    MAX_SESSIONS = 4          # a session pins ~frames x 0.7 MB per utterance of device memory: least-recently-used sessions are retired
    L_BUCKET = 32             # memories are zero-padded to a multiple of this many positions (lengths mask the padding), so that
                              # utterances of similar length share one session and its graphs instead of capturing per length

    @classmethod
    def bucket(cls, L):
        return ((L + cls.L_BUCKET - 1) // cls.L_BUCKET) * cls.L_BUCKET

    @classmethod
    def get(cls, B, L, frames, dims, dev, n_prenet, kq, precision):
        key = (torch.device(dev).index or 0, B, L, frames, tuple(dims), n_prenet, kq, precision)
        sess = cls._cache.pop(key, None)
        if sess is None:
            while len(cls._cache) >= cls.MAX_SESSIONS:          # retire the least recently used session and its graphs
                old = cls._cache.pop(next(iter(cls._cache)))
                with torch.cuda.device(old.memory.device):
                    lib().mtts_decoder_graphs_clear(ctypes.c_void_p(old.stream.cuda_stream))
            sess = cls(B, L, frames, dims, dev, n_prenet, kq, precision)
        cls._cache[key] = sess                                   # (re)insert as most recently used
        return sess

    def __init__(self, B, L, frames, dims, dev, n_prenet, kq, precision):
        M, P, H, A, Dm, ksz, C = dims
        self.st = DecoderState(B, L, frames, dims, dev, n_prenet, save_gates=False, fast=False, kq=kq, precision=precision)
        self.memory = torch.empty(B, L, Dm, dtype=torch.float32, device=dev)
        self.lengths32 = torch.empty(B, dtype=torch.int32, device=dev)
        self.w_out = torch.empty(M + 1, H + Dm, dtype=torch.float32, device=dev)
        self.b_out = torch.empty(M + 1, dtype=torch.float32, device=dev)
        self.masks = {f'prenet.{i}': torch.empty(frames, B, P, dtype=torch.uint8, device=dev) for i in range(n_prenet)}
        self.stream = torch.cuda.Stream(device=dev)
        self.replayed = 0          # chunks of the last decode that ran as a graph

    def load(self, memory, lengths, w, masks):
        """Copy this call's inputs into the persistent buffers (on the session stream)."""
        self.memory.copy_(memory)
        self.lengths32.copy_(K.to_device_async(lengths, self.memory.device, torch.int32))
        self.w_out.copy_(w['w_out']); self.b_out.copy_(w['b_out'])
        for k, buf in self.masks.items():
            m = masks.get(k)
            if m is None:
                buf.fill_(1)
            else:
                buf.copy_(m[:buf.shape[0]])
        self.st.h_att[0].zero_(); self.st.c_att[0].zero_(); self.st.h_gen[0].zero_(); self.st.c_gen[0].zero_()
        self.st.ctx[0].zero_(); self.st.cum[0].zero_(); self.st.out[0].zero_()
        self.replayed = 0
        return dict(w, w_out=self.w_out, b_out=self.b_out)

    def run(self, w, cfg, t0, t1):
        a = _C.DecoderArgs()
        fill_decoder_args(a, self.st, w, self.memory, self.lengths32, None, None, self.masks, cfg)
        a.t0, a.t1 = t0, t1
        flag = ctypes.c_int(0)
        check(lib().mtts_decoder_fwd_graphed(ctypes.byref(a), ctypes.c_void_p(self.stream.cuda_stream), ctypes.byref(flag)),
              'mtts_decoder_fwd_graphed')
        self.replayed += int(flag.value)
        return a


_COPY_STREAMS = {}


def _copy_stream(dev):
    key = torch.device(dev).index or 0
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _COPY_STREAMS[key]


def decode_free(memory, lengths, w, cfg, masks, max_frames, stop_frames, chunk=32, stop_threshold=0.5, dims=None,
                initial_frames=None, graph=False):
    """Free-running decode with the reference's stop rule (tacotron2.py:201-207), batch >= 1.

    Steps run in chunks of `chunk` on the device; after each chunk the stop logits come back to the host and
    the rule is evaluated per sample.  `masks` is a dict of [max_frames, ...] keep flags or a callable T -> dict (then the
    per-step buffers start at `initial_frames` and double when the decode outgrows them).
    Returns (frames [B,T',M], stop [B,T'], align [B,T',L], n_frames [B])."""
    require_gpu(memory)
    memory = memory.contiguous()
    B, L, Dm = memory.shape
    M = w['w_out'].shape[0] - 1
    P, H = w['prenet_w'][0].shape[0], w['att_w_hh'].shape[1]
    A, C, ksz = w['w_query'].shape[0], w['w_conv'].shape[0], w['w_conv'].shape[1]
    dev = memory.device
    mask_fn = masks if callable(masks) else None
    session = None
    if graph and max_frames <= 2048:          # hipGraph replay: fixed buffers for the whole range (mtts_decoder_fwd_graphed)
        Lb = GraphedDecode.bucket(L)
        if Lb != L:                            # pad the memory to the bucket length; the per-sample lengths mask the padding
            memory = torch.cat((memory, memory.new_zeros(B, Lb - L, Dm)), 1)
        session = GraphedDecode.get(B, Lb, max_frames, (M, P, H, A, Dm, ksz, C), dev, len(w['prenet_w']), cfg.get('kq', 8),
                                    cfg.get('precision', _C.get_precision()))
        caller = torch.cuda.current_stream(dev)
        session.stream.wait_stream(caller)
        with torch.cuda.stream(session.stream):
            out = _decode_free_loop(memory, lengths, w, cfg, mask_fn(max_frames) if mask_fn is not None else masks, max_frames, stop_frames,
                                    chunk, stop_threshold, session, dev, M)
        caller.wait_stream(session.stream)
        if Lb != L:
            out = (out[0], out[1], out[2][:, :, :L].contiguous(), out[3])
        for t in out[:3]:
            t.record_stream(caller)
        return out
    return _decode_free_loop(memory, lengths, w, cfg, masks, max_frames, stop_frames, chunk, stop_threshold, None, dev, M,
                             initial_frames=initial_frames)


def _decode_free_loop(memory, lengths, w, cfg, masks, max_frames, stop_frames, chunk, stop_threshold, session, dev, M, initial_frames=None):
    B, L, Dm = memory.shape
    P, H = w['prenet_w'][0].shape[0], w['att_w_hh'].shape[1]
    A, C, ksz = w['w_query'].shape[0], w['w_conv'].shape[0], w['w_conv'].shape[1]
    mask_fn = masks if callable(masks) else None
    if session is not None:
        w = session.load(memory, lengths, w, masks)
        st, cap = session.st, max_frames
    else:
        cap = min(max_frames, max(chunk, initial_frames or 1024)) if mask_fn is not None else max_frames
        if mask_fn is not None:
            masks = mask_fn(cap)
        st = DecoderState(B, L, cap, (M, P, H, A, Dm, ksz, C), dev, len(w['prenet_w']), save_gates=False, fast=False,
                          kq=cfg.get('kq', 8), precision=cfg.get('precision', _C.get_precision()))
    lengths32 = K.to_device_async(lengths, dev, torch.int32).contiguous()
    # The stop rule runs ON THE DEVICE (mtts_stop_rule_update: per-sample armed / done counters in a device int array); after
    # each chunk ONE int - how many utterances are still running - travels to the host through a copy stream, while the device
    # already runs the NEXT chunk (speculatively).
    state = torch.full((2 * B,), -1, dtype=torch.int32, device=dev)
    ring = torch.zeros(8, dtype=torch.int32, device=dev)          # running counts of the chunks in flight (two at a time)
    n_sub = 0
    copy_stream = _copy_stream(dev)

    def submit(t):
        nonlocal st, masks, cap, n_sub
        t1 = min(max_frames, t + chunk)
        if t1 > cap:                                 # outgrown: double the per-step buffers, keep what has been decoded
            cap = min(max_frames, max(2 * cap, t1))
            st = st.grown(cap)
            masks = mask_fn(cap)                     # fresh draws; only steps >= t read them
        if session is not None:
            session.run(w, cfg, t, t1)
        else:
            run_decoder(st, w, memory, lengths32, None, None, masks, cfg, t, t1)
        slot = ring[n_sub % 8:n_sub % 8 + 1]
        n_sub += 1
        check(lib().mtts_stop_rule_update(ptr(st.out), t, t1, B, st.Mo, M, ctypes.c_float(stop_threshold), int(stop_frames), ptr(state),
                                          ptr(slot), stream_ptr()), 'mtts_stop_rule_update')
        running = torch.empty(1, dtype=torch.int32, pin_memory=True)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ev)
            running.copy_(slot, non_blocking=True)
            done_ev = torch.cuda.Event()
            done_ev.record()
        return t1, running, done_ev

    pending = submit(0) if max_frames > 0 else None
    while pending is not None:
        nxt = submit(pending[0]) if pending[0] < max_frames else None
        pending[2].synchronize()
        if int(pending[1][0]) == 0:
            break
        pending = nxt
    torch.cuda.current_stream(dev).synchronize()
    done = state[B:].tolist()
    # (a speculative chunk may have run past the point where the last utterance ended: `done` is final once set)
    n = [d if d >= 0 else max_frames for d in done]
    Tn = max(n)
    frames = st.out[1:Tn + 1, :, :M].transpose(0, 1).contiguous()
    stop = st.out[1:Tn + 1, :, M].transpose(0, 1).contiguous()
    align = st.align[:Tn].transpose(0, 1).contiguous()
    return frames, stop, align, n
