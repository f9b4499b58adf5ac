"""Fused loss and optimizer step on top of mtts_tacotron_loss / mtts_clip_adam_step."""
import ctypes
import math

import torch

from . import _C
from ._C import check, lib, ptr, require_gpu, stream_ptr


class TacotronLossFn(torch.autograd.Function):
    """2*MSE(pre) + MSE(post) + BCE(stop, pos_weight 100)/(M+2) + guided attention, values and gradients in one pass
    (reference modules/tacotron2.py:443-485).  Returns the five-vector (mel_pre, mel_pos, stop_token, guided_att, total)."""

    @staticmethod
    def forward(ctx, pre, post, stop, align, target, stop_target, text_len, target_len, g, ga_on, pos_weight, post_target=None):
        require_gpu(pre, post, stop, target)
        pre, post, stop, target, stop_target = (t.contiguous() for t in (pre, post, stop, target, stop_target))
        post_target = None if post_target is None or post_target is target else post_target.contiguous()
        align = align.contiguous() if align is not None else None
        B, M, T = pre.shape
        dev = pre.device
        a = _C.TacoLossArgs()
        d_pre, d_post, d_stop = torch.empty_like(pre), torch.empty_like(post), torch.empty_like(stop)
        d_align = torch.empty_like(align) if align is not None else None
        nblk = 1024
        partials = torch.empty(nblk * 4, dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        from .kernels import to_device_async
        tl = to_device_async(text_len, dev, torch.int32).contiguous()
        fl = to_device_async(target_len, dev, torch.int32).contiguous()
        a.pre, a.post, a.target, a.stop, a.stop_target, a.align = ptr(pre), ptr(post), ptr(target), ptr(stop), ptr(stop_target), ptr(align)
        a.post_target = ptr(post_target)
        a.text_len, a.target_len = ptr(tl), ptr(fl)
        a.d_pre, a.d_post, a.d_stop, a.d_align, a.partials, a.out = ptr(d_pre), ptr(d_post), ptr(d_stop), ptr(d_align), ptr(partials), ptr(out)
        a.B, a.M, a.T, a.L, a.nblk = B, M, T, (align.shape[2] if align is not None else 0), nblk
        a.ga_on, a.g, a.pos_weight, a.gscale = int(ga_on), float(g), float(pos_weight), 1.0
        check(lib().mtts_tacotron_loss(ctypes.byref(a), stream_ptr()), 'mtts_tacotron_loss')
        ctx.save_for_backward(d_pre, d_post, d_stop, d_align)
        return out

    @staticmethod
    def backward(ctx, dout):
        d_pre, d_post, d_stop, d_align = ctx.saved_tensors
        # d(total)/d(x): every term enters `total` with weight 1, so the upstream scale is dout[4] plus the per-term entries
        s = dout[4]
        return ((dout[0] + s) * d_pre, (dout[1] + s) * d_post, (dout[2] + s) * d_stop,
                None if d_align is None else (dout[3] + s) * d_align, None, None, None, None, None, None, None, None)


class MaskedCrossEntropyFn(torch.autograd.Function):
    """scale * mean over valid characters of CE(pred[b, l, :], speakers[b])  (reference ReversalClassifier.loss,
    modules/classifier.py:62-69) - value and gradient in one kernel (mtts_masked_cross_entropy)."""

    @staticmethod
    def forward(ctx, pred, speakers, lengths, scale):
        require_gpu(pred)
        pred = pred.contiguous()
        B, L, S = pred.shape
        dev = pred.device
        from .kernels import to_device_async
        spk = to_device_async(speakers, dev, torch.int64).contiguous()
        lens = to_device_async(lengths, dev, torch.int32).contiguous()
        row_loss = torch.empty(B * L, 1, dtype=torch.float32, device=dev)
        dpred = torch.empty_like(pred)
        check(lib().mtts_masked_cross_entropy(ptr(pred), ptr(spk), ptr(lens), ptr(row_loss), ptr(dpred), B, L, S, ctypes.c_float(scale),
                                              stream_ptr()), 'mtts_masked_cross_entropy')
        out = torch.empty(1, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib().mtts_colsum_workspace_floats(1)), dtype=torch.float32, device=dev)
        check(lib().mtts_colsum(ptr(row_loss), ptr(out), B * L, 1, 1, ptr(ws), stream_ptr()), 'mtts_colsum')
        ctx.save_for_backward(dpred)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam (same hyper-parameters, same state_dict layout: step / exp_avg / exp_avg_sq) whose `step` runs
    clip_grad_norm_ + the Adam update in three kernel launches over device-side tensor tables
    (reference train.py:84-85: clip_grad_norm_(0.25) over ALL parameters, then Adam(lr, weight_decay=L2-coupled);
    train.py:261-270: optionally two parameter groups, the encoder with its own learning rate).

    One table spans every parameter that has a gradient (global norm / clip coefficient); the update runs once per
    (parameter group, bias-correction step) so per-group learning rates and parameters that skipped earlier steps keep
    torch's semantics.

    Skip semantics (differs from the reference, on purpose): when the gradient norm is not finite, or this GPU's device error word is
    set, the update is skipped ON THE DEVICE (clip coefficient -1 in `norm[1]`, mtts.h AdamArgs.guard) - the weights and moments stay
    untouched while `state['step']` and any LR schedule advance.  The reference's clip_grad_norm_ + Adam would write NaNs into every
    weight instead.  `poll_skipped()` surfaces it without stalling the stream: `skipped_steps` counts them and a warning names the
    first one; a data-parallel rank that skips while others update is caught by the device-error poll of the same step."""

    CHUNK = 1 << 16

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._tables = {}
        self.skipped_steps = 0
        self._skip_poll = None
        self._steps_issued = 0

    def _account_landed(self):
        """Count the copies that have landed, oldest first."""
        ring = self._skip_poll
        while ring and ring['slots'] and ring['slots'][0]['event'].query():
            st = ring['slots'].pop(0)
            if float(st['host'][1]) < 0:
                self.skipped_steps += 1
                if self.skipped_steps == 1 or self.skipped_steps % 100 == 0:
                    import warnings
                    warnings.warn(f'FusedAdam: {self.skipped_steps} optimizer step(s) skipped on the device (gradient norm '
                                  f'{float(st["host"][0])} not finite, or a device error word was set); weights unchanged')
            ring['free'].append(st)
        return self.skipped_steps

    def poll_skipped(self, wait=False):
        """Non-blocking: enqueue a copy of [grad_norm, clip_coefficient] of the step just issued into a pinned slot of a small ring and
        account for every earlier copy that has landed.  One slot per step in flight: the host may run several steps ahead of the GPU,
        and `norm[1]` of EVERY step is fetched before the next step overwrites it on the device (the copy is stream-ordered between
        the two).  A second call without a new step only accounts.  Returns the number of skipped steps seen so far - it trails the
        device by the steps still in flight; `wait=True` blocks until the step just issued has been accounted for."""
        norm = self._tables.get('norm')
        if norm is None:
            return self.skipped_steps
        if self._skip_poll is None:
            self._skip_poll = dict(slots=[], free=[], polled=0)
        ring = self._skip_poll
        self._account_landed()
        if ring['polled'] != self._steps_issued:
            ring['polled'] = self._steps_issued
            st = ring['free'].pop() if ring['free'] else dict(host=torch.zeros(2, dtype=torch.float32).pin_memory(), event=torch.cuda.Event())
            with torch.cuda.device(norm.device):
                st['host'].copy_(norm, non_blocking=True)
                st['event'].record()
            ring['slots'].append(st)
        if wait and ring['slots']:
            ring['slots'][-1]['event'].synchronize()
            self._account_landed()
        return self.skipped_steps

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's state_dict, including one saved by the REFERENCE's two-group optimizer (train.py:261-270): its first
        group lists the prenet / attention tensors twice (once through `_decoder`, once directly), so the packed `params` lists
        repeat indices.  train.make_optimizer registers every tensor once, in the same first-occurrence order; dropping the repeated
        indices makes the two layouts identical."""
        groups = []
        for g in state_dict['param_groups']:
            seen, uniq = set(), []
            for i in g['params']:
                if i not in seen:
                    seen.add(i)
                    uniq.append(i)
            groups.append(dict(g, params=uniq))
        super().load_state_dict(dict(state_dict, param_groups=groups))
        self._tables = {}               # the moment buffers were replaced: the cached device tables point at freed memory

    def _table(self, name, plist):
        """Device tables of one launch: per-tensor pointers (parameter, gradient, both moments) and the chunk map.  The chunk map depends
        on the sizes only and is built once.  The POINTERS change whenever the gradients are re-allocated (`zero_grad(set_to_none=True)`
        does that every step): they are uploaded again - asynchronously, through the pinned ring of kernels.to_device_async; round 5
        rebuilt all four tables with `torch.tensor(..., device=)` in that case, i.e. four stream synchronisations per optimizer step."""
        from .kernels import to_device_async
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr())
                    for p in plist)
        sizes = tuple(p.numel() for p in plist)
        t = self._tables.get(name)
        if t is not None and t['key'] == key:
            return t
        dev = plist[0].device
        ptrs = torch.tensor([x for k in key for x in k], dtype=torch.int64)
        if t is not None and t['sizes'] == sizes and t['ptrs'].device == dev:
            t['ptrs'].copy_(to_device_async(ptrs, dev))      # stream-ordered behind the previous step's kernels, ahead of this step's
            t['key'] = key
            return t
        ct, co, cl = [], [], []
        for i, n in enumerate(sizes):
            for off in range(0, n, self.CHUNK):
                ct.append(i); co.append(off); cl.append(min(self.CHUNK, n - off))
        up = lambda values, dtype: to_device_async(torch.tensor(values, dtype=dtype), dev) if len(values) * 8 <= (1 << 16) else torch.tensor(values, dtype=dtype, device=dev)
        t = dict(key=key, sizes=sizes, n=len(ct), ptrs=to_device_async(ptrs, dev), ct=up(ct, torch.int32), co=up(co, torch.int64), cl=up(cl, torch.int32),
                 partials=torch.empty(len(ct), dtype=torch.float32, device=dev))
        self._tables[name] = t
        return t

    def _launch(self, t, norm, phase, max_norm, group=None, step=1):
        a = _C.AdamArgs()
        a.ptrs, a.chunk_tensor, a.chunk_off, a.chunk_len = ptr(t['ptrs']), ptr(t['ct']), ptr(t['co']), ptr(t['cl'])
        a.norm_partials, a.norm_out, a.nchunks, a.phase = ptr(t['partials']), ptr(norm), t['n'], phase
        a.max_norm = float(max_norm)
        a.guard = ptr(self._guard)
        if group is not None:
            b1, b2 = group['betas']
            a.weight_decay, a.beta1, a.beta2, a.eps = group['weight_decay'], b1, b2, group['eps']
            a.step_size = group['lr'] / (1 - b1 ** step)
            a.inv_sqrt_bc2 = 1.0 / math.sqrt(1 - b2 ** step)
        check(lib().mtts_clip_adam_step(ctypes.byref(a), stream_ptr()), 'mtts_clip_adam_step')

    @torch.no_grad()
    def step(self, closure=None, max_norm=0.0):
        """One update; `max_norm > 0` applies clip_grad_norm_ over ALL of this optimizer's parameters first.  Returns the
        device tensor [grad_norm, clip_coefficient]."""
        work = []                                  # (group index, ordinal within the group, step value, parameters)
        everything = []
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group['params'] if p.grad is not None]
            if not plist:
                continue
            require_gpu(*plist)
            by_step = {}
            for p in plist:
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                assert p.is_contiguous() and p.grad.is_contiguous()
                st['step'] += 1
                by_step.setdefault(int(st['step']), []).append(p)
            everything += plist
            work += [(gi, k, step, ps) for k, (step, ps) in enumerate(sorted(by_step.items()))]
        if not everything:
            return None
        self._steps_issued += 1
        dev = everything[0].device
        # device-side guard: the update is skipped when this GPU's persistent-kernel error word is set or the gradient norm is not
        # finite (mtts.h AdamArgs.guard) - no host synchronisation; kernels.poll_device_errors raises one step later
        from .kernels import _err_flag
        from .dist import agree_on_guard
        self._guard = agree_on_guard(_err_flag(dev))        # data parallel: every rank skips when any rank has to (one tiny MAX all-reduce)
        norm = self._tables.get('norm')
        if norm is None or norm.device != dev:
            norm = self._tables['norm'] = torch.zeros(2, dtype=torch.float32, device=dev)
        if len(work) == 1:                         # the common case: one group, one step value -> one fused call
            gi, _, step, ps = work[0]
            self._launch(self._table(('all',), ps), norm, 0, max_norm, self.param_groups[gi], step)
            return norm
        self._launch(self._table(('all',), everything), norm, 1, max_norm)
        for gi, k, step, ps in work:
            self._launch(self._table((gi, k), ps), norm, 2, max_norm, self.param_groups[gi], step)
        return norm
