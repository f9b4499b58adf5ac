"""Data-parallel training across the GPUs of one node: one process per GPU, RCCL (torch.distributed backend
"nccl" on ROCm) gradient all-reduce over xGMI.

Replaces the reference's single-process torch.nn.DataParallel (train.py:173-179,255-256): no per-step parameter
broadcast and no output gather - every rank owns a full replica, computes its local loss and the gradients are
summed bucket by bucket.  BatchNorm statistics stay per rank, which is what DataParallel does per replica.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise the default process group from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('MTTS_SINGLE_DEVICE') == '1':      # plumbing tests: several ranks share device 0 (use with MTTS_DIST_BACKEND=gloo)
        local = 0
    backend = backend or os.environ.get('MTTS_DIST_BACKEND') or None
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_bounds(global_batch, rank, world, groups=1):
    """Contiguous shard [lo, hi) of the global batch for `rank` (what DataParallel.scatter does on dim 0).
    Grouped encoders need whole language groups per shard: sample i belongs to language i mod G
    (reference utils/samplers.py:70-73, modules/encoder.py:206-208)."""
    if global_batch % (groups * world) != 0:
        raise ValueError(f'global batch {global_batch} must be divisible by languages*ranks = {groups}*{world}')
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradientBuckets:
    """Flat fp32 buckets over the parameters (reverse registration order ~ backward order); all-reduce (sum) each
    bucket and scale by 1/world so that the result equals the gradient of the global-batch mean loss.

    Bucket size is chosen for xGMI's point-to-point links: a few large collectives (default 64 MiB) rather than
    per-tensor launches.

    overlap=True (the training loop): every parameter's .grad is a VIEW into its bucket's flat buffer and a
    post-accumulate hook launches the bucket's asynchronous all-reduce the moment its last gradient has been written, so
    the collective of the post-net / decoder buckets runs under the rest of the backward pass.  Use `zero_grad()` instead
    of the optimizer's (which would detach the views) and `all_reduce()` after backward to launch whatever is left (buckets
    with unused parameters), wait and scale.  overlap=False reduces after backward from a concatenated copy."""

    def __init__(self, params, bucket_bytes=64 << 20, overlap=False):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        cur, cur_n = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_n += p.numel()
            if cur_n * 4 >= bucket_bytes:
                self.buckets.append(cur)
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat = [None] * len(self.buckets)
        self.overlap = bool(overlap)
        self._works = [None] * len(self.buckets)
        self._pending = [0] * len(self.buckets)
        self._bucket_of = {}
        self._hooks = []
        if self.overlap:
            for i, bucket in enumerate(self.buckets):
                n = sum(p.numel() for p in bucket)
                self.flat[i] = torch.zeros(n, dtype=bucket[0].dtype, device=bucket[0].device)
                for p in bucket:
                    self._bucket_of[id(p)] = i
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            self.zero_grad()

    def zero_grad(self):
        """Zero the flat buffers and (re)attach every .grad as a view into them; arms the hooks for the next backward."""
        for i, bucket in enumerate(self.buckets):
            self.flat[i].zero_()
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad = self.flat[i][off:off + n].view_as(p)
                off += n
            self._pending[i] = len(bucket)
            self._works[i] = None

    def _launch(self, i):
        if dist.is_initialized() and dist.get_world_size() > 1 and self._works[i] is None:
            self._works[i] = dist.all_reduce(self.flat[i], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def all_reduce(self, world=None, async_op=True):
        if not dist.is_initialized():
            return
        world = world or dist.get_world_size()
        if world == 1:
            return
        if self.overlap:
            for i in range(len(self.buckets)):
                self._launch(i)                       # buckets holding parameters that received no gradient this step
            for i in range(len(self.buckets)):
                self._works[i].wait()
                self.flat[i].mul_(1.0 / world)
            return
        works = []
        for i, bucket in enumerate(self.buckets):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            self.flat[i] = flat
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op))
        for i, bucket in enumerate(self.buckets):
            if async_op:
                works[i].wait()
            flat = self.flat[i]
            flat.mul_(1.0 / world)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s weights and buffers (one-time, replaces DataParallel's per-step replicate)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def agree_on_guard(flag):
    """Data-parallel ranks must take the SAME decision about an optimizer step.  The gradient norm is identical everywhere after the
    all-reduce (a non-finite value on one rank is non-finite on all), but the device error words (`kernels._err_flag`: a persistent
    decoder kernel whose grid barrier gave up, an embedding id outside its table) are per GPU: a rank that skips its guarded Adam
    step (mtts.h AdamArgs.guard) while the others update leaves the replicas different for the rest of the run.  One MAX
    all-reduce over the words, in place, makes every rank see the worst word of any rank: all of them skip the step, and all of
    them raise at their next error poll.  A COLLECTIVE (call it once per optimizer step on every rank); no-op outside data parallel.
    Reference semantics it protects: train.py:84-85, 173-179 (one model, one optimizer step per global batch)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flag
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return flag


def global_mean_scale(n_local, device):
    """Factor that turns a rank's LOCAL mean over `n_local` items into its share of the GLOBAL mean after the gradient all-reduce
    (which averages over ranks): mean_global = (1 / world) * sum_r [ local_mean_r * n_local_r * world / n_global ].
    Used for the adversarial classifier loss, a mean over the VALID characters of the batch (reference
    modules/classifier.py:62-69 on the gathered global batch): shards with different text lengths hold different numbers of them.
    `n_local` may be a python number or a (device) tensor - a tensor is used as it is, so the caller's stream never waits for the
    host.  Returns a 1-element tensor on `device`; 1.0 when not running data parallel.  A COLLECTIVE: every rank must call it the
    same number of times (TacotronLoss does so in the sharded training step only)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return torch.ones(1, dtype=torch.float32, device=device)
    world = dist.get_world_size()
    if torch.is_tensor(n_local):
        from .kernels import to_device_async          # no stream synchronisation for a host-side count
        local = to_device_async(n_local.detach(), device, torch.float32).reshape(1) if torch.device(device).type == 'cuda' else \
            n_local.detach().to(device=device, dtype=torch.float32).reshape(1)
    else:
        local = torch.tensor([float(n_local)], dtype=torch.float32, device=device)
    total = local.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    return (local * world) / total.clamp_min(1.0)
