"""Small host helpers that callers of the reference import from `utils` (reference utils/__init__.py:7-37 for the
behaviour of each name)."""
import torch

_DP_PREFIX = 'module.'


def lengths_to_mask(lengths, max_length=None):
    """lengths [B] -> bool [B, max_length] (default: the longest entry), True on valid positions."""
    width = int(lengths.max()) if max_length is None else int(max_length)
    positions = torch.arange(width, device=lengths.device)
    return positions.unsqueeze(0) < lengths.unsqueeze(1)


def to_gpu(x):
    """None stays None; tensors are made contiguous and, when a GPU exists, copied to it asynchronously."""
    if x is None:
        return None
    dense = x.contiguous()
    if not torch.cuda.is_available():
        return dense
    return dense.to('cuda', non_blocking=True)


def remove_dataparallel_prefix(state_dict):
    """Checkpoints written through (Distributed)DataParallel carry 'module.' in front of every key: drop it, keep the order."""
    return type(state_dict)((key[len(_DP_PREFIX):] if key.startswith(_DP_PREFIX) else key, value) for key, value in state_dict.items())


def settle_host_heap():
    """Call once a training (or synthesis) loop has reached its steady state - model, optimizer, data pipeline built, one or two steps run.

    A train step here is ~4 300 small launches that the host keeps only a few milliseconds ahead of the GPU (the hardware queues are
    short), so any host pause longer than that is a GPU pause.  Python's cyclic garbage collector makes exactly one such pause per run:
    its first full (generation-2) pass comes ~10 steps in - when the allocation counters of the young generations have overflowed often
    enough - and walks every container object the model, the optimizer state, the ctypes argument blocks and torch's autograd graph
    have created: 85 ms on the bench configuration, one whole train step, during which the GPU drains its queue and idles
    (scripts/dbg_outlier_step.py, profiles/r06_host_gc_outlier.txt: step 10 takes 155 ms instead of 70; with the collector disabled no step
    does).  It is what the round-5 "78-104 ms for the same step" spread at 200 characters was (one such pass inside a 3-5 step window).

    Collect NOW, outside any timed or latency-sensitive region, and freeze the survivors (gc.freeze: the permanent generation is never
    walked again); the collector stays enabled for everything allocated afterwards, whose passes take microseconds.
    MTTS_HOST_GC_FREEZE=0 switches this off (A/B).  Returns the number of objects frozen."""
    import gc
    import os
    if os.environ.get('MTTS_HOST_GC_FREEZE', '1') == '0':
        return 0
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()


def build_model(checkpoint, force_cpu=False):
    """Hyper-parameters and weights from a checkpoint file -> Tacotron on the GPU (on the CPU with `force_cpu`: weights can be
    inspected there, the hot path itself has no CPU implementation)."""
    from ..modules.tacotron2 import Tacotron
    from ..params.params import Params as hp
    where = 'cpu' if force_cpu or not torch.cuda.is_available() else 'cuda'
    state = torch.load(checkpoint, map_location=where, weights_only=False)
    hp.load_state_dict(state['parameters'])           # sizes the model
    model = Tacotron()
    model.load_state_dict(remove_dataparallel_prefix(state['model']))
    return model.to(where)
