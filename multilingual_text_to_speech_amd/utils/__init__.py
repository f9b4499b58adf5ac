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
