"""Host utilities with the reference's names (utils/__init__.py:7-37)."""
from collections import OrderedDict

import torch


def lengths_to_mask(lengths, max_length=None):
    """Boolean mask [B, max_length] from a tensor of lengths."""
    ml = torch.max(lengths) if max_length is None else max_length
    return torch.arange(ml, device=lengths.device)[None, :] < lengths[:, None]


def to_gpu(x):
    """Compact a CPU tensor and move it to the GPU (non-blocking) when one is present."""
    if x is None:
        return x
    x = x.contiguous()
    return x.cuda(non_blocking=True) if torch.cuda.is_available() else x


def remove_dataparallel_prefix(state_dict):
    """Strip the 'module.' prefix that DataParallel / DistributedDataParallel checkpoints carry."""
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k[7:] if k[:7] == "module." else k] = v
    return out


def build_model(checkpoint, force_cpu=False):
    """Load hyper-parameters + weights from a checkpoint file and build the model (reference utils/__init__.py:29-37).
    The hot path needs a GPU; `force_cpu` only controls where the weights are materialised."""
    from ..modules.tacotron2 import Tacotron
    from ..params.params import Params as hp
    device = torch.device("cuda" if torch.cuda.is_available() and not force_cpu else "cpu")
    state = torch.load(checkpoint, map_location=device, weights_only=False)
    hp.load_state_dict(state['parameters'])
    model = Tacotron()
    model.load_state_dict(remove_dataparallel_prefix(state['model']))
    model.to(device)
    return model
