"""Location-sensitive attention parameters; reference modules/attention.py:6-86.

The per-step arithmetic (query projection, location filter, energies, masked softmax, context) runs inside
mtts_decoder_fwd/bwd (csrc/attention.hip); this module only owns the parameters under the reference's names.
The reference's ForwardAttention variants are marked "undebugged" there and are not provided."""
import torch
from torch.nn import Linear, Parameter, Conv1d, Module


class AttentionBase(Module):
    def __init__(self, representation_dim, query_dim, memory_dim):
        super().__init__()
        self._bias = Parameter(torch.zeros(1, representation_dim))
        self._energy = Linear(representation_dim, 1, bias=False)
        self._query = Linear(query_dim, representation_dim, bias=False)
        self._memory = Linear(memory_dim, representation_dim, bias=False)
        self._memory_dim = memory_dim


class LocationSensitiveAttention(AttentionBase):
    def __init__(self, kernel_size, channels, smoothing, representation_dim, query_dim, memory_dim):
        super().__init__(representation_dim, query_dim, memory_dim)
        assert not smoothing, 'only softmax normalisation is implemented (every reference config uses smoothing=False)'
        assert kernel_size % 2 == 1, 'attention kernel size must be odd'
        self._location = Linear(channels, representation_dim, bias=False)
        self._loc_features = Conv1d(1, channels, kernel_size, padding=(kernel_size - 1) // 2, bias=False)
        self._smoothing = smoothing
