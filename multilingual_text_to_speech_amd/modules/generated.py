"""Parameter generators of the 'generated' encoder; reference modules/generated.py.

`Conv1dGenerated.generate(e)` returns the conv kernel [O, I/G, k] for all groups,
`BatchNorm1dGenerated.generate(e)` the affine (scale, bias) [G*C]; both are Linear(emb->bottleneck)
-> Linear(bottleneck->params) evaluated by the MFMA GEMM (tiny-K, bandwidth bound)."""
import torch
from torch.nn import Linear, Module

from .. import kernels as K


class Conv1dGenerated(Module):
    def __init__(self, embedding_dim, bottleneck_dim, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True):
        super().__init__()
        assert not bias and stride == 1, 'reference call sites use bias=False (batch norm follows), stride 1'
        self._in_channels, self._out_channels, self._kernel_size = in_channels, out_channels, kernel_size
        self._stride, self._padding, self._dilation, self._groups = stride, padding, dilation, groups
        self._bottleneck = Linear(embedding_dim, bottleneck_dim)
        # same left-to-right integer expression as reference modules/generated.py:31
        self._kernel = Linear(bottleneck_dim, out_channels // groups * in_channels // groups * kernel_size)
        self._bias = None

    def generate(self, generator_embedding):
        assert generator_embedding.shape[0] == self._groups, \
            'Number of groups of a convolutional layer must match the number of generators.'
        e = K.linear(generator_embedding, self._bottleneck.weight, self._bottleneck.bias)
        kernel = K.linear(e, self._kernel.weight, self._kernel.bias)
        return kernel.view(self._out_channels, self._in_channels // self._groups, self._kernel_size)


class BatchNorm1dGenerated(Module):
    def __init__(self, embedding_dim, bottleneck_dim, num_features, groups=1, eps=1e-8, momentum=0.1):
        super().__init__()
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._num_features = num_features // groups
        self._eps, self._momentum, self._groups = eps, momentum, groups
        self._bottleneck = Linear(embedding_dim, bottleneck_dim)
        self._affine = Linear(bottleneck_dim, self._num_features + self._num_features)

    def generate(self, generator_embedding):
        assert generator_embedding.shape[0] == self._groups, \
            'Number of groups of a batchnorm layer must match the number of generators.'
        e = K.linear(generator_embedding, self._bottleneck.weight, self._bottleneck.bias)
        affine = K.linear(e, self._affine.weight, self._affine.bias)
        scale = affine[:, :self._num_features].contiguous().view(-1)
        bias = affine[:, self._num_features:].contiguous().view(-1)
        return scale, bias
