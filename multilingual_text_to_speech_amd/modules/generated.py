"""Parameter generators of the 'generated' encoder (reference modules/generated.py: Conv1dGenerated :6-42, BatchNorm1dGenerated :45-96).

A language embedding e [G, gen_dim] goes through Linear(gen_dim -> bottleneck) and a second Linear that emits, per group, either a
convolution kernel or a BatchNorm scale/shift pair.  Both Linears run on the library's GEMM (tiny K: bandwidth bound); the
generated tensors are then consumed by the same grouped conv + BN kernels as stored parameters.  Sub-module and buffer names are the
reference's, the checkpoint layout depends on them."""
import torch
from torch.nn import Linear, Module

from .. import kernels as K


class _TwoStageGenerator(Module):
    """embedding -> bottleneck -> flat parameter rows, one row per language group."""

    def _emit(self, embedding, head, what):
        if embedding.shape[0] != self._groups:
            raise AssertionError(f'{what}: {embedding.shape[0]} generator embeddings for {self._groups} groups')
        hidden = K.linear(embedding, self._bottleneck.weight, self._bottleneck.bias)
        return K.linear(hidden, head.weight, head.bias)


class Conv1dGenerated(_TwoStageGenerator):
    def __init__(self, embedding_dim, bottleneck_dim, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True):
        super().__init__()
        if bias or stride != 1:
            raise AssertionError('generated convolutions are always followed by batch norm (bias=False) and use stride 1')
        self._in_channels, self._out_channels, self._kernel_size = in_channels, out_channels, kernel_size
        self._stride, self._padding, self._dilation, self._groups = stride, padding, dilation, groups
        self._bottleneck = Linear(embedding_dim, bottleneck_dim)
        per_group = out_channels // groups * in_channels // groups * kernel_size      # evaluated left to right like generated.py:31
        self._kernel = Linear(bottleneck_dim, per_group)
        self._bias = None

    def generate(self, generator_embedding):
        """-> kernel [O, I / G, k]: group g owns output channels [g * O/G, (g + 1) * O/G)."""
        flat = self._emit(generator_embedding, self._kernel, 'Conv1dGenerated')
        return flat.view(self._out_channels, self._in_channels // self._groups, self._kernel_size)

    def generate_packed(self, generator_embedding):
        """-> kernel in the implicit-GEMM layout [O, k, I / G], written directly by the generator kernel (mtts_gen_params_fwd);
        None when the shape is outside that kernel's bounds (the caller then uses generate())."""
        G, k = self._groups, self._kernel_size
        Og, Cg = self._out_channels // G, self._in_channels // G
        bott = self._bottleneck.weight.shape[0]
        if generator_embedding.shape[0] != G:
            raise AssertionError(f'Conv1dGenerated: {generator_embedding.shape[0]} generator embeddings for {G} groups')
        if G > 16 or bott > 8 or k > 8 or Og * Cg * k != self._kernel.weight.shape[0]:
            return None
        hidden = K.linear(generator_embedding, self._bottleneck.weight, self._bottleneck.bias)
        return K.generated_kernel(hidden, self._kernel.weight, self._kernel.bias, Og, Cg, k)


class BatchNorm1dGenerated(_TwoStageGenerator):
    def __init__(self, embedding_dim, bottleneck_dim, num_features, groups=1, eps=1e-8, momentum=0.1):
        super().__init__()
        for name, init in (('running_mean', torch.zeros(num_features)), ('running_var', torch.ones(num_features)),
                           ('num_batches_tracked', torch.tensor(0, dtype=torch.long))):
            self.register_buffer(name, init)
        self._groups, self._eps, self._momentum = groups, eps, momentum
        self._num_features = num_features // groups
        self._bottleneck = Linear(embedding_dim, bottleneck_dim)
        self._affine = Linear(bottleneck_dim, 2 * self._num_features)

    def generate(self, generator_embedding):
        """-> (scale [G * C], shift [G * C]); the first half of every generated row scales, the second half shifts."""
        rows = self._emit(generator_embedding, self._affine, 'BatchNorm1dGenerated')
        c = self._num_features
        return rows[:, :c].reshape(-1), rows[:, c:].reshape(-1)
