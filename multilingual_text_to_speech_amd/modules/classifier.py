"""Adversarial speaker classifier with gradient reversal; reference modules/classifier.py:6-69."""
import torch
from torch.nn import Sequential, Linear, Module

from .. import kernels as K


class GradientReversalFunction(torch.autograd.Function):
    """forward identity; backward clamp(g, +-c) * (-l); reference modules/classifier.py:6-18."""

    @staticmethod
    def forward(ctx, x, l, c):
        ctx.l, ctx.c = l, c
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return K.grad_reverse_clamp(grad_output, ctx.l, ctx.c), None, None


class ReversalClassifier(Module):
    def __init__(self, input_dim, hidden_dim, output_dim, gradient_clipping_bounds, scale_factor=1.0):
        super().__init__()
        self._lambda = scale_factor
        self._clipping = gradient_clipping_bounds
        self._output_dim = output_dim
        self._classifier = Sequential(Linear(input_dim, hidden_dim), Linear(hidden_dim, output_dim))

    def forward(self, x):
        x = GradientReversalFunction.apply(x, self._lambda, self._clipping)
        x = K.linear(x, self._classifier[0].weight, self._classifier[0].bias)
        return K.linear(x, self._classifier[1].weight, self._classifier[1].bias)

    @staticmethod
    def loss(input_lengths, speakers, prediction, embeddings=None):
        """Cross entropy over valid characters (padding -> ignore_index); reference modules/classifier.py:62-69."""
        from ..optim import MaskedCrossEntropyFn       # value + gradient in one HIP kernel (mtts_masked_cross_entropy); GPU tensors only
        return MaskedCrossEntropyFn.apply(prediction, speakers, input_lengths, 1.0)
