"""Convolutional blocks and regularised LSTM cells (parameter holders + HIP forward/backward).

Module/parameter names replicate the reference's `state_dict` layout (modules/layers.py) so that
checkpoints load unchanged; the arithmetic runs in libmtts_hip (conv as implicit MFMA GEMM, fused
BatchNorm/activation/dropout/highway kernels) on channel-last tensors [N, L, C].
"""
import torch
from torch.nn import Sequential, Conv1d, BatchNorm1d, LSTMCell, Module

from .. import kernels as K
from ..masks import provider
from .generated import Conv1dGenerated, BatchNorm1dGenerated


class _Slot(Module):
    """Parameter-free placeholder keeping the reference's Sequential indices (pad / activation / dropout)."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what


class ZoneoutLSTMCell(LSTMCell):
    """LSTMCell parameters + zoneout rates (reference modules/layers.py:18-34); stepped inside mtts_decoder_*."""

    def __init__(self, input_size, hidden_size, zoneout_rate_hidden, zoneout_rate_cell, bias=True):
        super().__init__(input_size, hidden_size, bias)
        self.zoneout_c = zoneout_rate_cell
        self.zoneout_h = zoneout_rate_hidden


class DropoutLSTMCell(LSTMCell):
    """LSTMCell parameters + hidden-state dropout rate (reference modules/layers.py:37-47)."""

    def __init__(self, input_size, hidden_size, dropout_rate, bias=True):
        super().__init__(input_size, hidden_size, bias)
        self._dropout = _Slot(f'dropout p={dropout_rate}')
        self._dropout_rate = dropout_rate


class ConvBlock(Module):
    """pad -> Conv1d(no bias) -> BatchNorm1d -> activation -> dropout; reference modules/layers.py:50-86.
    Input/ output are channel-last [N, L, C]."""

    def __init__(self, input_channels, output_channels, kernel, dropout=0.0, activation='identity', dilation=1, groups=1,
                 batch_norm=True):
        super().__init__()
        assert batch_norm, 'the HIP conv block always carries BatchNorm (every reference call site does)'
        self._groups, self._kernel, self._dilation = groups, kernel, dilation
        self._dropout_rate, self._activation_name = dropout, activation
        self._block = Sequential(_Slot('same padding'),
                                 Conv1d(input_channels, output_channels, kernel, padding=0, dilation=dilation, groups=groups,
                                        bias=False),
                                 BatchNorm1d(output_channels), _Slot(activation), _Slot(f'dropout p={dropout}'))

    def _run(self, x, mask_name, highway):
        conv, bn = self._block[1], self._block[2]
        training = self.training
        mask = None
        if training and self._dropout_rate > 0:
            mask = provider.keep(mask_name, (x.shape[0], x.shape[1], conv.weight.shape[0]), self._dropout_rate, x.device)
        if training:
            bn.num_batches_tracked += 1
        return K.conv_bn_act(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, mask,
                             kernel=self._kernel, dilation=self._dilation, groups=self._groups, act=self._activation_name,
                             training=training, momentum=bn.momentum, eps=bn.eps,
                             mask_scale=1.0 / (1.0 - self._dropout_rate) if self._dropout_rate > 0 else 1.0, highway=highway)

    def forward(self, x, mask_name=None):
        return self._run(x, mask_name, False)


class HighwayConvBlock(ConvBlock):
    """Gated convolution h2*sigmoid(h1) + x*(1-sigmoid(h1)); reference modules/layers.py:134-153."""

    def __init__(self, input_channels, output_channels, kernel, dropout=0.0, activation='identity', dilation=1, groups=1,
                 batch_norm=True):
        super().__init__(input_channels, 2 * output_channels, kernel, dropout, activation, dilation, groups, batch_norm)
        self._gate = _Slot('sigmoid gate')

    def forward(self, x, mask_name=None):
        return self._run(x, mask_name, True)


class ConvBlockGenerated(Module):
    """ConvBlock whose kernel and BatchNorm affine come from the language embedding `e`;
    reference modules/layers.py:89-131."""

    def __init__(self, embedding_dim, bottleneck_dim, input_channels, output_channels, kernel, dropout=0.0,
                 activation='identity', dilation=1, groups=1, batch_norm=True):
        super().__init__()
        assert batch_norm
        self._groups, self._kernel, self._dilation = groups, kernel, dilation
        self._dropout_rate, self._activation_name = dropout, activation
        self._padding = _Slot('same padding')
        self._convolution = Conv1dGenerated(embedding_dim, bottleneck_dim, input_channels, output_channels, kernel, padding=0,
                                            dilation=dilation, groups=groups, bias=False)
        self._regularizer = BatchNorm1dGenerated(embedding_dim, bottleneck_dim, output_channels, groups=groups)
        self._activation = Sequential(_Slot(activation), _Slot(f'dropout p={dropout}'))

    def _run(self, e, x, mask_name, highway):
        training = self.training
        weight = self._convolution.generate_packed(e)           # implicit-GEMM layout straight from the generator kernel
        packed = weight is not None
        if not packed:
            weight = self._convolution.generate(e)
        gamma, beta = self._regularizer.generate(e)
        reg = self._regularizer
        mask = None
        if training and self._dropout_rate > 0:
            mask = provider.keep(mask_name, (x.shape[0], x.shape[1], weight.shape[0]), self._dropout_rate, x.device)
        if training:
            reg.num_batches_tracked += 1
        return K.conv_bn_act(x, weight, gamma, beta, reg.running_mean, reg.running_var, mask, kernel=self._kernel,
                             dilation=self._dilation, groups=self._groups, act=self._activation_name, training=training,
                             momentum=reg._momentum, eps=reg._eps,
                             mask_scale=1.0 / (1.0 - self._dropout_rate) if self._dropout_rate > 0 else 1.0, highway=highway,
                             packed=packed)

    def forward(self, e, x, mask_name=None):
        return self._run(e, x, mask_name, False)


class HighwayConvBlockGenerated(ConvBlockGenerated):
    """reference modules/layers.py:156-178."""

    def __init__(self, embedding_dim, bottleneck_dim, input_channels, output_channels, kernel, dropout=0.0,
                 activation='identity', dilation=1, groups=1, batch_norm=True):
        super().__init__(embedding_dim, bottleneck_dim, input_channels, 2 * output_channels, kernel, dropout, activation,
                         dilation, groups, batch_norm)
        self._gate = _Slot('sigmoid gate')

    def forward(self, e, x, mask_name=None):
        return self._run(e, x, mask_name, True)
