"""Encoders (reference modules/encoder.py) on channel-last tensors.

simple / shared / separate: conv stack + BiLSTM (HIP conv+BN kernels, HIP recurrent BiLSTM);
convolutional / generated: 14 grouped (highway) conv blocks where sample b belongs to language group b mod G.
"""
import torch
from torch.nn import Sequential, ModuleList, LSTM, Embedding, Module

from .. import kernels as K
from ..params.params import Params as hp
from .layers import ConvBlock, HighwayConvBlock, ConvBlockGenerated, HighwayConvBlockGenerated


class Encoder(Module):
    """Vanilla Tacotron 2 encoder: 3x ConvBlock(k=5, relu) + BiLSTM; reference modules/encoder.py:9-45."""

    def __init__(self, input_dim, output_dim, num_blocks, kernel_size, dropout, generated=False):
        super().__init__()
        assert num_blocks > 0, 'There must be at least one convolutional block in the encoder.'
        assert output_dim % 2 == 0, 'Bidirectional LSTM output dimension must be divisible by 2.'
        convs = [ConvBlock(input_dim, output_dim, kernel_size, dropout, 'relu')] + \
                [ConvBlock(output_dim, output_dim, kernel_size, dropout, 'relu') for _ in range(num_blocks - 1)]
        self._convs = Sequential(*convs)
        self._lstm = LSTM(output_dim, output_dim // 2, batch_first=True, bidirectional=True)

    def forward(self, x, x_lenghts, x_langs=None, mask_prefix='enc'):
        for i, block in enumerate(self._convs):
            x = block(x, f'{mask_prefix}.{i}')
        l = self._lstm
        return K.bilstm(x, x_lenghts, (l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0,
                                       l.weight_ih_l0_reverse, l.weight_hh_l0_reverse, l.bias_ih_l0_reverse,
                                       l.bias_hh_l0_reverse))


class ConditionalEncoder(Module):
    """Language embedding concatenated to every character embedding; reference modules/encoder.py:48-71."""

    def __init__(self, num_langs, langs_embedding_dim, encoder_args):
        super().__init__()
        self._language_embedding = Embedding(num_langs, langs_embedding_dim)
        encoder_args = list(encoder_args)
        encoder_args[0] += langs_embedding_dim
        self._encoder = Encoder(*encoder_args)

    def forward(self, x, x_lenghts, x_langs):
        ids = torch.argmax(x_langs, dim=2)
        l = K.embedding(self._language_embedding.weight, ids)
        return self._encoder(torch.cat((x, l), dim=-1), x_lenghts)


class MultiEncoder(Module):
    """One vanilla encoder per language, outputs blended by per-character language weights;
    reference modules/encoder.py:74-97 (which only broadcasts correctly for batch size 1)."""

    def __init__(self, num_langs, encoder_args):
        super().__init__()
        self._num_langs = num_langs
        self._encoders = ModuleList([Encoder(*encoder_args) for _ in range(num_langs)])

    def forward(self, x, x_lenghts, x_langs, blend=False):
        xs = None
        # the reference divides by the FIRST sample's sums (it is only ever blended at batch 1); batched synthesis
        # (blend=True) normalises every utterance by its own sums, i.e. the batch-1 result per utterance
        x_langs_normed = x_langs / (x_langs.sum(2, keepdim=True) if blend else x_langs.sum(2, keepdim=True)[0])
        for l in range(self._num_langs):
            w = x_langs_normed[:, :, l]
            if not bool(w.bool().any()):
                continue
            ex = self._encoders[l](x, x_lenghts, mask_prefix=f'enc{l}')
            ex = ex * w.reshape(-1, 1).reshape(ex.shape[0], ex.shape[1], 1) if ex.shape[0] == 1 else ex * w.unsqueeze(-1)
            xs = ex if xs is None else xs + ex
        return xs


def _to_groups(x, groups, channels):
    """[B, L, C] with sample b in group b mod G -> [B/G, L, G*C] (reference reshape modules/encoder.py:206-208)."""
    bs, L = x.shape[0], x.shape[1]
    return x.reshape(bs // groups, groups, L, channels).permute(0, 2, 1, 3).reshape(bs // groups, L, groups * channels)


def _from_groups(x, groups, channels):
    n, L = x.shape[0], x.shape[1]
    return x.reshape(n, L, groups, channels).permute(0, 2, 1, 3).reshape(n * groups, L, channels)


def _expand_groups(x, groups):
    """Every utterance replicated into all G language groups: row b*G + g belongs to group g."""
    B, L, C = x.shape
    return x.unsqueeze(1).expand(B, groups, L, C).reshape(B * groups, L, C)


def _blend_batch(x, x_langs, groups):
    """Batched synthesis: rows b*G + g of x are utterance b through group g; mix by the utterance's own weights."""
    BG, L, C = x.shape
    norm = x_langs / x_langs.sum(2, keepdim=True)                       # [B, L, G]
    return (x.reshape(BG // groups, groups, L, C) * norm.permute(0, 2, 1).unsqueeze(-1)).sum(1)


def _blend_groups(x, x_langs, groups):
    """Batch-1 inference: mix the G group outputs by per-character language weights (modules/encoder.py:213-219)."""
    norm = x_langs / x_langs.sum(2, keepdim=True)[0]
    xr = torch.zeros(1, x.shape[1], x.shape[2], device=x.device)
    for l in range(groups):
        xr[0] = xr[0] + norm[0, :, l].reshape(-1, 1) * x[l]
    return xr


def _pure_languages(x_langs):
    """Batched synthesis: the language id of every utterance whose per-character weights name ONE language for all its characters
    (plain multilingual synthesis), or None when some utterance mixes languages (code switching)."""
    nz = x_langs != 0
    if not bool((nz.sum(2) == 1).all()):
        return None
    ids = torch.argmax(nz.to(torch.int8), dim=2)                          # [B, L]
    if not bool((ids == ids[:, :1]).all()):
        return None
    return ids[:, 0]


def _compact_groups(x, lang, groups):
    """Utterance b goes through its OWN language group only: rows are laid out as (slot r, group g) -> row r * G + g like a training
    batch (sample i belongs to group i mod G), groups with fewer utterances are padded with zero rows.  Returns the compact batch and
    the row of every utterance in it.  In eval mode the blocks have no cross-sample coupling (BatchNorm uses its running statistics),
    so an utterance's rows are exactly the rows the all-groups expansion computes for its language - the other G - 1 copies, which the
    blend multiplies by zero, are simply not computed (5x less encoder work for 5 languages)."""
    B = x.shape[0]
    order = torch.argsort(lang, stable=True)
    counts = torch.bincount(lang, minlength=groups)
    n_max = int(counts.max())
    starts = torch.cumsum(counts, 0) - counts
    slot = torch.arange(B, device=x.device) - starts[lang[order]]         # position of the utterance inside its group
    rows = torch.empty(B, dtype=torch.int64, device=x.device)
    rows[order] = slot * groups + lang[order]
    xc = x.new_zeros(n_max * groups, x.shape[1], x.shape[2])
    xc[rows] = x
    return xc, rows


_LAYERS = [(1, 1, False), (1, 1, False)] + [(3, 3 ** i, True) for i in range(4)] * 2 + [(3, 1, True)] * 2 + [(1, 1, True)] * 2


class ConvolutionalEncoder(Module):
    """Grouped fully-convolutional encoder; reference modules/encoder.py:100-156."""

    def __init__(self, input_dim, output_dim, dropout, groups=1):
        super().__init__()
        self._groups, self._input_dim, self._output_dim = groups, input_dim, output_dim
        i, o = input_dim * groups, output_dim * groups
        layers = []
        for n, (k, dil, highway) in enumerate(_LAYERS):
            cls = HighwayConvBlock if highway else ConvBlock
            layers.append(cls(i if n == 0 else o, o, k, dropout, activation='relu' if n == 0 else 'identity', dilation=dil,
                              groups=groups))
        self._layers = Sequential(*layers)

    def forward(self, x, x_lenghts=None, x_langs=None, blend=False):
        single = x_langs is not None and x_langs.shape[0] == 1 and not blend
        rows = None
        if single:
            x = x.expand((self._groups, -1, -1))
        elif blend:
            lang = _pure_languages(x_langs) if not self.training else None
            if lang is not None:
                x, rows = _compact_groups(x, lang, self._groups)
            else:
                x = _expand_groups(x, self._groups)
        x = _to_groups(x.contiguous(), self._groups, self._input_dim)
        for n, layer in enumerate(self._layers):
            x = layer(x, f'enc.{n}')
        x = _from_groups(x, self._groups, self._output_dim)
        if rows is not None:
            return x[rows]
        if blend:
            return _blend_batch(x, x_langs, self._groups)
        return _blend_groups(x, x_langs, self._groups) if single else x


class GeneratedConvolutionalEncoder(Module):
    """Grouped convolutional encoder with generated weights; reference modules/encoder.py:159-221."""

    def __init__(self, input_dim, output_dim, dropout, embedding_dim, bottleneck_dim, groups=1):
        super().__init__()
        self._groups, self._input_dim, self._output_dim = groups, input_dim, output_dim
        i, o = input_dim * groups, output_dim * groups
        layers = []
        for n, (k, dil, highway) in enumerate(_LAYERS):
            cls = HighwayConvBlockGenerated if highway else ConvBlockGenerated
            layers.append(cls(embedding_dim, bottleneck_dim, i if n == 0 else o, o, k, dropout=dropout,
                              activation='relu' if n == 0 else 'identity', dilation=dil, groups=groups))
        self._layers = Sequential(*layers)
        self._embedding = Embedding(groups, embedding_dim)

    def forward(self, x, x_lenghts=None, x_langs=None, blend=False):
        single = x_langs is not None and x_langs.shape[0] == 1 and not blend
        rows = None
        if single:
            x = x.expand((self._groups, -1, -1))
        elif blend:
            lang = _pure_languages(x_langs) if not self.training else None
            if lang is not None:
                x, rows = _compact_groups(x, lang, self._groups)
            else:
                x = _expand_groups(x, self._groups)
        e = K.embedding(self._embedding.weight, torch.arange(self._groups, device=x.device))
        x = _to_groups(x.contiguous(), self._groups, self._input_dim)
        for n, layer in enumerate(self._layers):
            x = layer(e, x, f'enc.{n}')
        x = _from_groups(x, self._groups, self._output_dim)
        if rows is not None:
            return x[rows]
        if blend:
            return _blend_batch(x, x_langs, self._groups)
        return _blend_groups(x, x_langs, self._groups) if single else x
