"""CPU oracle for the text->mel hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional, single-threaded-friendly restatement (torch CPU fp32 tensors, explicit
formulas, explicit dropout masks) of the reference model's arithmetic.  It is the *checker* for the
HIP path: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  Nothing under `multilingual_text_to_speech_amd/` imports it, and the product path raises when
the HIP library is missing instead of falling back to this code.

Pinning: the reference ships no tests or golden vectors ("parity unpinned by the
reference"), so the oracle is pinned against the reference implementation itself, executed in this
container by `oracle/make_golden.py` (imports /root/reference, records inputs / dropout masks /
outputs / gradients into `tests/golden/*.pt`).  `tests/test_oracle_golden.py` replays those fixtures
through this file on CPU.

Every function cites the reference file:line it restates.  Layouts follow the reference
(channel-first `[B, C, L]` inside conv stacks) so that fixtures recorded from the reference can be
replayed verbatim; the HIP path uses channel-last and the tests transpose.

Dropout: the reference calls `F.dropout` / `nn.Dropout`; here every dropout site takes an explicit
multiplier tensor `masks[name]` (values 0 or 1/(1-p)); `None`/missing means "no dropout".
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------------

# bf16-operand mode (tests of the product's bf16 path only; the reference itself is fp32): the product rounds the OPERANDS of its
# GEMM-class contractions to bf16 (RNE) and keeps everything else in fp32.  `BF16_SITES` names the contraction sites
# that round; `_lin` / `_conv` apply the same rounding here, so that an oracle run differs from the product only by summation order.
# Sites: 'conv' (conv blocks incl. post-net), 'bilstm_in' (BiLSTM input projection), 'lstm' (decoder LSTM gates, both operands),
# 'memory' (memory projection), 'loc' (the location filter bank U = W_loc W_conv is formed from rounded factors), 'prenet',
# 'proj' (frame / stop projection), 'linear' (classifier / generator bottlenecks).  Never rounded (fp32 in the product): BiLSTM
# recurrent product, query projection, U applied to the cumulative alignment, energies, softmax, context.
BF16_SITES = frozenset()


def _r(x, site):
    return x.to(torch.bfloat16).to(x.dtype) if site in BF16_SITES else x


# Backward of the bf16-operand mode.  The product's BATCHED backward GEMMs round their operands as well (dY, and the W / X they multiply it
# with); since round 5 so do the per-step input-gradient products of the decoder LSTMs in the all-teacher-forced schedule (the tests
# then list 'lstm' in BF16_BWD_SITES); the attention backward and the general schedule's per-step products stay fp32.  `BF16_BWD_SITES` lists the sites whose
# input AND weight gradient come from rounded operands; for the sites in `BF16_BWD_WGRAD_ONLY` (the decoder LSTMs: weight gradients are
# batched GEMMs, input gradients per-step fp32 products) only the weight gradient does.  Empty (default): plain autograd through the
# forward rounding (straight-through).  Test infrastructure only, like everything in this file.
BF16_BWD_SITES = frozenset()
BF16_BWD_WGRAD_ONLY = frozenset()


def _rb(x):
    return x.to(torch.bfloat16).to(x.dtype)          # dtype-generic: the fp64 run of tests/test_gpu_more.py rounds the same operands


class _RoundedLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, wgrad_only):
        xr, wr = _rb(x), _rb(w)
        ctx.save_for_backward(xr, wr)
        ctx.wgrad_only = wgrad_only
        return F.linear(xr, wr)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = _rb(dy)
        dx = (dy if ctx.wgrad_only else dyr) @ wr
        dw = dyr.reshape(-1, dyr.shape[-1]).t() @ xr.reshape(-1, xr.shape[-1])
        return dx, dw, None


class _RoundedConv1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, dilation, groups):
        xr, wr = _rb(x), _rb(w)
        ctx.save_for_backward(xr, wr)
        ctx.cfg = (dilation, groups)
        return F.conv1d(xr, wr, None, 1, 0, dilation, groups)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dilation, groups = ctx.cfg
        dyr = _rb(dy)
        dx = torch.nn.grad.conv1d_input(xr.shape, wr, dyr, 1, 0, dilation, groups)
        dw = torch.nn.grad.conv1d_weight(xr, wr.shape, dyr, 1, 0, dilation, groups)
        return dx, dw, None, None


def _lin(site, x, w, b=None):
    if site in BF16_SITES and (site in BF16_BWD_SITES or site in BF16_BWD_WGRAD_ONLY):
        y = _RoundedLinear.apply(x, w, site in BF16_BWD_WGRAD_ONLY)
        return y if b is None else y + b
    return F.linear(_r(x, site), _r(w, site), b)


def _conv(site, x, w, *args):
    if site in BF16_SITES and site in BF16_BWD_SITES:
        bias, stride, padding, dilation, groups = args
        assert bias is None and stride == 1 and padding == 0
        return _RoundedConv1d.apply(x, w, dilation, groups)
    return F.conv1d(_r(x, site), _r(w, site), *args)


def _mask(masks, name, x):
    m = None if masks is None else masks.get(name)
    return x if m is None else x * m


def lengths_to_mask(lengths, max_length=None):
    """reference utils/__init__.py:7-10"""
    ml = int(torch.max(lengths)) if max_length is None else max_length
    return torch.arange(ml)[None, :] < lengths[:, None]


def _act(name, x):
    """reference modules/layers.py:8-15"""
    if name == 'relu':
        return torch.relu(x)
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    return x


def _same_pad(x, kernel, dilation):
    """reference modules/layers.py:72-74 (ConstantPad1d; even kernels pad (p, p+1))"""
    p = (kernel - 1) * dilation // 2
    return F.pad(x, (p, p) if kernel % 2 else (p, p + 1))


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum, eps, stats_out=None, key=None):
    """F.batch_norm semantics (reference modules/generated.py:94-96, modules/layers.py:78):
    training -> biased batch variance normalises, running stats get the *unbiased* variance."""
    if training:
        n = x.shape[0] * x.shape[2]
        mean = x.mean(dim=(0, 2))
        var = x.var(dim=(0, 2), unbiased=False)
        if stats_out is not None:
            stats_out[key + '.running_mean'] = (1 - momentum) * running_mean + momentum * mean.detach()
            stats_out[key + '.running_var'] = (1 - momentum) * running_var + momentum * var.detach() * n / max(n - 1, 1)
    else:
        mean, var = running_mean, running_var
    y = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + eps)
    return y * weight[None, :, None] + bias[None, :, None]


# --------------------------------------------------------------------------------------------------
# conv blocks (reference modules/layers.py:50-178, modules/generated.py:7-96)
# --------------------------------------------------------------------------------------------------

def conv_block(sd, prefix, x, kernel, activation, masks, mask_name, training, dilation=1, groups=1,
               stats_out=None):
    """ConvBlock: pad -> Conv1d(bias=False) -> BatchNorm1d(eps 1e-5, momentum .1) -> act -> dropout.
    reference modules/layers.py:66-86; Sequential indices: 0 pad, 1 conv, 2 bn."""
    x = _same_pad(x, kernel, dilation)
    x = _conv('conv', x, sd[prefix + '._block.1.weight'], None, 1, 0, dilation, groups)
    x = batch_norm(x, sd[prefix + '._block.2.weight'], sd[prefix + '._block.2.bias'],
                   sd[prefix + '._block.2.running_mean'], sd[prefix + '._block.2.running_var'],
                   training, 0.1, 1e-5, stats_out, prefix + '._block.2')
    x = _act(activation, x)
    return _mask(masks, mask_name, x)


def highway_combine(h, x, groups):
    """reference modules/layers.py:149-153 / :174-178: even chunks gate, odd chunks value."""
    chunks = torch.chunk(h, 2 * groups, 1)
    h1 = torch.cat(chunks[0::2], 1)
    h2 = torch.cat(chunks[1::2], 1)
    p = torch.sigmoid(h1)
    return h2 * p + x * (1.0 - p)


def highway_conv_block(sd, prefix, x, kernel, masks, mask_name, training, dilation, groups, stats_out=None):
    """HighwayConvBlock (reference modules/layers.py:134-153)."""
    h = conv_block(sd, prefix, x, kernel, 'identity', masks, mask_name, training, dilation, groups, stats_out)
    return highway_combine(h, x, groups)


def generated_conv_block(sd, prefix, e, x, in_ch, out_ch, kernel, activation, masks, mask_name, training,
                         dilation, groups, stats_out=None):
    """ConvBlockGenerated (reference modules/layers.py:89-131) with Conv1dGenerated
    (modules/generated.py:34-42) and BatchNorm1dGenerated (modules/generated.py:71-96, eps 1e-8)."""
    x = _same_pad(x, kernel, dilation)
    cp = prefix + '._convolution'
    # bf16-operand mode: the bottleneck Linears and the BN affine Linear are library GEMMs (site 'linear'), the kernel generator
    # itself (mtts_gen_params_fwd) is an fp32 kernel - not rounded -, the grouped convolution is a GEMM (site 'conv')
    eb = _lin('linear', e, sd[cp + '._bottleneck.weight'], sd[cp + '._bottleneck.bias'])
    w = F.linear(eb, sd[cp + '._kernel.weight'], sd[cp + '._kernel.bias']).view(out_ch, in_ch // groups, kernel)
    x = _conv('conv', x, w, None, 1, 0, dilation, groups)
    rp = prefix + '._regularizer'
    er = _lin('linear', e, sd[rp + '._bottleneck.weight'], sd[rp + '._bottleneck.bias'])
    affine = _lin('linear', er, sd[rp + '._affine.weight'], sd[rp + '._affine.bias'])
    nf = out_ch // groups
    scale = affine[:, :nf].contiguous().view(-1)
    bias = affine[:, nf:].contiguous().view(-1)
    x = batch_norm(x, scale, bias, sd[rp + '.running_mean'], sd[rp + '.running_var'], training, 0.1, 1e-8,
                   stats_out, rp)
    x = _act(activation, x)
    return _mask(masks, mask_name, x)


# --------------------------------------------------------------------------------------------------
# LSTM pieces
# --------------------------------------------------------------------------------------------------

def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh, site_x='lstm', site_h='lstm'):
    """torch.nn.LSTMCell arithmetic (gate order i, f, g, o) - reference modules/layers.py:18,37."""
    gates = _lin(site_x, x, w_ih, b_ih) + _lin(site_h, h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, 1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h_new = torch.sigmoid(o) * torch.tanh(c_new)
    return h_new, c_new


def bilstm_packed(sd, prefix, x, lengths):
    """nn.LSTM(bidirectional, batch_first) over a packed sequence + pad_packed (zeros beyond length).
    reference modules/encoder.py:41-44.  x [B, L, C] -> [B, L, 2*H]."""
    B, L, _ = x.shape
    H = sd[prefix + '.weight_hh_l0'].shape[1]
    out = x.new_zeros(B, L, 2 * H)
    for d, sfx in enumerate(('', '_reverse')):
        w_ih, w_hh = sd[prefix + '.weight_ih_l0' + sfx], sd[prefix + '.weight_hh_l0' + sfx]
        b_ih, b_hh = sd[prefix + '.bias_ih_l0' + sfx], sd[prefix + '.bias_hh_l0' + sfx]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        steps = range(L) if d == 0 else range(L - 1, -1, -1)
        outs = [None] * L
        for t in steps:
            valid = (lengths > t).to(x.dtype)[:, None]
            h_new, c_new = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh, 'bilstm_in', 'bilstm_rec')
            h = valid * h_new + (1 - valid) * h
            c = valid * c_new + (1 - valid) * c
            outs[t] = h * valid
        out = out.clone()
        out[:, :, d * H:(d + 1) * H] = torch.stack(outs, 1)
    return out


# --------------------------------------------------------------------------------------------------
# encoders (reference modules/encoder.py)
# --------------------------------------------------------------------------------------------------

def simple_encoder(sd, prefix, cfg, x, lengths, masks, training, stats_out=None, mask_prefix='enc'):
    """Encoder.forward, reference modules/encoder.py:35-45."""
    x = x.transpose(1, 2)
    for i in range(cfg['encoder_blocks']):
        x = conv_block(sd, f'{prefix}._convs.{i}', x, cfg['encoder_kernel_size'], 'relu', masks,
                       f'{mask_prefix}.{i}', training, stats_out=stats_out)
    x = x.transpose(1, 2)
    return bilstm_packed(sd, prefix + '._lstm', x, lengths)


# (kernel, dilation, highway) of the 14 layers - reference modules/encoder.py:180-191 / :123-130
CONV_ENCODER_LAYERS = [(1, 1, False), (1, 1, False)] + [(3, 3 ** i, True) for i in range(4)] * 2 + \
                      [(3, 1, True)] * 2 + [(1, 1, True)] * 2


def grouped_encoder(sd, prefix, cfg, x, x_langs, masks, training, generated, stats_out=None):
    """ConvolutionalEncoder.forward (reference modules/encoder.py:134-156) and
    GeneratedConvolutionalEncoder.forward (:196-221)."""
    G = cfg['language_number'] if cfg['multi_language'] else 1
    cin, cout = cfg['embedding_dimension'], cfg['encoder_dimension']
    single = x_langs is not None and x_langs.shape[0] == 1
    if single:
        x = x.expand((G, -1, -1))
    e = sd[prefix + '._embedding.weight'] if generated else None
    bs = x.shape[0]
    x = x.transpose(1, 2).reshape(bs // G, G * cin, -1)
    for i, (k, dil, highway) in enumerate(CONV_ENCODER_LAYERS):
        ic = G * (cin if i == 0 else cout)
        oc = G * cout * (2 if highway else 1)
        act = 'relu' if i == 0 else 'identity'
        lp = f'{prefix}._layers.{i}'
        if generated:
            h = generated_conv_block(sd, lp, e, x, ic, oc, k, act, masks, f'enc.{i}', training, dil, G, stats_out)
        else:
            h = conv_block(sd, lp, x, k, act, masks, f'enc.{i}', training, dil, G, stats_out)
        x = highway_combine(h, x, G) if highway else h
    x = x.reshape(bs, cout, -1).transpose(1, 2)
    if single:
        # reference modules/encoder.py:213-219 (normaliser uses only the first batch element's row sums)
        xr = x.new_zeros(1, x.shape[1], x.shape[2])
        norm = x_langs / x_langs.sum(2, keepdim=True)[0]
        for l in range(G):
            xr[0] = xr[0] + norm[0, :, l].reshape(-1, 1) * x[l]
        x = xr
    return x


def encode(sd, cfg, embedded, text_length, languages, masks, training, stats_out=None):
    """Tacotron._get_encoder dispatch, reference modules/tacotron2.py:286-304."""
    t = cfg['encoder_type']
    if t == 'simple':
        return simple_encoder(sd, '_encoder', cfg, embedded, text_length, masks, training, stats_out)
    if t == 'shared':
        # ConditionalEncoder, reference modules/encoder.py:67-71
        ids = torch.argmax(languages, dim=2)
        emb = F.embedding(ids, sd['_encoder._language_embedding.weight'])
        return simple_encoder(sd, '_encoder._encoder', cfg, torch.cat((embedded, emb), -1), text_length, masks,
                              training, stats_out)
    if t == 'separate':
        # MultiEncoder, reference modules/encoder.py:87-97
        xs = None
        norm = languages / languages.sum(2, keepdim=True)[0]
        for l in range(cfg['language_number']):
            w = norm[:, :, l].reshape(-1, 1)
            if not w.bool().any():
                continue
            ex = simple_encoder(sd, f'_encoder._encoders.{l}', cfg, embedded, text_length, masks, training,
                                stats_out, mask_prefix=f'enc{l}')
            ex = ex * w.reshape(ex.shape[0], ex.shape[1], 1)
            xs = ex if xs is None else xs + ex
        return xs
    return grouped_encoder(sd, '_encoder', cfg, embedded, languages, masks, training, t == 'generated', stats_out)


# --------------------------------------------------------------------------------------------------
# attention + decoder (reference modules/attention.py, modules/tacotron2.py:79-219)
# --------------------------------------------------------------------------------------------------

def prenet(sd, x, masks, name):
    """Prenet.forward: Linear -> ReLU -> dropout(always on), reference modules/tacotron2.py:37-46."""
    i = 0
    while f'_prenet._layers.{i}.weight' in sd:
        x = torch.relu(_lin('prenet', x, sd[f'_prenet._layers.{i}.weight'], sd[f'_prenet._layers.{i}.bias']))
        x = _mask(masks, f'{name}.{i}', x)
        i += 1
    return x


def lsa_step(sd, query, memory, memory_transform, cum_weights, mask):
    """One LocationSensitiveAttention step.  reference modules/attention.py:39-45 (forward),
    :67-74 (_attent), :76-83 (_normalize), :85-86 (_combine_weights)."""
    p = '_attention'
    q = F.linear(query, sd[p + '._query.weight']).unsqueeze(1)                          # [B,1,A]
    ksz = sd[p + '._loc_features.weight'].shape[2]
    if 'loc' in BF16_SITES:
        # product: the two location layers are folded into ONE filter bank U = W_loc W_conv by a (rounded-operand) GEMM, then U is
        # applied to the cumulative alignment in fp32
        U = _r(sd[p + '._location.weight'], 'loc') @ _r(sd[p + '._loc_features.weight'].squeeze(1), 'loc')       # [A, ksz]
        loc = F.conv1d(cum_weights.unsqueeze(1), U.unsqueeze(1), None, 1, (ksz - 1) // 2).transpose(1, 2)
    else:
        loc = F.conv1d(cum_weights.unsqueeze(1), sd[p + '._loc_features.weight'], None, 1, (ksz - 1) // 2)
        loc = F.linear(loc.transpose(1, 2), sd[p + '._location.weight'])                    # [B,L,A]
    energy = torch.tanh(q + memory_transform + loc + sd[p + '._bias'])
    energy = F.linear(energy, sd[p + '._energy.weight']).squeeze(-1)                    # [B,L]
    energy = energy.masked_fill(~mask, float('-inf'))
    weights = F.softmax(energy, dim=1)
    context = torch.bmm(weights.unsqueeze(1), memory).squeeze(1)
    return context, weights, cum_weights + weights


def decoder_cell(sd, name, cfg, x, h, c, masks, mask_name, training):
    """DropoutLSTMCell / ZoneoutLSTMCell, reference modules/layers.py:18-47."""
    p = f'_decoder.{name}'
    h_new, c_new = lstm_cell(x, h, c, sd[p + '.weight_ih'], sd[p + '.weight_hh'], sd[p + '.bias_ih'],
                             sd[p + '.bias_hh'])
    if cfg['decoder_regularization'] == 'zoneout':
        zh, zc = cfg['zoneout_hidden'], cfg['zoneout_cell']
        if training:
            h_new = (1 - zh) * _mask(masks, mask_name + '.h', h_new - h) + h
            c_new = (1 - zc) * _mask(masks, mask_name + '.c', c_new - c) + c
        else:
            h_new = zh * h + (1 - zh) * h_new
            c_new = zc * c + (1 - zc) * c_new
    elif training:
        h_new = _mask(masks, mask_name, h_new)
    return h_new, c_new


def decode(sd, cfg, encoded, mask, target, teacher, speaker, language, masks, training, max_frames=None,
           stop_rule=False):
    """Decoder._decode, reference modules/tacotron2.py:148-209.

    target [B, M, T] or None (free running); teacher = bool [T] (the `torch.rand(T) > 1 - tf` draw of
    :171 made explicit).  Per-step masks: masks['att_lstm'][i], masks['gen_lstm'][i] ([T,B,H]) and for
    free-running prenet masks['prenet_step.{layer}'][i]."""
    B, L, _ = encoded.shape
    M, H = cfg['num_mels'], cfg['decoder_dimension']
    if cfg['multi_speaker'] and '_decoder._speaker_embedding.weight' in sd:
        encoded = torch.cat((encoded, F.embedding(speaker, sd['_decoder._speaker_embedding.weight'])), -1)
    if cfg['multi_language'] and '_decoder._language_embedding.weight' in sd:
        encoded = torch.cat((encoded, F.embedding(language, sd['_decoder._language_embedding.weight'])), -1)
    memory_transform = _lin('memory', encoded, sd['_attention._memory.weight'])
    cum = encoded.new_zeros(B, L)
    context = encoded.new_zeros(B, encoded.shape[2])
    h_att = encoded.new_zeros(B, H); c_att = encoded.new_zeros(B, H)
    h_gen = encoded.new_zeros(B, H); c_gen = encoded.new_zeros(B, H)
    frame = encoded.new_zeros(B, M)
    inference = target is None
    T = max_frames if inference else target.shape[2]
    if not inference:
        tgt = torch.cat((encoded.new_zeros(B, 1, M), target.transpose(1, 2)), 1)       # :129-131
        tgt = prenet(sd, tgt, masks, 'prenet')
    frames, stops, aligns = [], [], []
    stop_frames = -1
    for i in range(T):
        if inference or not bool(teacher[i]):
            step_masks = None if masks is None else {k: v[i] for k, v in masks.items() if k.startswith('prenet_step')}
            prev = prenet(sd, frame, step_masks, 'prenet_step')
        else:
            prev = tgt[:, i]
        sm = None if masks is None else {k: v[i] for k, v in masks.items() if k.startswith(('att_lstm', 'gen_lstm'))}
        h_att, c_att = decoder_cell(sd, '_attention_lstm', cfg, torch.cat((prev, context), 1), h_att, c_att, sm,
                                    'att_lstm', training)
        context, weights, cum = lsa_step(sd, h_att, encoded, memory_transform, cum, mask)
        h_gen, c_gen = decoder_cell(sd, '_generator_lstm', cfg, torch.cat((h_att, context), 1), h_gen, c_gen, sm,
                                    'gen_lstm', training)
        proto = torch.cat((h_gen, context), 1)
        frame = _lin('proj', proto, sd['_decoder._frame_prediction.weight'], sd['_decoder._frame_prediction.bias'])
        stop = _lin('proj', proto, sd['_decoder._stop_prediction.weight'], sd['_decoder._stop_prediction.bias'])
        frames.append(frame); stops.append(stop); aligns.append(weights)
        if inference and stop_rule and bool(torch.sigmoid(stop).ge(0.5).all()):          # :201-207 (batch 1)
            if stop_frames == -1:
                stop_frames = cfg['stop_frames']
                continue
            stop_frames -= 1
            if stop_frames == 0:
                break
    return torch.stack(frames, 1), torch.stack(stops, 1).squeeze(2), torch.stack(aligns, 1)


def postnet(sd, cfg, x, masks, training, stats_out=None):
    """Postnet.forward, reference modules/tacotron2.py:66-76 (tanh x (n-1), identity last, residual)."""
    n = cfg['postnet_blocks']
    r = x
    for i in range(n):
        x = conv_block(sd, f'_postnet._convs.{i}', x, cfg['postnet_kernel_size'],
                       'tanh' if i < n - 1 else 'identity', masks, f'post.{i}', training, stats_out=stats_out)
    return x + r


def reversal_classifier(sd, x):
    """ReversalClassifier.forward (forward = identity, two Linears), reference modules/classifier.py:57-60."""
    x = _lin('linear', x, sd['_reversal_classifier._classifier.0.weight'], sd['_reversal_classifier._classifier.0.bias'])
    return _lin('linear', x, sd['_reversal_classifier._classifier.1.weight'], sd['_reversal_classifier._classifier.1.bias'])


class _GradReverse(torch.autograd.Function):
    """GradientReversalFunction, reference modules/classifier.py:6-18."""

    @staticmethod
    def forward(ctx, x, l, c):
        ctx.l, ctx.c = l, c
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.l * g.clamp(-ctx.c, ctx.c).neg(), None, None


def tacotron_forward(sd, cfg, text, text_length, target, target_length, speakers, languages, teacher,
                     masks=None, training=True):
    """Tacotron.forward, reference modules/tacotron2.py:355-385.  Returns a dict with the reference's
    six outputs plus `bn_stats` (updated running statistics, training mode only)."""
    stats = {}
    if speakers is not None and speakers.dim() == 1:
        speakers = speakers.unsqueeze(1).expand((-1, text.size(1)))
    if languages is not None and languages.dim() == 1:
        languages = languages.unsqueeze(1).expand((-1, text.size(1)))
    embedded = F.embedding(text, sd['_embedding.weight'], padding_idx=0)
    encoded = encode(sd, cfg, embedded, text_length, languages, masks, training, stats)
    spk_pred = None
    if cfg['reversal_classifier']:
        spk_pred = reversal_classifier(sd, _GradReverse.apply(encoded, 1.0, cfg['reversal_gradient_clipping']))
    if languages is not None and languages.dim() == 3:
        languages = torch.argmax(languages, dim=2)
    mask = lengths_to_mask(text_length, encoded.size(1))
    pred, stop, align = decode(sd, cfg, encoded, mask, target, teacher, speakers, languages, masks, training)
    pre = pred.transpose(1, 2)
    post = postnet(sd, cfg, pre, masks, training, stats)
    tmask = lengths_to_mask(target_length, target.size(2))
    stop = stop.masked_fill(~tmask, 1000)
    tm = tmask.unsqueeze(1).to(pre.dtype)
    return dict(post=post * tm, pre=pre * tm, stop=stop, alignment=align, speaker_prediction=spk_pred,
                encoder_output=encoded, bn_stats=stats)


# --------------------------------------------------------------------------------------------------
# loss (reference modules/tacotron2.py:443-485, modules/classifier.py:62-69)
# --------------------------------------------------------------------------------------------------

def guided_attention(alignments, input_lengths, target_lengths, g):
    """TacotronLoss._guided_attention, reference modules/tacotron2.py:443-457."""
    weights = torch.zeros_like(alignments)
    for i, (f, l) in enumerate(zip(target_lengths.tolist(), input_lengths.tolist())):
        gf = torch.arange(f, dtype=alignments.dtype)[:, None]
        gl = torch.arange(l, dtype=alignments.dtype)[None, :]
        weights[i, :f, :l] = 1 - torch.exp(-(gl / l - gf / f) ** 2 / (2 * g ** 2))
    loss = torch.sum(weights * alignments, dim=(1, 2))
    return torch.mean(loss / target_lengths.to(loss.dtype))


def tacotron_loss(cfg, out, source_length, target_length, mel_target, stop_target, speakers, g, g_steps=1):
    """TacotronLoss.forward, reference modules/tacotron2.py:459-485."""
    M = cfg['num_mels']
    losses = {
        'mel_pre': 2 * F.mse_loss(out['pre'], mel_target),
        'mel_pos': F.mse_loss(out['post'], mel_target),
        'stop_token': F.binary_cross_entropy_with_logits(out['stop'], stop_target,
                                                         pos_weight=torch.tensor([100.0], dtype=out['stop'].dtype)) / (M + 2),
    }
    if cfg['reversal_classifier']:
        ml = int(torch.max(source_length))
        im = torch.arange(ml)[None, :] < source_length[:, None]
        tgt = speakers.repeat(ml, 1).transpose(0, 1).clone()
        tgt[~im] = -100
        ce = F.cross_entropy(out['speaker_prediction'].transpose(1, 2), tgt, ignore_index=-100)
        losses['lang_class'] = ce * cfg['reversal_classifier_w'] / (M + 2)
    if cfg['guided_attention_loss'] and g_steps > 0:
        losses['guided_att'] = guided_attention(out['alignment'], source_length, target_length, g)
    return sum(losses.values()), losses


def cfg_from_params(hp):
    """Snapshot the hyper-parameters the oracle needs from a Params-like class."""
    keys = ['encoder_type', 'encoder_blocks', 'encoder_kernel_size', 'embedding_dimension', 'encoder_dimension',
            'multi_language', 'multi_speaker', 'language_number', 'num_mels', 'decoder_dimension',
            'decoder_regularization', 'zoneout_hidden', 'zoneout_cell', 'postnet_blocks', 'postnet_kernel_size',
            'reversal_classifier', 'reversal_gradient_clipping', 'reversal_classifier_w', 'stop_frames',
            'guided_attention_loss', 'dropout', 'dropout_hidden', 'prenet_dimension', 'attention_dimension',
            'speaker_embedding_dimension', 'language_embedding_dimension', 'generator_dim',
            'generator_bottleneck_dim', 'input_language_embedding', 'speaker_number']
    return {k: getattr(hp, k) for k in keys}
