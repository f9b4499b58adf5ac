"""Tacotron 2 assembly with the reference's public API (modules/tacotron2.py:222-408) on the HIP hot path.

`Tacotron()` reads the global `hp`, exposes the same attribute tree and `state_dict` names (including the
aliased `_prenet` / `_attention` entries), the same `forward(...)` 6-tuple and `inference(...)`.  All
arithmetic is dispatched to libmtts_hip; there is no CPU execution path.
"""
import os

import torch
from torch.nn import Sequential, ModuleList, Linear, Embedding, Module

from .. import kernels as K
from .. import decoder_ops as D
from ..masks import provider
from ..params.params import Params as hp
from ..utils import lengths_to_mask
from .layers import ZoneoutLSTMCell, DropoutLSTMCell, ConvBlock, _Slot
from .attention import LocationSensitiveAttention
from .encoder import Encoder, MultiEncoder, ConditionalEncoder, ConvolutionalEncoder, GeneratedConvolutionalEncoder
from .classifier import ReversalClassifier


class Prenet(Module):
    """2x(Linear -> ReLU -> dropout that is ALWAYS on); reference modules/tacotron2.py:15-46.
    Evaluated inside the decoder kernels (batched over teacher frames, or per free-running step)."""

    def __init__(self, input_dim, output_dim, num_layers, dropout):
        super().__init__()
        assert num_layers > 0, 'There must be at least one layer in the pre-net.'
        self._dropout_rate = dropout
        self._activation = _Slot('relu')
        self._layers = ModuleList([Linear(input_dim, output_dim)] + [Linear(output_dim, output_dim) for _ in range(num_layers - 1)])


class Postnet(Module):
    """5 ConvBlocks (tanh x4, identity) + residual on channel-last frames; reference modules/tacotron2.py:49-76."""

    def __init__(self, input_dimension, postnet_dimension, num_blocks, kernel_size, dropout):
        super().__init__()
        if num_blocks < 2:
            raise AssertionError('the post-net needs an input block and an output block at least')
        widths = [input_dimension] + [postnet_dimension] * (num_blocks - 1) + [input_dimension]
        blocks = [ConvBlock(widths[i], widths[i + 1], kernel_size, dropout, 'tanh' if i < num_blocks - 1 else 'identity')
                  for i in range(num_blocks)]
        self._convs = Sequential(*blocks)

    def forward(self, x, x_lengths=None):
        """x [B, T, M] channel-last -> [B, T, M]."""
        residual = x
        for i, block in enumerate(self._convs):
            x = block(x, f'post.{i}')
        return x + residual


_WARNED_SHAPES = set()


def _warn_if_not_persistent(decoder_dim, attention_dim, memory_dim, kernel_size):
    """The persistent (weights-stationary) decoder kernels are laid out for the widths every reference configuration uses: decoder 1024,
    attention 128, memory width a multiple of 32 up to 768, odd location kernel up to 31 taps (csrc/persist.hip: pdec_supported /
    pgen_supported).  Any other model still runs - on the per-step launch schedule, at roughly 2.5x the decoder time - and used to do so
    silently; say it once per shape."""
    why = []
    if decoder_dim != 1024:
        why.append(f'decoder_dimension {decoder_dim} != 1024')
    if attention_dim != 128:
        why.append(f'attention_dimension {attention_dim} != 128')
    if memory_dim % 32 != 0 or memory_dim > 768:
        why.append(f'memory width {memory_dim} (encoder + speaker + language embedding) is not a multiple of 32 up to 768')
    if kernel_size % 2 == 0 or kernel_size > 31:
        why.append(f'attention_kernel_size {kernel_size} is even or above 31')
    key = (decoder_dim, attention_dim, memory_dim, kernel_size)
    if why and key not in _WARNED_SHAPES:
        _WARNED_SHAPES.add(key)
        import warnings
        warnings.warn('multilingual_text_to_speech_amd: the persistent decoder kernels do not take this model (' + '; '.join(why) +
                      '): the teacher-forced decoder will run on the per-step launch schedule (about 2.5x its time at batch 64)')


class Decoder(Module):
    """Attention + 2 LSTM cells + frame/stop projections; reference modules/tacotron2.py:79-219."""

    def __init__(self, output_dim, decoder_dim, attention, generator_rnn, attention_rnn, context_dim, prenet, prenet_dim,
                 max_frames):
        super().__init__()
        self._output_dim, self._decoder_dim, self._max_frames = output_dim, decoder_dim, max_frames
        # sub-module registration order = the reference's state_dict order
        self._prenet, self._attention = prenet, attention
        self._attention_lstm, self._generator_lstm = attention_rnn, generator_rnn
        projection_in = context_dim + decoder_dim                      # cat(h_gen, context)
        self._frame_prediction = Linear(projection_in, output_dim)
        self._stop_prediction = Linear(projection_in, 1)
        # rows appended to every memory position (tacotron2.py:143-146): one table per conditioning signal that is switched on
        self._speaker_embedding = self._get_embedding(hp.speaker_embedding_dimension, hp.speaker_number) \
            if hp.multi_speaker and hp.speaker_embedding_dimension > 0 else None
        self._language_embedding = self._get_embedding(hp.language_embedding_dimension, len(hp.languages)) \
            if hp.multi_language and hp.language_embedding_dimension > 0 else None
        _warn_if_not_persistent(decoder_dim, hp.attention_dimension, context_dim, hp.attention_kernel_size)

    @staticmethod
    def _get_embedding(embedding_dimension, size=None):
        table = Embedding(size, embedding_dimension)
        torch.nn.init.xavier_uniform_(table.weight)
        return table

    def _memory(self, encoded_input, speaker, language):
        """Concatenate speaker / language embedding rows to the encoder output (tacotron2.py:143-146,158-161)."""
        parts = [encoded_input]
        if hp.multi_speaker and self._speaker_embedding is not None:
            parts.append(K.embedding(self._speaker_embedding.weight, speaker))
        if hp.multi_language and self._language_embedding is not None:
            parts.append(K.embedding(self._language_embedding.weight, language))
        return torch.cat(parts, dim=-1) if len(parts) > 1 else encoded_input

    def _cfg(self):
        zone = hp.decoder_regularization == 'zoneout'
        return dict(training=self.training, zone=zone, p_prenet=float(self._prenet._dropout_rate),
                    p_hidden=float(hp.zoneout_hidden if zone else hp.dropout_hidden), p_cell=float(hp.zoneout_cell))

    def _step_masks(self, T, B, device):
        """uint8 keep flags for the decoder loop: prenet (always) and LSTM hidden-state regularisation (training)."""
        P, H = hp.prenet_dimension, self._decoder_dim
        masks = {}
        for i in range(len(self._prenet._layers)):
            masks[f'prenet.{i}'] = provider.keep(f'dec.prenet.{i}', (T, B, P), self._prenet._dropout_rate, device)
        if self.training:
            if hp.decoder_regularization == 'zoneout':
                masks['att_h'] = provider.keep('dec.att_lstm.h', (T, B, H), hp.zoneout_hidden, device)
                masks['att_c'] = provider.keep('dec.att_lstm.c', (T, B, H), hp.zoneout_cell, device)
                masks['gen_h'] = provider.keep('dec.gen_lstm.h', (T, B, H), hp.zoneout_hidden, device)
                masks['gen_c'] = provider.keep('dec.gen_lstm.c', (T, B, H), hp.zoneout_cell, device)
            else:
                masks['att_h'] = provider.keep('dec.att_lstm', (T, B, H), hp.dropout_hidden, device)
                masks['gen_h'] = provider.keep('dec.gen_lstm', (T, B, H), hp.dropout_hidden, device)
        return masks

    def forward(self, encoded_input, encoded_lenghts, target, teacher_forcing_ratio, speaker, language):
        """target [B, M, T] (reference layout) -> spectrogram [B,T,M], stop [B,T], alignments [B,T,L]."""
        memory = self._memory(encoded_input, speaker, language)
        B, T = target.shape[0], target.shape[2]
        teacher = provider.teacher(T, teacher_forcing_ratio)
        masks = self._step_masks(T, B, memory.device)
        w = D.decoder_weights(self, self._attention, self._prenet)
        tgt = target.transpose(1, 2).contiguous()
        return D.decode_train(memory, tgt, encoded_lenghts, teacher, masks, self._cfg(), w)

    def inference(self, encoded_input, speaker, language, lengths=None, stop_threshold=0.5):
        memory = self._memory(encoded_input, speaker, language)
        B, L = memory.shape[0], memory.shape[1]
        if lengths is None:
            lengths = torch.full((B,), L, dtype=torch.int64)
        if provider.injected is not None:      # tests inject the reference's draws for all max_frames steps
            masks = self._step_masks(self._max_frames, B, memory.device)
        else:                                   # buffers and draws grow with the utterance (not hp.max_output_length up front)
            masks = lambda T: self._step_masks(T, B, memory.device)
        w = D.decoder_weights(self, self._attention, self._prenet)
        with torch.no_grad():
            # MTTS_DECODE_GRAPH=1: chunks of the free-running loop replay as hipGraphs from the third call with the same shapes on
            # (mtts_decoder_fwd_graphed; needs max_output_length <= 2048 so that the whole range has fixed buffers)
            frames, _, _, n = D.decode_free(memory, lengths, w, self._cfg(), masks, self._max_frames, hp.stop_frames,
                                            stop_threshold=stop_threshold, graph=os.environ.get('MTTS_DECODE_GRAPH', '0') == '1')
        return frames, n


class Tacotron(Module):
    """reference modules/tacotron2.py:222-408."""

    def __init__(self):
        super().__init__()
        other_symbols = 3   # PAD, EOS, UNK
        self._embedding = Embedding(hp.symbols_count() + other_symbols, hp.embedding_dimension, padding_idx=0)
        torch.nn.init.xavier_uniform_(self._embedding.weight)
        self._encoder = self._get_encoder(hp.encoder_type)
        if hp.reversal_classifier:
            self._reversal_classifier = self._get_adversarial_classifier(hp.reversal_classifier_type)
        self._prenet = Prenet(hp.num_mels, hp.prenet_dimension, hp.prenet_layers, hp.dropout)
        # memory width Dm = encoder output + the embedding rows the decoder appends per position
        memory_dim = hp.encoder_dimension + (hp.speaker_embedding_dimension if hp.multi_speaker else 0) + \
            (hp.language_embedding_dimension if hp.multi_language else 0)
        self._attention = self._get_attention(hp.attention_type, memory_dim)
        H = hp.decoder_dimension

        def cell(input_dim):       # the generator cell is created first, like the reference (parameter-initialisation order)
            if hp.decoder_regularization == 'zoneout':
                return ZoneoutLSTMCell(input_dim, H, hp.zoneout_hidden, hp.zoneout_cell)
            return DropoutLSTMCell(input_dim, H, hp.dropout_hidden)
        generator_rnn = cell(memory_dim + H)                      # input = [h_att, context]
        attention_rnn = cell(memory_dim + hp.prenet_dimension)     # input = [prenet(frame), context]
        self._decoder = Decoder(hp.num_mels, H, self._attention, generator_rnn, attention_rnn, memory_dim, self._prenet,
                                hp.prenet_dimension, hp.max_output_length)
        self._postnet = self._get_postnet("cbhg" if hp.predict_linear else "conv")

    def _get_encoder(self, name):
        """hp.encoder_type -> encoder (tacotron2.py:300-317).  The conv-stack encoders share one argument tuple; the grouped
        ones get one group per language."""
        stack = (hp.embedding_dimension, hp.encoder_dimension, hp.encoder_blocks, hp.encoder_kernel_size, hp.dropout)
        n_groups = hp.language_number if hp.multi_language else 1
        builders = {
            'simple': lambda: Encoder(*stack),
            'separate': lambda: MultiEncoder(hp.language_number, stack),
            'shared': lambda: ConditionalEncoder(hp.language_number, hp.input_language_embedding, stack),
            'convolutional': lambda: ConvolutionalEncoder(hp.embedding_dimension, hp.encoder_dimension, 0.05, n_groups),
            'generated': lambda: GeneratedConvolutionalEncoder(hp.embedding_dimension, hp.encoder_dimension, 0.05, hp.generator_dim,
                                                               hp.generator_bottleneck_dim, groups=n_groups),
        }
        if name not in builders:
            raise ValueError(f'unknown encoder_type {name!r}')
        return builders[name]()

    def _get_adversarial_classifier(self, name):
        if name == "reversal":
            return ReversalClassifier(hp.encoder_dimension, hp.reversal_classifier_dim, hp.speaker_number,
                                      hp.reversal_gradient_clipping)
        raise NotImplementedError("only the 'reversal' classifier is provided ('cosine' does not converge per the reference)")

    def _get_attention(self, name, memory_dimension):
        if name == "location_sensitive":
            return LocationSensitiveAttention(hp.attention_kernel_size, hp.attention_location_dimension, False,
                                              hp.attention_dimension, hp.decoder_dimension, memory_dimension)
        raise NotImplementedError("only 'location_sensitive' attention is provided (the reference marks the others undebugged)")

    def _get_postnet(self, name):
        if name == "conv":
            return Postnet(hp.num_mels, hp.postnet_dimension, hp.postnet_blocks, hp.postnet_kernel_size, hp.dropout)
        raise NotImplementedError('predict_linear=True (CBHG post-net) is outside the MI355X hot path')

    @staticmethod
    def _per_position(ids, length):
        """One id per utterance [B] -> the same id at every input position [B, L]; other ranks pass through."""
        return ids.unsqueeze(1).expand(-1, length) if ids is not None and ids.dim() == 1 else ids

    def forward(self, text, text_length, target, target_length, speakers, languages, teacher_forcing_ratio=0.0):
        speakers = self._per_position(speakers, text.size(1))
        languages = self._per_position(languages, text.size(1))

        embedded = K.embedding(self._embedding.weight, text, padding_idx=0)
        encoded = self._encoder(embedded, text_length, languages)
        encoder_output = encoded
        speaker_prediction = self._reversal_classifier(encoded) if hp.reversal_classifier else None

        if languages is not None and languages.dim() == 3:
            languages = torch.argmax(languages, dim=2)
        prediction, stop_token, alignment = self._decoder(encoded, text_length, target, teacher_forcing_ratio, speakers, languages)
        post = self._postnet(prediction, target_length)                    # channel-last [B,T,M]
        pre_prediction = prediction.transpose(1, 2)
        post_prediction = post.transpose(1, 2)

        target_mask = lengths_to_mask(K.to_device_async(target_length, stop_token.device), target.size(2))
        stop_token = stop_token.masked_fill(~target_mask, 1000)
        target_mask = target_mask.unsqueeze(1).float()
        pre_prediction = pre_prediction * target_mask
        post_prediction = post_prediction * target_mask
        return post_prediction, pre_prediction, stop_token, alignment, speaker_prediction, encoder_output

    def inference(self, text, speaker=None, language=None):
        """Batch-1 synthesis with the reference's semantics (mutates `text` in place like tacotron2.py:389)."""
        text.unsqueeze_(0)
        speaker = self._per_position(speaker, text.size(1))
        language = self._per_position(language, text.size(1))
        with torch.no_grad():
            embedded = K.embedding(self._embedding.weight, text, padding_idx=0)
            encoded = self._encoder(embedded, torch.LongTensor([text.size(1)]), language)
            if language is not None and language.dim() == 3:
                language = torch.argmax(language, dim=2)
            frames, n = self._decoder.inference(encoded, speaker, language)
            post = self._postnet(frames[:, :n[0]].contiguous(), None)
        K.check_device_errors(post.device)         # an id outside an embedding table must not pass silently
        return post.transpose(1, 2).squeeze(0)


    def inference_batch(self, texts, speakers=None, languages=None, stop_threshold=0.5):
        """Batched synthesis with the batch-1 reference semantics per utterance.

        texts: list of 1-D int64 token tensors (EOS included); speakers: list of speaker ids or None; languages: list of
        per-character weight matrices [L_i, n_languages] (synthesize.language_weights) or None.
        Encoder and post-net run once per bucket of equal length (no padding inside a bucket, so 'same' convolutions and
        BatchNorm see exactly what the batch-1 call sees); the autoregressive decoder - the expensive part - runs ONCE
        for all utterances over zero-padded memories with per-sample lengths and the reference's per-sample stop rule.
        Returns a list of [num_mels, n_i] spectrograms in input order."""
        dev = self._embedding.weight.device
        n_utt = len(texts)
        lens = [int(t.numel()) for t in texts]
        Lmax = max(lens)
        grouped = hp.encoder_type in ('convolutional', 'generated')
        blended = hp.encoder_type in ('convolutional', 'generated', 'separate')
        with torch.no_grad():
            memory_in, lang = None, None
            for L in sorted(set(lens)):
                idx = [i for i in range(n_utt) if lens[i] == L]
                # one host -> device copy per bucket (not per utterance), one scatter of the bucket's rows into the padded memory
                text = torch.stack([texts[i].cpu() for i in idx]).to(dev)
                lw = torch.stack([languages[i].cpu().reshape(L, -1) for i in idx]).to(dev) if languages is not None else None
                emb = K.embedding(self._embedding.weight, text, padding_idx=0)
                tl = torch.full((len(idx),), L, dtype=torch.int64)
                enc = self._encoder(emb, tl, lw, blend=True) if blended else self._encoder(emb, tl, lw)
                rows = torch.as_tensor(idx, dtype=torch.int64, device=dev)
                if memory_in is None:
                    memory_in = torch.zeros(n_utt, Lmax, enc.shape[-1], device=dev)
                    lang = torch.zeros(n_utt, Lmax, dtype=torch.int64, device=dev) if languages is not None else None
                memory_in[rows, :L] = enc
                if lang is not None:
                    lang[rows, :L] = torch.argmax(lw, dim=2)
            spk = None
            if speakers is not None:
                spk = torch.as_tensor([int(v) for v in speakers], dtype=torch.int64, device=dev).unsqueeze(1).expand(-1, Lmax)
            frames, n = self._decoder.inference(memory_in, spk, lang, lengths=torch.as_tensor(lens, dtype=torch.int64),
                                                stop_threshold=stop_threshold)
            out = [None] * n_utt
            for nf in sorted(set(n)):
                idx = [i for i in range(n_utt) if n[i] == nf]
                post = self._postnet(frames[idx, :nf].contiguous(), None)
                for j, i in enumerate(idx):
                    out[i] = post[j].transpose(0, 1)
        K.check_device_errors(dev)
        return out


class TacotronLoss(Module):
    """Loss wrapper with the reference's API and state (modules/tacotron2.py:411-485)."""

    def __init__(self, guided_att_steps, guided_att_variance, guided_att_gamma):
        super().__init__()
        self._g = guided_att_variance
        self._gamma = guided_att_gamma
        self._g_steps = guided_att_steps

    def load_state_dict(self, d):
        for k, v in d.items():
            setattr(self, k, v)

    def state_dict(self):
        return {"_g": self._g, "_g_steps": self._g_steps}

    def update_states(self):
        self._g *= self._gamma
        self._g_steps = max(0, self._g_steps - 1)

    def forward(self, source_length, target_length, pre_prediction, pre_target, post_prediction, post_target, stop, target_stop,
                alignment, speaker, speaker_prediction, encoder_outputs, classifier):
        pre_target.requires_grad = False
        post_target.requires_grad = False
        target_stop.requires_grad = False
        # all four terms and their gradients in one pass of the library's loss kernel (mtts_tacotron_loss); no torch fallback
        from ..optim import TacotronLossFn
        ga_on = bool(hp.guided_attention_loss) and self._g_steps > 0
        v = TacotronLossFn.apply(pre_prediction, post_prediction, stop, alignment if hp.guided_attention_loss else None,
                                 pre_target, target_stop, source_length, target_length, self._g, ga_on, 100.0,
                                 None if post_target is pre_target else post_target)
        losses = {'mel_pre': v[0], 'mel_pos': v[1], 'stop_token': v[2]}
        total = v[4]
        if hp.guided_attention_loss:
            losses['guided_att'] = v[3] if ga_on else 0
        if hp.reversal_classifier:
            losses['lang_class'] = ReversalClassifier.loss(source_length, speaker, speaker_prediction) * \
                (hp.reversal_classifier_w / (hp.num_mels + 2))
            if torch.is_grad_enabled() and speaker_prediction.requires_grad:
                # data-parallel TRAINING step only (every rank holds a shard of ONE global batch and calls this exactly once per
                # step): the reference's CE is a mean over the GLOBAL batch's valid characters.  Evaluation deals whole, unrelated
                # batches to the ranks (train.evaluate) - no collective there: ranks make different numbers of calls.
                from ..dist import global_mean_scale
                losses['lang_class'] = losses['lang_class'] * global_mean_scale(source_length.sum(), speaker_prediction.device)[0]
            total = total + losses['lang_class']
        return total, losses
