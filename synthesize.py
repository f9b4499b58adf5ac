#!/usr/bin/env python
"""Synthesis entry point (reference synthesize.py:41-84,91-123) on the MI355X-native hot path.

Reads lines `id|text|speaker|language spec` from stdin (the reference's format) and writes `<id>.npy` mel spectrograms,
de-normalised with the checkpoint's constants like the reference (synthesize.py:41-84).  The text is cleaned and mapped
to symbol ids by multilingual_text_to_speech_amd.data (the reference's utils/text.py rules); phonemisation and
Griffin-Lim vocoding need packages outside the hot path, so with hp.use_phonemes the text must already be phonemes.
`--token_ids` reads space separated ids instead of text.  The language spec keeps the reference's syntax:
`de` | `de-10,fr-9,de` (code switching by character counts) | `fr*0.75:de*0.25` (blend).
"""
import argparse
import os
import sys

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')      # kernel arguments in device memory: -2 % per train step (read when the HIP runtime loads, i.e. before torch)
import numpy as np
import torch


def language_weights(spec, length, languages):
    """Per-character language weight matrix [1, L, NL]; reference synthesize.py:55-70."""
    w = torch.zeros(1, length, len(languages))
    pos = 0
    for part in spec.split(','):
        name, _, count = part.partition('-')
        n = int(count) if count else length - pos
        for blend in name.split(':'):
            lang, _, weight = blend.partition('*')
            w[0, pos:pos + n, languages.index(lang)] = float(weight) if weight else 1.0
        pos += n
    return w


def tokens_of(field, token_ids=False):
    """Symbol ids (EOS appended) of the text field: cleaned text through the symbol table, or literal ids."""
    from multilingual_text_to_speech_amd.params import Params as hp
    from multilingual_text_to_speech_amd import data
    if token_ids:
        return torch.tensor([int(t) for t in field.split()] + [1], dtype=torch.int64)
    # the reference's synthesize() lower-cases whenever hp.case_sensitive is off, phoneme input included (synthesize.py:46-51)
    return torch.tensor(data.to_sequence(data.clean_text(field, False), use_phonemes=hp.use_phonemes), dtype=torch.int64)


def speaker_id(field):
    """Speaker name -> index through hp.unique_speakers (stored in the checkpoint, synthesize.py:72), or a literal index."""
    from multilingual_text_to_speech_amd.params import Params as hp
    names = getattr(hp, 'unique_speakers', None)
    return names.index(field) if names and field in names else int(field)


def denormalize(mel):
    """audio.denormalize_spectrogram for mels (utils/audio.py:111-114); identity when the model was trained un-normalised."""
    from multilingual_text_to_speech_amd.params import Params as hp
    if not hp.normalize_spectrogram or not hasattr(hp, 'mel_normalize_mean'):
        return mel
    return mel * hp.mel_normalize_variance + hp.mel_normalize_mean


def synthesize(model, input_data, force_cpu=False, token_ids=False):
    from multilingual_text_to_speech_amd.params import Params as hp
    item = input_data.strip().split('|')
    ids = tokens_of(item[1], token_ids)
    dev = next(model.parameters()).device
    spk = torch.tensor([speaker_id(item[2])], dtype=torch.int64, device=dev) if hp.multi_speaker else None
    lang = language_weights(item[3], len(ids), hp.languages).to(dev) if hp.multi_language else None
    return item[0], denormalize(model.inference(ids.to(dev), spk, lang).cpu().numpy())


def synthesize_batch(model, lines, token_ids=False):
    """Many input lines at once through Tacotron.inference_batch (one batched decoder run); returns [(id, mel)]."""
    from multilingual_text_to_speech_amd.params import Params as hp
    items = [l.strip().split('|') for l in lines if l.strip()]
    texts = [tokens_of(it[1], token_ids) for it in items]
    spk = [speaker_id(it[2]) for it in items] if hp.multi_speaker else None
    lang = [language_weights(it[3], len(t), hp.languages)[0] for it, t in zip(items, texts)] if hp.multi_language else None
    mels = model.inference_batch(texts, spk, lang)
    return [(it[0], denormalize(m.cpu().numpy())) for it, m in zip(items, mels)]


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", type=str, required=True)
    ap.add_argument("--output", type=str, default=".")
    ap.add_argument("--batch", type=int, default=1, help="utterances per batched decoder run (1 = the reference's loop)")
    ap.add_argument("--token_ids", action="store_true", help="the text field holds space separated symbol ids")
    args = ap.parse_args()
    from multilingual_text_to_speech_amd.utils import build_model
    model = build_model(args.checkpoint).eval()
    from multilingual_text_to_speech_amd.utils import settle_host_heap
    settle_host_heap()      # the garbage collector's full pass over the model's objects now, not in the middle of some utterance's decode loop
    if args.batch > 1:
        lines = [l for l in sys.stdin if l.strip()]
        for i in range(0, len(lines), args.batch):
            for name, mel in synthesize_batch(model, lines[i:i + args.batch], args.token_ids):
                np.save(f'{args.output}/{name}.npy', mel, allow_pickle=False)
    else:
        for line in sys.stdin:
            if line.strip():
                name, mel = synthesize(model, line, token_ids=args.token_ids)
                np.save(f'{args.output}/{name}.npy', mel, allow_pickle=False)
