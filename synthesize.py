#!/usr/bin/env python
"""Synthesis entry point (reference synthesize.py:41-84,91-123) on the MI355X-native hot path.

Reads lines `id|token ids (space separated)|speaker|language spec` from stdin and writes `<id>.npy` mel spectrograms.
The reference's text front end (cleaning, phonemisation) and Griffin-Lim vocoding need packages that are outside the
hot path; this entry point therefore takes token ids and emits (normalised) mels.  The language spec keeps the
reference's syntax: `de` | `de-10,fr-9,de` (code switching by character counts) | `fr*0.75:de*0.25` (blend).
"""
import argparse
import sys

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


def synthesize(model, input_data, force_cpu=False):
    from multilingual_text_to_speech_amd.params import Params as hp
    item = input_data.strip().split('|')
    ids = torch.tensor([int(t) for t in item[1].split()] + [1], dtype=torch.int64)      # + EOS
    dev = next(model.parameters()).device
    spk = torch.tensor([int(item[2])], dtype=torch.int64, device=dev) if hp.multi_speaker else None
    lang = language_weights(item[3], len(ids), hp.languages).to(dev) if hp.multi_language else None
    return item[0], model.inference(ids.to(dev), spk, lang).cpu().numpy()


def synthesize_batch(model, lines):
    """Many input lines at once through Tacotron.inference_batch (one batched decoder run); returns [(id, mel)]."""
    from multilingual_text_to_speech_amd.params import Params as hp
    items = [l.strip().split('|') for l in lines if l.strip()]
    texts = [torch.tensor([int(t) for t in it[1].split()] + [1], dtype=torch.int64) for it in items]
    spk = [int(it[2]) for it in items] if hp.multi_speaker else None
    lang = [language_weights(it[3], len(t), hp.languages)[0] for it, t in zip(items, texts)] if hp.multi_language else None
    mels = model.inference_batch(texts, spk, lang)
    return [(it[0], m.cpu().numpy()) for it, m in zip(items, mels)]


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", type=str, required=True)
    ap.add_argument("--output", type=str, default=".")
    ap.add_argument("--batch", type=int, default=1, help="utterances per batched decoder run (1 = the reference's loop)")
    args = ap.parse_args()
    from multilingual_text_to_speech_amd.utils import build_model
    model = build_model(args.checkpoint).eval()
    if args.batch > 1:
        lines = [l for l in sys.stdin if l.strip()]
        for i in range(0, len(lines), args.batch):
            for name, mel in synthesize_batch(model, lines[i:i + args.batch]):
                np.save(f'{args.output}/{name}.npy', mel, allow_pickle=False)
    else:
        for line in sys.stdin:
            if line.strip():
                name, mel = synthesize(model, line)
                np.save(f'{args.output}/{name}.npy', mel, allow_pickle=False)
