"""Secondary measurement (BASELINE configs[4]): batched autoregressive synthesis, generated_switching architecture,
128 utterances x 200 characters (+EOS), frame count pinned to 600 (stop rule disabled), random-init weights, fp32.
Prints one JSON line; `python scripts/bench_inference.py [--utterances 128] [--chars 200] [--frames 600]`."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--utterances', type=int, default=128)
    ap.add_argument('--chars', type=int, default=200)
    ap.add_argument('--frames', type=int, default=600)
    ap.add_argument('--preset', default='generated_switching')
    ap.add_argument('--repeats', type=int, default=3)
    args = ap.parse_args()
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    presets.apply(args.preset, speaker_number=91)
    hp.max_output_length = args.frames
    torch.manual_seed(0)
    dev = torch.device('cuda')
    model = Tacotron().to(dev).eval()
    g = torch.Generator().manual_seed(1)
    L = args.chars + 1
    texts = [torch.cat([torch.randint(3, hp.symbols_count() + 3, (args.chars,), generator=g), torch.tensor([1])]) for _ in range(args.utterances)]
    n_lang = len(hp.languages)
    langs = None
    if hp.multi_language:
        langs = []
        for i in range(args.utterances):
            w = torch.zeros(L, n_lang); w[:, i % n_lang] = 1.0
            langs.append(w)
    spks = [i % hp.speaker_number for i in range(args.utterances)] if hp.multi_speaker else None
    times = []
    for r in range(args.repeats + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model.inference_batch(texts, spks, langs, stop_threshold=2.0)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    assert all(o.shape == (hp.num_mels, args.frames) for o in out), out[0].shape
    dt = sorted(times[1:])[len(times[1:]) // 2]
    print(json.dumps({'metric': 'mel-frames/sec (batched synthesis: encoder + %d-step free-running decoder + post-net)' % args.frames,
                      'value': round(args.utterances * args.frames / dt, 1), 'unit': 'mel-frames/s', 'n_gpus': 1,
                      'seconds_per_batch': round(dt, 4), 'us_per_decoder_step': round(dt / args.frames * 1e6, 1), 'dtype': 'f32', 'data': 'synthetic',
                      'config': {'workload': f'params/{args.preset} synthesis, {args.utterances} utterances x {L} tokens -> {args.frames} frames, '
                                             'stop rule disabled, random-init weights'}}))


if __name__ == '__main__':
    main()
