"""Batched synthesis once warm, once inside marker kernels (for rocprofv3 --kernel-trace + scripts/trace_summary.py --region 1):
    python scripts/prof_inference.py [--utterances 128] [--chars 200] [--frames 120]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--utterances', type=int, default=128)
    ap.add_argument('--chars', type=int, default=200)
    ap.add_argument('--frames', type=int, default=120)
    args = ap.parse_args()
    from multilingual_text_to_speech_amd import _C
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    presets.apply('generated_switching', speaker_number=91)
    hp.max_output_length = args.frames
    torch.manual_seed(0)
    model = Tacotron().cuda().eval()
    g = torch.Generator().manual_seed(1)
    L = args.chars + 1
    texts = [torch.cat([torch.randint(3, hp.symbols_count() + 3, (args.chars,), generator=g), torch.tensor([1])]) for _ in range(args.utterances)]
    n_lang = len(hp.languages)
    langs = []
    for i in range(args.utterances):
        w = torch.zeros(L, n_lang); w[:, i % n_lang] = 1.0
        langs.append(w)
    spks = [i % hp.speaker_number for i in range(args.utterances)]
    model.inference_batch(texts, spks, langs, stop_threshold=2.0)
    torch.cuda.synchronize()
    lib = _C.lib()
    _C.check(lib.mtts_prof_marker(1, _C.stream_ptr()), 'marker')
    t0 = time.perf_counter()
    model.inference_batch(texts, spks, langs, stop_threshold=2.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _C.check(lib.mtts_prof_marker(2, _C.stream_ptr()), 'marker')
    torch.cuda.synchronize()
    print('seconds', dt, 'us/step', dt / args.frames * 1e6)


if __name__ == '__main__':
    main()
