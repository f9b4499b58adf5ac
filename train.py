#!/usr/bin/env python
"""Training entry point with the reference's CLI (train.py:187-197) on the MI355X-native hot path.

    python train.py --hyper_parameters generated_switching --synthetic            # single GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...   # data parallel (RCCL)

The reference's dataset / audio / TensorBoard stack (librosa, phonemizer, ...) is outside the hot path and absent from
this image, so this entry point drives the model with the synthetic batches of SURVEY.md section 8(d) (`--synthetic`,
the default) or with pre-collated tensors saved by the user (`--batches file.pt`: a list of dicts with the keys of
bench.synthetic_batch).  Checkpoints use the reference's dictionary layout (train.py:302-310).
"""
import argparse
import os
import time

import torch

from bench import synthetic_batch, train_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base_directory", type=str, default=".")
    ap.add_argument("--checkpoint", type=str, default=None)
    ap.add_argument("--checkpoint_root", type=str, default="checkpoints")
    ap.add_argument("--hyper_parameters", type=str, default=None, help="preset name or path of a json file")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--chars", type=int, default=120)
    ap.add_argument("--batches", type=str, default=None)
    ap.add_argument("--synthetic", action="store_true", default=True)
    args = ap.parse_args()

    from multilingual_text_to_speech_amd import dist as D
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
    rank, world, local = D.init()
    torch.manual_seed(42)
    ckpt_dir = os.path.join(args.base_directory, args.checkpoint_root)
    os.makedirs(ckpt_dir, exist_ok=True)
    state = None
    if args.checkpoint:
        state = torch.load(os.path.join(ckpt_dir, args.checkpoint), map_location='cpu', weights_only=False)
        hp.load_state_dict(state['parameters'])
    if args.hyper_parameters:
        if args.hyper_parameters in presets.PRESETS:
            presets.apply(args.hyper_parameters, reset=state is None)
        else:
            hp.load(args.hyper_parameters)
    if hp.multi_speaker and not hp.speaker_number:
        hp.speaker_number = 91
    hp.language_number = len(hp.languages) if hp.multi_language else 0
    device = torch.device('cuda', local)
    model = Tacotron().to(device).train()
    opt = torch.optim.Adam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    epoch0 = 0
    if state is not None:
        sd = model.state_dict()
        sd.update({k: v for k, v in state['model'].items() if k in sd})
        model.load_state_dict(sd)
        opt.load_state_dict(state['optimizer'])
        crit.load_state_dict(state['criterion'])
        epoch0 = state['epoch'] + 1
    D.broadcast_parameters(model)
    buckets = D.GradientBuckets(model.parameters()) if world > 1 else None
    G = hp.language_number if hp.encoder_type in ('generated', 'convolutional') else 1
    per = hp.batch_size // world
    D.shard_bounds(per * world, rank, world, G)
    batches = torch.load(args.batches) if args.batches else None
    for step in range(args.steps):
        batch = batches[step % len(batches)] if batches else synthetic_batch(hp, per, args.chars, args.frames, device, seed=step * world + rank)
        t0 = time.time()
        loss = train_step(model, crit, opt, buckets, batch, hp)
        torch.cuda.synchronize()
        if rank == 0:
            print(f'step {step}: loss {loss.item():.4f}  {per * world * args.frames / (time.time() - t0):.0f} frames/s', flush=True)
    if rank == 0:
        path = os.path.join(ckpt_dir, f'{hp.version}_loss-{epoch0}-{loss.item():2.3f}')
        torch.save({'epoch': epoch0, 'model': model.state_dict(), 'optimizer': opt.state_dict(), 'scheduler': {},
                    'parameters': hp.state_dict(), 'criterion': crit.state_dict()}, path)
        print('saved', path)


if __name__ == '__main__':
    main()
