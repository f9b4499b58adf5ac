#!/usr/bin/env python
"""Training entry point with the reference's CLI (train.py:187-197) on the MI355X-native hot path.

    python train.py --hyper_parameters generated_switching --synthetic            # single GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...   # data parallel (RCCL)

The reference's audio / phonemiser / TensorBoard stack (librosa, phonemizer, ...) is outside the hot path and absent from
this image.  With `--data_root DIR` (containing train.txt [+ val.txt] meta-files and cached mel `.npy` files, the
reference's on-disk format) the loop runs epochs over multilingual_text_to_speech_amd.data (language-ordered, per-rank
sharded batches; StepLR and checkpoint cadence of train.py:260-310).  Otherwise it drives the model with the synthetic
batches of SURVEY.md section 8(d) or with pre-collated tensors (`--batches file.pt`: a list of dicts with the keys of
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
    ap.add_argument("--data_root", type=str, default=None, help="directory with train.txt / val.txt and cached spectrograms")
    ap.add_argument("--loader_workers", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=None, help="override hp.epochs (dataset mode)")
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
    from multilingual_text_to_speech_amd.optim import FusedAdam     # torch.optim.Adam's state_dict, one fused clip+update launch set
    opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
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
    buckets = D.GradientBuckets(model.parameters(), overlap=os.environ.get('MTTS_DDP_OVERLAP', '1') != '0') if world > 1 else None
    G = hp.language_number if hp.encoder_type in ('generated', 'convolutional') else 1
    per = hp.batch_size // world
    D.shard_bounds(per * world, rank, world, G)
    if args.data_root:
        return train_on_dataset(args, hp, model, opt, crit, buckets, rank, world, device, epoch0, ckpt_dir, state)
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


def train_on_dataset(args, hp, model, opt, crit, buckets, rank, world, device, epoch0, ckpt_dir, state):
    """Epoch loop of the reference (train.py:218-310) over cached spectrograms, one process per GPU."""
    from torch.utils.data import DataLoader
    from multilingual_text_to_speech_amd import data as DT
    train_set = DT.MelDataset(os.path.join(args.data_root, 'train.txt'), args.data_root)
    val_path = os.path.join(args.data_root, 'val.txt')
    val_set = DT.MelDataset(val_path, args.data_root, train_set.unique_speakers) if os.path.exists(val_path) else None
    grouped = hp.encoder_type in ('generated', 'convolutional')
    if hp.normalize_spectrogram and state is None:      # train.py:246-250; restored from the checkpoint's parameters otherwise
        hp.mel_normalize_mean, hp.mel_normalize_variance = train_set.get_normalization_constants()

    def loader(ds, shuffle, drop_last):
        if grouped:
            sampler = DT.PerfectBatchSampler(ds, hp.languages, hp.batch_size, shuffle=shuffle, drop_last=drop_last, rank=rank, world=world)
        else:       # one "language" bucket = plain (optionally balanced) batches, sharded the same way
            class _Flat:
                items = [{'language': 0}] * len(ds)
            sampler = DT.PerfectBatchSampler(_Flat(), [None], hp.batch_size, shuffle=shuffle, drop_last=drop_last, rank=rank, world=world)
        return DataLoader(ds, batch_sampler=sampler, collate_fn=DT.Collate(not grouped), num_workers=args.loader_workers), sampler

    train_data, train_sampler = loader(train_set, True, True)
    # StepLR counted in epochs like the reference (train.py:266-271,296-297)
    step_size = max(1, hp.learning_rate_decay_each // max(1, len(train_data)))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size, hp.learning_rate_decay)
    if state is not None and state.get('scheduler'):
        sched.load_state_dict(state['scheduler'])
    for epoch in range(epoch0, args.epochs if args.epochs is not None else hp.epochs):
        train_sampler.set_epoch(epoch)
        model.train()
        t0, frames = time.time(), 0
        for collated in train_data:
            batch = DT.batch_to_device(collated, device)
            loss = train_step(model, crit, opt, buckets, batch, hp)
            frames += int(batch['target_length'].sum())
        torch.cuda.synchronize()
        if hp.learning_rate_decay_start - hp.learning_rate_decay_each < epoch * len(train_data):
            sched.step()
        if rank == 0:
            print(f'epoch {epoch}: loss {loss.item():.4f}  {frames * world / (time.time() - t0):.0f} frames/s', flush=True)
            if (epoch + 1) % hp.checkpoint_each_epochs == 0:
                DT.save_checkpoint(os.path.join(ckpt_dir, f'{hp.version}_loss-{epoch}-{loss.item():2.3f}'), epoch, model, opt, sched, crit)


if __name__ == '__main__':
    main()
