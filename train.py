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
import math
import os
import time

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')      # kernel arguments in device memory: -2 % per train step (read when the HIP runtime loads, i.e. before torch)
import torch

from bench import synthetic_batch, train_step
from multilingual_text_to_speech_amd.utils import settle_host_heap


def cos_decay(global_step, decay_steps):
    """Teacher-forcing schedule of the reference (train.py:18-26)."""
    global_step = min(global_step, decay_steps)
    return 0.5 * (1 + math.cos(math.pi * global_step / decay_steps))


def teacher_forcing_ratio(hp, global_step):
    """train.py:58-60: constant, or cosine decay that starts after hp.teacher_forcing_start_steps."""
    if hp.constant_teacher_forcing:
        return hp.teacher_forcing
    return cos_decay(max(global_step - hp.teacher_forcing_start_steps, 0), hp.teacher_forcing_steps)


def make_optimizer(hp, model):
    """Adam (L2-coupled decay); with hp.encoder_optimizer the encoder gets its own learning rate (train.py:260-270)."""
    from multilingual_text_to_speech_amd.optim import FusedAdam     # torch.optim.Adam's state_dict, fused clip + update
    if not hp.encoder_optimizer:
        return FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
    encoder_params = list(model._encoder.parameters())
    other_params = list(model._decoder.parameters()) + list(model._postnet.parameters()) + list(model._prenet.parameters()) + \
        list(model._embedding.parameters()) + list(model._attention.parameters())
    if hp.reversal_classifier:
        other_params += list(model._reversal_classifier.parameters())
    seen, unique_other = set(id(p) for p in encoder_params), []
    for p in other_params:          # _decoder aliases the prenet / attention modules: every tensor once
        if id(p) not in seen:
            seen.add(id(p))
            unique_other.append(p)
    opt = FusedAdam([{'params': unique_other}, {'params': encoder_params, 'lr': hp.learning_rate_encoder}],
                    lr=hp.learning_rate, weight_decay=hp.weight_decay)
    return opt


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
    hp.language_number = len(hp.languages) if hp.multi_language else 0
    datasets = None
    if args.data_root:
        # the corpus sizes the model (train.py:238-250): speaker table / classifier width, speaker names, mel statistics
        datasets = open_datasets(args, hp, state)
    elif hp.multi_speaker and not getattr(hp, 'speaker_number', 0):
        hp.speaker_number = 91          # synthetic batches: the speaker count of data/css_comvoi
    device = torch.device('cuda', local)
    model = Tacotron().to(device).train()
    opt = make_optimizer(hp, model)
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
    if datasets is not None:
        return train_on_dataset(args, hp, datasets, model, opt, crit, buckets, rank, world, device, epoch0, ckpt_dir, state)
    from multilingual_text_to_speech_amd.kernels import check_device_errors
    batches = torch.load(args.batches) if args.batches else None
    for step in range(args.steps):
        batch = batches[step % len(batches)] if batches else synthetic_batch(hp, per, args.chars, args.frames, device, seed=step * world + rank)
        t0 = time.time()
        loss = train_step(model, crit, opt, buckets, batch, hp, teacher_forcing_ratio(hp, step))
        if step == 1:
            settle_host_heap()           # steady state: the garbage collector's first full pass happens here, not 85 ms into some later step
        torch.cuda.synchronize()
        check_device_errors(device)
        if rank == 0:
            print(f'step {step}: loss {loss.item():.4f}  {per * world * args.frames / (time.time() - t0):.0f} frames/s', flush=True)
    if rank == 0:
        path = os.path.join(ckpt_dir, f'{hp.version}_loss-{epoch0}-{loss.item():2.3f}')
        torch.save({'epoch': epoch0, 'model': model.state_dict(), 'optimizer': opt.state_dict(), 'scheduler': {},
                    'parameters': hp.state_dict(), 'criterion': crit.state_dict()}, path)
        print('saved', path)


def open_datasets(args, hp, state):
    """Train / validation collections and the hyper-parameters the reference derives from them BEFORE building the model
    (train.py:238-250): speaker count, speaker names (kept from the checkpoint on resume), mel normalisation constants."""
    from multilingual_text_to_speech_amd import data as DT
    known = list(getattr(hp, 'unique_speakers', [])) if state is not None else []
    train_set = DT.MelDataset(os.path.join(args.data_root, 'train.txt'), args.data_root, known)
    val_path = os.path.join(args.data_root, 'val.txt')
    val_set = DT.MelDataset(val_path, args.data_root, train_set.unique_speakers) if os.path.exists(val_path) else None
    hp.speaker_number = 0 if not hp.multi_speaker else train_set.get_num_speakers()
    if hp.multi_speaker and state is None:
        hp.unique_speakers = list(train_set.unique_speakers)
    if state is not None and hp.multi_speaker and train_set.get_num_speakers() != len(known):
        raise SystemExit(f'the corpus has {train_set.get_num_speakers()} speakers but the checkpoint was trained with {len(known)}')
    if hp.normalize_spectrogram and state is None:      # restored from the checkpoint's parameters otherwise
        hp.mel_normalize_mean, hp.mel_normalize_variance = train_set.get_normalization_constants()
    return train_set, val_set


def make_loader(hp, ds, train, rank, world, workers):
    """Sampler selection of train.py:225-236.  Perfect (language-ordered) batches when hp.perfect_sampling (or a grouped encoder,
    which cannot run on anything else); with-replacement language-balanced draws when hp.balanced_sampling; plain shuffled
    batches otherwise.  Every sampler yields this rank's contiguous shard of the GLOBAL batch, padded to the global max T."""
    from torch.utils.data import DataLoader
    from multilingual_text_to_speech_amd import data as DT
    grouped = hp.encoder_type in ('generated', 'convolutional')
    balanced = bool(hp.multi_language and hp.balanced_sampling)
    if grouped or (balanced and hp.perfect_sampling):
        sampler = DT.PerfectBatchSampler(ds, hp.languages, hp.batch_size, shuffle=train, drop_last=train, rank=rank, world=world)
        collate = DT.Collate(False)
    else:
        sampler = DT.GlobalBatchSampler(ds, hp.batch_size, shuffle=train and not balanced, balanced=train and balanced,
                                        drop_last=train, rank=rank, world=world)
        collate = DT.Collate(True)
    # pinned batches: `batch_to_device` copies them with non_blocking=True, which only IS asynchronous from pinned memory (from pageable memory
    # every copy is hipMemcpyAsync + hipStreamSynchronize: the host would lose its run-ahead - and the GPU some milliseconds - once per step)
    return DataLoader(ds, batch_sampler=sampler, collate_fn=collate, num_workers=workers, pin_memory=torch.cuda.is_available()), sampler


EVAL_TERMS = ('mel_pre', 'mel_pos', 'stop_token', 'guided_att', 'lang_class')      # the loss dictionary of TacotronLoss.forward
EVAL_EXTRAS = ('free_running_mse', 'classifier_accuracy')                              # reported beside the loss, never summed into it


def _free_running_mse(hp, post0, stop0, target, target_length):
    """Mean squared error of the FREE-RUNNING prediction against the target, per utterance over the frames both have: the generated
    length follows the reference's rule (train.py:137-139: first frame with sigmoid(stop) > 0.5, plus hp.stop_frames, capped).  A
    stand-in for the reference's mel-cepstral distortion (train.py:134-145), whose MFCC / DTW need librosa + fastdtw (absent here)."""
    probs = torch.sigmoid(stop0)
    B, T = probs.shape
    hit = probs > 0.5
    first = torch.where(hit.any(1), hit.float().argmax(1), torch.full((B,), T, device=probs.device))
    n_gen = torch.clamp(first + getattr(hp, 'stop_frames', 5), max=T)
    n = torch.minimum(n_gen, target_length.to(probs.device)).clamp_min(1)
    m = (torch.arange(T, device=probs.device)[None, :] < n[:, None]).to(post0.dtype)            # [B, T]
    d2 = ((post0 - target) ** 2).mean(1) * m                                                       # mean over mel channels
    return float((d2.sum(1) / n).mean())


def _classifier_accuracy(hp, text_length, speakers, spk_pred):
    """Share of valid characters whose adversarial-classifier prediction is the utterance's speaker (train.py:147-156)."""
    mask = (torch.arange(spk_pred.shape[1], device=spk_pred.device)[None, :] < text_length.to(spk_pred.device)[:, None])
    match = (spk_pred.argmax(-1) == speakers.to(spk_pred.device)[:, None]) & mask
    return float(match.sum()) / max(float(mask.sum()), 1.0)


def evaluate(hp, data, model, crit, device, rank=0, world=1, free_running=True):
    """Validation (train.py:100-160): per batch the model runs TWICE like the reference's (`:124-125`) - teacher forced (the loss terms)
    and free running (teacher forcing 0.0: the general decoder schedule on the model's own frames) - and the result is the mean of
    the per-batch loss terms plus, beside them, the free-running reconstruction error and the adversarial classifier's accuracy
    (EVAL_EXTRAS; the reference's MCD and alignment plots need its audio stack, absent here).
    Data parallel: the batches are the reference's full (unsharded) validation batches; rank r takes batches r, r + world, ...
    and the per-term sums and the batch count are all-reduced, so every rank works (no rank idles in a barrier while rank 0
    evaluates - a long validation pass would otherwise run into the collective watchdog) and all ranks return the same means.
    No collective runs inside the loop (the ranks see different numbers of batches)."""
    from multilingual_text_to_speech_amd import data as DT
    model.eval()
    sums, n = {}, 0
    with torch.no_grad():
        for i, collated in enumerate(data):
            if i % world != rank:
                continue
            b = DT.batch_to_device(collated, device)
            post, pre, stop, align, spk, enc = model(b['text'], b['text_length'], b['target'], b['target_length'], b['speakers'],
                                                     b['languages'], 1.0)
            _, parts = crit(b['text_length'].to(device), b['target_length'].to(device), pre, b['target'], post, b['target'], stop,
                            b['stop'], align, b['speakers'], spk, enc, None)
            parts = dict(parts)
            if free_running:
                post0, _, stop0, _, _, _ = model(b['text'], b['text_length'], b['target'], b['target_length'], b['speakers'],
                                                 b['languages'], 0.0)
                parts['free_running_mse'] = _free_running_mse(hp, post0, stop0, b['target'], b['target_length'])
            if spk is not None and b['speakers'] is not None:
                parts['classifier_accuracy'] = _classifier_accuracy(hp, b['text_length'], b['speakers'], spk)
            for k, v in parts.items():
                sums[k] = sums.get(k, 0.0) + float(v)
            n += 1
    model.train()
    if world > 1:
        keys = sorted(EVAL_TERMS + EVAL_EXTRAS)
        t = torch.tensor([sums.get(k, 0.0) for k in keys] + [float(n)], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t)
        n = int(t[-1].item())
        sums = {k: float(v) for k, v in zip(keys, t[:-1]) if k in sums or v != 0.0}
    return {k: v / max(n, 1) for k, v in sums.items()}


def train_on_dataset(args, hp, datasets, model, opt, crit, buckets, rank, world, device, epoch0, ckpt_dir, state):
    """Epoch loop of the reference (train.py:29-97,289-310) over cached spectrograms, one process per GPU."""
    from multilingual_text_to_speech_amd import data as DT
    from multilingual_text_to_speech_amd.kernels import check_device_errors
    train_set, val_set = datasets
    train_data, train_sampler = make_loader(hp, train_set, True, rank, world, args.loader_workers)
    eval_data = make_loader(hp, val_set, False, 0, 1, args.loader_workers)[0] if val_set is not None and len(val_set) else None
    # StepLR counted in epochs like the reference (train.py:271,296-297)
    step_size = max(1, hp.learning_rate_decay_each // max(1, len(train_data)))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size, hp.learning_rate_decay)
    if state is not None and state.get('scheduler'):
        sched.load_state_dict(state['scheduler'])
    for epoch in range(epoch0, args.epochs if args.epochs is not None else hp.epochs):
        train_sampler.set_epoch(epoch)
        model.train()
        t0, frames, done = time.time(), 0, 0
        for collated in train_data:
            batch = DT.batch_to_device(collated, device)
            global_step = done + epoch * len(train_data)
            loss = train_step(model, crit, opt, buckets, batch, hp, teacher_forcing_ratio(hp, global_step))
            frames += int(batch['target_length'].sum())
            done += 1
            if done == 2:
                settle_host_heap()       # once per epoch, two steps in (the evaluation / checkpoint of the previous epoch left new long-lived objects)
        torch.cuda.synchronize()
        check_device_errors(device)
        if hp.learning_rate_decay_start - hp.learning_rate_decay_each < epoch * len(train_data):
            sched.step()
        # validation over the full (unsharded) batches of the reference's single-process evaluate(), dealt round-robin to the ranks
        eval_losses = evaluate(hp, eval_data, model, crit, device, rank, world) if eval_data is not None else {}
        if rank == 0:
            eval_loss = sum(v for k, v in eval_losses.items() if k in EVAL_TERMS) if eval_losses else float(loss.item())
            extras = '  '.join(f'{k} {eval_losses[k]:.4f}' for k in EVAL_EXTRAS if k in eval_losses)
            print(f'epoch {epoch}: train loss {loss.item():.4f}  eval loss {eval_loss:.4f}  {extras}  '
                  f'{frames * world / (time.time() - t0):.0f} frames/s', flush=True)
            if (epoch + 1) % hp.checkpoint_each_epochs == 0:
                DT.save_checkpoint(os.path.join(ckpt_dir, f'{hp.version}_loss-{epoch}-{eval_loss:2.3f}'), epoch, model, opt, sched, crit)


if __name__ == '__main__':
    main()
