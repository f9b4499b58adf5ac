#!/usr/bin/env python
"""Training-throughput benchmark of the MI355X-native text->mel hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): params/shared_training (CSS10, 10 languages, simple encoder + language
embedding, Dm = 544), per-GPU batch 64, 120 characters -> 600 mel frames, fp32, synthetic seeded inputs,
random-init weights.  One step = forward + TacotronLoss + backward + gradient all-reduce (N > 1) +
clip_grad_norm_(0.25) + Adam step.  Prints ONE JSON line (rank 0).

Extra objects on the line:
  roofline     - the dominant kernel (skinny_kernel<4>: the attention-LSTM recurrent step, 1 launch per frame):
                 algorithmic FLOP per launch / its average duration sampled live with HIP events on its stream.
  cpu_baseline - the CPU oracle (oracle/tacotron_oracle.py, a torch-CPU port of the reference's arithmetic) timed on
                 this host's cores on a bounded sample of the same workload (smaller batch / fewer frames).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L_CHARS, T_FRAMES, PER_GPU_BATCH = 120, 600, 64
PRESET = 'shared_training'


def synthetic_batch(hp, B, L, T, device, seed=1):
    """SURVEY.md section 8(d) synthetic inputs."""
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(3, hp.symbols_count() + 3, (B, L), generator=g)
    target = torch.randn(B, hp.num_mels, T, generator=g)
    stop = torch.zeros(B, T)
    stop[:, T - hp.stop_frames:] = 1.0
    batch = dict(text=text, text_length=torch.full((B,), L, dtype=torch.int64), target=target,
                 target_length=torch.full((B,), T, dtype=torch.int64), stop=stop,
                 speakers=torch.randint(0, max(hp.speaker_number, 1), (B,), generator=g) if hp.multi_speaker else None,
                 languages=(torch.arange(B) % hp.language_number) if hp.multi_language else None)
    return {k: (v.to(device) if torch.is_tensor(v) and k not in ('text_length', 'target_length') else v) for k, v in batch.items()}


def train_step(model, crit, opt, buckets, batch, hp, teacher_forcing=1.0):
    if buckets is not None and buckets.overlap:
        buckets.zero_grad()              # gradients are views into the all-reduce buckets
    else:
        opt.zero_grad(set_to_none=True)
    post, pre, stop, align, spk, enc = model(batch['text'], batch['text_length'], batch['target'], batch['target_length'],
                                             batch['speakers'], batch['languages'], teacher_forcing)
    dev = post.device
    loss, _ = crit(batch['text_length'].to(dev), batch['target_length'].to(dev), pre, batch['target'], post, batch['target'],
                   stop, batch['stop'], align, batch['speakers'], spk, enc, None)
    loss.backward()
    if buckets is not None:
        buckets.all_reduce()
    if hasattr(opt, '_tables'):
        opt.step(max_norm=hp.gradient_clipping)          # fused clip_grad_norm_ + Adam (mtts_clip_adam_step)
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), hp.gradient_clipping)
        opt.step()
    crit.update_states()
    return loss


def pmc_traffic(preset, B):
    """HBM bytes per launch of the roofline kernel from the committed PMC measurement (rocprofv3 --pmc cannot run inside the
    timed process); None when this workload was not measured."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            return json.load(f).get(f'{preset}/B{B}', {}).get('traffic_bytes')
    except OSError:
        return None


def cpu_baseline(seconds_budget=25.0):
    """Time the CPU oracle (port of the reference) on a bounded sample: same config, batch 8, 40 frames."""
    from oracle import tacotron_oracle as O
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    presets.apply(PRESET)
    cores = min(os.cpu_count() or 1, 16)      # small-op oracle: more threads only add synchronisation cost
    torch.set_num_threads(cores)
    torch.set_flush_denormal(True)
    B, L, T = 8, L_CHARS, 40
    torch.manual_seed(0)
    model = Tacotron()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith(('running_mean', 'running_var')))
          for k, v in model.state_dict().items()}
    cfg = O.cfg_from_params(hp)
    b = synthetic_batch(hp, B, L, T, 'cpu')
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=hp.learning_rate, weight_decay=hp.weight_decay)
    g = torch.Generator().manual_seed(3)
    keep = lambda shape, p: (torch.rand(shape, generator=g) >= p).float() / (1 - p)
    H, P, E, PD = hp.decoder_dimension, hp.prenet_dimension, hp.encoder_dimension, hp.postnet_dimension
    masks = {f'enc.{i}': keep((B, E, L), hp.dropout) for i in range(hp.encoder_blocks)}
    masks.update({f'prenet.{i}': keep((B, T + 1, P), hp.dropout) for i in range(hp.prenet_layers)})
    masks.update(att_lstm=keep((T, B, H), hp.dropout_hidden), gen_lstm=keep((T, B, H), hp.dropout_hidden))
    masks.update({f'post.{i}': keep((B, PD if i < hp.postnet_blocks - 1 else hp.num_mels, T), hp.dropout)
                  for i in range(hp.postnet_blocks)})
    teacher = torch.ones(T, dtype=torch.bool)
    times = []
    t_start = time.time()
    for it in range(4):
        t0 = time.time()
        opt.zero_grad()
        out = O.tacotron_forward(sd, cfg, b['text'], b['text_length'], b['target'], b['target_length'], b['speakers'],
                                 b['languages'], teacher, masks, True)
        loss, _ = O.tacotron_loss(cfg, out, b['text_length'], b['target_length'], b['target'], b['stop'], b['speakers'], 0.25)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, hp.gradient_clipping)
        opt.step()
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget:
            break
    steady = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    return dict(value=round(B * T / steady, 1), unit='mel-frames/s', cores=cores, kind='port',
                sample=f'oracle/tacotron_oracle.py train step (fwd+loss+bwd+clip+Adam), {PRESET}, batch {B}, '
                       f'{L} chars -> {T} frames, {len(times)} steps, median of steady steps, flush-denormal on')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=PER_GPU_BATCH, help='per-GPU batch')
    ap.add_argument('--frames', type=int, default=T_FRAMES)
    ap.add_argument('--preset', default=PRESET)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return

    from multilingual_text_to_speech_amd import _C, dist as D
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
    rank, world, local = D.init()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the hot path has no CPU fallback)')
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    device = torch.device('cuda', local)
    presets.apply(args.preset, speaker_number=91)
    G = hp.language_number if hp.encoder_type in ('generated', 'convolutional') else 1
    B, L, T = args.batch, L_CHARS, args.frames
    D.shard_bounds(B * world, rank, world, G)          # validates divisibility (weak scaling: B per GPU fixed)
    torch.manual_seed(0)
    model = Tacotron().to(device).train()
    D.broadcast_parameters(model)
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    from multilingual_text_to_speech_amd.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
    buckets = D.GradientBuckets(model.parameters(), overlap=os.environ.get('MTTS_DDP_OVERLAP', '1') != '0') if world > 1 else None
    batch = synthetic_batch(hp, B, L, T, device, seed=1 + rank)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(model, crit, opt, buckets, batch, hp)
    lib = _C.lib()
    n_samples = 16
    _C.check(lib.mtts_prof_begin(n_samples * args.steps, max(1, T // n_samples)), 'prof_begin')
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(model, crit, opt, buckets, batch, hp)
    barrier()
    dt = time.perf_counter() - t0
    tot_ms, cnt = ctypes.c_float(0), ctypes.c_int(0)
    _C.check(lib.mtts_prof_end(ctypes.byref(tot_ms), ctypes.byref(cnt)), 'prof_end')
    t_max = torch.tensor([dt], device=device)
    if world > 1:
        torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX)
    dt = float(t_max.item())

    if rank == 0:
        frames = B * T * world * args.steps
        H, P = hp.decoder_dimension, hp.prenet_dimension
        Dm = hp.encoder_dimension + (hp.speaker_embedding_dimension if hp.multi_speaker else 0) + \
            (hp.language_embedding_dimension if hp.multi_language else 0)
        # dominant kernel: attention-LSTM recurrent step, [B, Dm+H] x [4H, Dm+H]^T + fused cell (prenet part hoisted)
        k_rec = Dm + H
        flop = 2.0 * B * k_rec * 4 * H
        bytes_alg = 4.0 * (4 * H * k_rec + B * k_rec + B * 4 * H + 2 * 4 * H + 3 * B * H + B * 4 * H + B * H)
        # every sampled launch sits in a HIP-event bracket preceded by an EMPTY bracket on the same stream; the empty one measures
        # what an event pair costs by itself and is subtracted (rocprofv3 reports the bare kernel; see profiles/)
        lib.mtts_prof_empty_ms.restype = ctypes.c_float
        raw_s = (tot_ms.value / max(cnt.value, 1)) * 1e-3
        empty_s = (float(lib.mtts_prof_empty_ms()) / max(cnt.value, 1)) * 1e-3
        avg_s = max(raw_s - empty_s, 1e-9)
        achieved = flop / avg_s / 1e12 if avg_s > 0 else 0.0
        line = {
            'metric': 'mel-frames/sec (train, fwd+bwd+optimizer, batch 64/GPU, 120 chars -> 600 frames)',
            'value': round(frames / dt, 1), 'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'params/{args.preset}' + (' (BASELINE configs[1])' if args.preset == PRESET else '') + f' train step, per-GPU batch {B}, L={L} -> T={T}, '
                                   f'fp32, random-init weights', 'gemm_core': 'fp32 in/out; products as 6 bf16x bf16 MFMA terms of exact 3-way operand splits, fp32 accumulate (error <= fp32 chain)',
                       'global_batch': B * world, 'parallelism': f'dp{world}',
                       'loss': float(loss.item())},
            'roofline': {'bound': 'mfma', 'kernel': 'skinny_kernel<4> (attention-LSTM step: [B,Dm+H]x[4H,Dm+H]^T + LSTM cell)',
                         'achieved': round(achieved, 2), 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': round(achieved / 157.3, 4),
                         'flop_per_launch': flop, 'bytes_per_launch': bytes_alg, 'avg_launch_us': round(avg_s * 1e6, 2),
                         'event_bracket_us': round(raw_s * 1e6, 2), 'empty_bracket_us': round(empty_s * 1e6, 2),
                         'hbm_frac_of_8TBps': round(bytes_alg / avg_s / 8e12, 4) if avg_s > 0 else 0.0, 'samples': cnt.value,
                         'traffic': pmc_traffic(args.preset, B)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:      # separate process + hard timeout: the baseline is reporting only, never lose the GPU number over it
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only'], capture_output=True,
                                   text=True, timeout=180, env={**os.environ, 'HIP_VISIBLE_DEVICES': ''})
                line['cpu_baseline'] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as exc:
                line['cpu_baseline'] = {'error': repr(exc)[:200]}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
