#!/usr/bin/env python
"""Training-throughput benchmark of the MI355X-native text->mel hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): params/shared_training (CSS10, 10 languages, simple encoder + language
embedding, Dm = 544), per-GPU batch 64, 120 characters -> 600 mel frames, fp32, synthetic seeded inputs,
random-init weights.  One step = forward + TacotronLoss + backward + gradient all-reduce (N > 1) +
clip_grad_norm_(0.25) + Adam step.  Prints ONE JSON line (rank 0).

Extra objects on the line (N = 1):
  roofline     - SURVEY 8(d)'s quantity for the workload's batch: the attention+decoder FORWARD step (everything one output
                 frame costs in the teacher-forced decoder: prenet, attention LSTM, query, attention, generator LSTM,
                 frame/stop projection).  achieved = algorithmic bytes per step 4*(W + B*act) / measured time per step
                 (HIP events on the launch stream around the whole 600-step decoder forward); peak 8 TB/s.
                 `traffic` = HBM bytes per step from live rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate passes,
                 difference of a long and a short decode so that everything not proportional to the frame count cancels).
                 `kernels` keeps the per-launch figures of the dominant kernel as a sub-field, sampled inside the timed train steps:
                 `persistent_attention_decoder` (pdec_kernel: ONE launch = all 600 steps of attention LSTM + attention, reported
                 per step) or, when the per-step launch schedule runs (batch > 64), `attention_lstm_step`.
  train_b40_bf16 - the whole train step of ONE RANK of BASELINE configs[3] (generated_switching, batch 40, classifier on), bf16 and fp32
  roofline_b240 / roofline_b240_bf16 / roofline_b40_bf16 - the same quantity for params/generated_switching at batch 240 (the valid
                 batch next to the north star's 256) in fp32 and bf16, and at batch 40 (one rank's shard of configs[3]) in bf16.
  inference    - BASELINE configs[4]: batched synthesis, 128 utterances x 201 tokens -> 600 frames, with its own step roofline.
  cpu_baseline - the CPU oracle (oracle/tacotron_oracle.py, a torch-CPU port of the reference's arithmetic, kind "port") timed
                 on this host's cores on the SAME configuration (batch 64 x 600 frames; 2 train steps, <= 150 s), plus
                 `reference_recorded`: the reference itself (kind "reference") as recorded in the build container by
                 scripts/cpu_reference_baseline.py, and the ratios gpu_over_port / gpu_over_reference_recorded.

`python bench.py --gpus N` (N > 1) outside a torch.distributed launcher re-executes itself under torch.distributed.run
(127.0.0.1, free port); under the driver's own launcher it is a plain worker.
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')      # kernel arguments in device memory: -2 % per train step (read when the HIP runtime loads, i.e. before torch)
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L_CHARS, T_FRAMES, PER_GPU_BATCH = 120, 600, 64
GEMM_CORE = {'f32': 'fp32 in/out; products as 6 bf16x bf16 MFMA terms of exact 3-way operand splits, fp32 accumulate (error <= fp32 chain)',
             'bf16': 'operands rounded to bf16 (RNE) at tile staging, recurrent weights stored bf16; one MFMA term, fp32 accumulate, fp32 state/outputs'}
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
    # the sequence lengths stay host tensors: the loss moves them to the device without synchronising the stream (kernels.to_device_async)
    loss, _ = crit(batch['text_length'], batch['target_length'], pre, batch['target'], post, batch['target'],
                   stop, batch['stop'], align, batch['speakers'], spk, enc, None)
    loss.backward()
    if buckets is not None:
        buckets.all_reduce()
    if hasattr(opt, '_tables'):
        opt.step(max_norm=hp.gradient_clipping)          # fused clip_grad_norm_ + Adam (mtts_clip_adam_step)
        opt.poll_skipped()                               # non-blocking: counts / warns about steps the device-side guard skipped
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), hp.gradient_clipping)
        opt.step()
    crit.update_states()
    from multilingual_text_to_speech_amd.kernels import poll_device_errors
    poll_device_errors(dev)          # non-blocking: raises for what the previous step's poll fetched (the guarded Adam step protects the weights)
    return loss


def model_dims(hp):
    Dm = hp.encoder_dimension + (hp.speaker_embedding_dimension if hp.multi_speaker else 0) + \
        (hp.language_embedding_dimension if hp.multi_language else 0)
    return dict(H=hp.decoder_dimension, P=hp.prenet_dimension, A=hp.attention_dimension, M=hp.num_mels,
                C=hp.attention_location_dimension, ks=hp.attention_kernel_size, Dm=Dm)


def step_algorithmic(hp, B, L, elem_bytes=4):
    """algorithmic bytes and FLOPs of ONE forward decoder step of B samples (weights read once per step)."""
    d = model_dims(hp)
    H, P, A, M, C, ks, Dm = d['H'], d['P'], d['A'], d['M'], d['C'], d['ks'], d['Dm']
    W = 4 * H * (P + Dm + H) + 4 * H * (H + Dm + H) + 16 * H + A * H + C * ks + A * C + 2 * A + (M + 1) * (H + Dm + 1)
    act = L * A + L * Dm + 3 * L + 8 * H + P + 2 * Dm + M + 1
    flop = 2.0 * B * (4 * H * (P + Dm + H) + 4 * H * (H + Dm + H) + A * H + L * (C * ks + A * C + A + Dm) + (M + 1) * (H + Dm))
    return dict(weights=W, act_per_sample=act, bytes=float(elem_bytes) * (W + B * act), flop=flop, Dm=Dm)


def decoder_forward_us(model, hp, batch, L, repeats=3):
    """Median time (us) of the whole teacher-forced decoder forward, bracketed by HIP events on the launch stream (the helper
    streams join the caller's stream before mtts_decoder_fwd returns, so the bracket covers them)."""
    import multilingual_text_to_speech_amd.kernels as K
    with torch.no_grad():
        langs = batch['languages']
        lang = langs.unsqueeze(1).expand(-1, L) if langs is not None else None
        spk = batch['speakers'].unsqueeze(1).expand(-1, L) if batch['speakers'] is not None else None
        emb = K.embedding(model._embedding.weight, batch['text'], 0)
        enc = model._encoder(emb, batch['text_length'], lang)
        times = []
        for _ in range(repeats + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model._decoder(enc, batch['text_length'], batch['target'], 1.0, spk, lang)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
    steady = sorted(times[1:])
    return steady[len(steady) // 2], enc, spk, lang


def decoder_backward_us(model, hp, batch, L, repeats=2):
    """Median time (us) of the whole decoder BACKWARD (mtts_decoder_bwd: both recurrence chains, the per-chunk weight-gradient and
    input-gradient GEMMs, the batched tail), bracketed by HIP events on the launch stream (the helper streams join before the call
    returns).  The forward runs un-timed in front of it; random upstream gradients for mels, stop logits and alignments."""
    import multilingual_text_to_speech_amd.kernels as K
    langs = batch['languages']
    lang = langs.unsqueeze(1).expand(-1, L) if langs is not None else None
    spk = batch['speakers'].unsqueeze(1).expand(-1, L) if batch['speakers'] is not None else None
    with torch.no_grad():
        emb = K.embedding(model._embedding.weight, batch['text'], 0)
        enc = model._encoder(emb, batch['text_length'], lang)
    dec_params = [p for p in model._decoder.parameters() if p.requires_grad]
    times = []
    for _ in range(repeats + 1):
        e_in = enc.detach().clone().requires_grad_(True)
        outs = model._decoder(e_in, batch['text_length'], batch['target'], 1.0, spk, lang)
        gs = [torch.randn_like(o) * 1e-3 for o in outs]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.autograd.grad(list(outs), [e_in] + dec_params, gs, allow_unused=True)
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
        del outs, gs, e_in
    steady = sorted(times[1:])
    return steady[len(steady) // 2]


def backward_roofline(model, hp, batch, B, L, T, preset, dtype='f32'):
    """'backward re-reads the same tensors plus saved per-step state: bwd algorithmic bytes = 2 x fwd + saved-state
    traffic, reported separately'.  Saved state per sample and step (written by the forward, read by the backward, fp32): h / c of
    both cells (4H), activated gates of both cells (8H), context (Dm), cumulative alignment + alignment (2L), query (A), prenet
    activations of both layers (2P), frame + stop (M + 1)."""
    us = decoder_backward_us(model, hp, batch, L)
    us_step = us / T
    alg = step_algorithmic(hp, B, L)
    d = model_dims(hp)
    saved = 12 * d['H'] + d['Dm'] + 2 * L + d['A'] + 2 * d['P'] + d['M'] + 1
    fwd_bytes = alg['bytes'] if dtype != 'bf16' else 2.0 * alg['weights'] + 4.0 * B * alg['act_per_sample']
    bytes_step = 2.0 * fwd_bytes + 2.0 * 4.0 * B * saved          # saved state: one write (forward) + one read (backward)
    gbps = bytes_step / (us_step * 1e-6) / 1e9
    return {'bound': 'hbm', 'what': f'decoder BACKWARD per step (teacher forced), params/{preset}, batch {B}, L={L}: attention backward, both LSTM '
                                    'cell backwards, per-step input-gradient products, weight-gradient GEMMs, batched tail',
            'achieved': round(gbps, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbps / 8000.0, 4), 'us_per_step': round(us_step, 2),
            'ms_per_backward': round(us * 1e-3, 2), 'bytes_per_step': bytes_step,
            'bytes_formula': '2 x forward algorithmic bytes + (write + read) of the saved per-step state', 'saved_state_elems_per_sample': saved,
            'flop_per_step': 2.0 * alg['flop'], 'fp32_mfma_frac_of_157TF': round(2.0 * alg['flop'] / (us_step * 1e-6) / 157.3e12, 4),
            'frames_timed': T, 'dtype': dtype}


def step_roofline(model, hp, batch, B, L, T, preset, dtype='f32'):
    """The north star's quantity: attention+decoder forward step against the HBM roofline.  In bf16 mode the weights are
    counted at 2 bytes (the step kernels stream bf16-packed copies), activations at 4 (they stay fp32 in HBM)."""
    us, _, _, _ = decoder_forward_us(model, hp, batch, L)
    us_step = us / T
    alg = step_algorithmic(hp, B, L)
    if dtype == 'bf16':
        alg['bytes'] = 2.0 * alg['weights'] + 4.0 * B * alg['act_per_sample']
    gbps = alg['bytes'] / (us_step * 1e-6) / 1e9
    return {'bound': 'hbm', 'what': f'attention+decoder forward step (teacher forced), params/{preset}, batch {B}, L={L}, Dm={alg["Dm"]}: '
                                    'prenet + attention LSTM + query + location-sensitive attention + generator LSTM + frame/stop projection',
            'achieved': round(gbps, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbps / 8000.0, 4),
            'us_per_step': round(us_step, 2), 'bytes_per_step': alg['bytes'], 'flop_per_step': alg['flop'],
            'frac_of_measured_copy_6290GBps': round(gbps / 6290.0, 4),
            'fp32_mfma_frac_of_157TF': round(alg['flop'] / (us_step * 1e-6) / 157.3e12, 4),
            'frames_timed': T, 'dtype': dtype, 'traffic': None}


def secondary_step_roofline(preset, B, L, T, device, dtype='f32'):
    """Same quantity for another preset / batch (fresh random-init model; BASELINE north star: batch 256 -> nearest valid 240)."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    presets.apply(preset, speaker_number=91)
    torch.manual_seed(0)
    model = Tacotron().to(device).train()
    batch = synthetic_batch(hp, B, L, T, device)
    from multilingual_text_to_speech_amd import _C
    before = _C.get_precision()
    _C.set_precision(dtype)
    try:
        out = step_roofline(model, hp, batch, B, L, T, preset, dtype)
    finally:
        _C.set_precision('bf16' if before else 'fp32')
    del model
    return out


def secondary_train_step(preset, B, L, T, device, dtype='bf16', train_steps=3, warm_steps=2):
    """`train_b40_bf16` (round 6): the PER-RANK work of BASELINE configs[3] on one GPU - params/generated_switching (5 language groups,
    91 speakers, adversarial speaker classifier on), per-rank batch 40 (global 320 over 8 GPUs), 120 characters -> T frames, full train
    step (forward + TacotronLoss incl. the classifier's cross entropy + backward + fused clip / Adam) in `dtype` arithmetic, and the
    same step in fp32 on the same box for the ratio.  No all-reduce: one rank (the 8-GPU run is the driver's)."""
    from multilingual_text_to_speech_amd import _C
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
    from multilingual_text_to_speech_amd.optim import FusedAdam
    presets.apply(preset, speaker_number=91)
    before = _C.get_precision()
    out = {'workload': f'params/{preset} train step (classifier {"on" if hp.reversal_classifier else "off"}), per-rank batch {B}, L={L} -> T={T}, '
                       'one rank of BASELINE configs[3]', 'frames_per_step': B * T}
    try:
        for dt_ in ('f32', dtype):
            _C.set_precision(dt_)
            torch.manual_seed(0)
            model = Tacotron().to(device).train()
            batch = synthetic_batch(hp, B, L, T, device)
            crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
            opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)

            def one():
                opt.zero_grad(set_to_none=True)
                post, pre, stop, align, spk, enc = model(batch['text'], batch['text_length'], batch['target'], batch['target_length'],
                                                         batch['speakers'], batch['languages'], 1.0)
                loss, _ = crit(batch['text_length'].to(device), batch['target_length'].to(device), pre, batch['target'], post, batch['target'],
                               stop, batch['stop'], align, batch['speakers'], spk, enc, None)
                loss.backward()
                opt.step(max_norm=hp.gradient_clipping)
                crit.update_states()
                return loss
            for _ in range(warm_steps):
                one()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(train_steps):
                loss = one()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / train_steps
            out[f'{dt_}_ms_per_step'] = round(ms, 2)
            out[f'{dt_}_frames_per_s'] = round(B * T / (ms * 1e-3), 1)
            out[f'{dt_}_loss'] = float(loss.item())
            del model, opt, crit, batch
            torch.cuda.empty_cache()
        out['ratio'] = round(out[f'{dtype}_ms_per_step'] / out['f32_ms_per_step'], 3)
        out['ms_per_step'] = out[f'{dtype}_ms_per_step']
        out['dtype'] = dtype
    finally:
        _C.set_precision('bf16' if before else 'fp32')
    return out


def long_input_roofline(device, preset=PRESET, B=PER_GPU_BATCH, L=200, T=300, train_steps=3, warm_steps=2):
    """`roofline_L200` (round 5): the headline preset on inputs the length real CSS10 batches have - 200 characters at most, ragged
    lengths U[100, 200] sorted like the collate function sorts them (the reference's validation set has min 25 / median 124
    / max 304 characters, so a batch of 60 almost always exceeds the 128 positions the round-3/4 persistent decoder was limited to).
    Decoder forward step against the HBM roofline with the algorithmic bytes of the MEAN valid length (padding is not algorithmic
    work), and the whole train step on the same batch."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
    from multilingual_text_to_speech_amd.optim import FusedAdam
    presets.apply(preset, speaker_number=91)
    torch.manual_seed(0)
    model = Tacotron().to(device).train()
    batch = synthetic_batch(hp, B, L, T, device)
    g = torch.Generator().manual_seed(7)
    tl = torch.sort(torch.randint(L // 2, L + 1, (B,), generator=g), descending=True).values
    tl[0] = L
    batch['text_length'] = tl
    for b in range(B):
        batch['text'][b, int(tl[b]):] = 0
    out = step_roofline(model, hp, batch, B, L, T, preset, 'f32')
    mean_len = float(tl.float().mean())
    alg = step_algorithmic(hp, B, mean_len)
    gbps = alg['bytes'] / (out['us_per_step'] * 1e-6) / 1e9
    out.update(achieved=round(gbps, 1), frac=round(gbps / 8000.0, 4), bytes_per_step=alg['bytes'], bytes_per_step_padded=out['bytes_per_step'],
               mean_valid_length=round(mean_len, 1), lengths='U[100, 200] sorted descending, first = 200',
               what=out['what'] + f'; ragged lengths (mean {mean_len:.0f} of {L}), bytes counted at the mean valid length')
    out.pop('frac_of_measured_copy_6290GBps', None)
    # the whole train step on this batch (what a real CSS10 batch costs against the L = 120 headline)
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
    for _ in range(warm_steps):
        train_step(model, crit, opt, None, batch, hp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(train_steps):
        train_step(model, crit, opt, None, batch, hp)
    torch.cuda.synchronize()
    out['train_ms_per_step'] = round(1e3 * (time.perf_counter() - t0) / train_steps, 2)
    out['train_frames_per_s'] = round(B * T / (out['train_ms_per_step'] * 1e-3), 1)
    del model, opt
    return out


def inference_bench(device, preset='generated_switching', utterances=128, chars=200, frames=600, repeats=2):
    """BASELINE configs[4]: batched autoregressive synthesis (encoder + free-running decoder + post-net), frame count pinned
    (stop rule disabled), plus the free-running decoder step against its HBM roofline (L = 201)."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    presets.apply(preset, speaker_number=91)
    hp.max_output_length = frames
    torch.manual_seed(0)
    model = Tacotron().to(device).eval()
    g = torch.Generator().manual_seed(1)
    L = chars + 1
    texts = [torch.cat([torch.randint(3, hp.symbols_count() + 3, (chars,), generator=g), torch.tensor([1])]) for _ in range(utterances)]
    n_lang = len(hp.languages)
    langs = None
    if hp.multi_language:
        langs = []
        for i in range(utterances):
            w = torch.zeros(L, n_lang); w[:, i % n_lang] = 1.0
            langs.append(w)
    spks = [i % hp.speaker_number for i in range(utterances)] if hp.multi_speaker else None
    def timed(n):
        times = []
        for i in range(n + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = model.inference_batch(texts, spks, langs, stop_threshold=2.0)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
            if i == 0:
                from multilingual_text_to_speech_amd.utils import settle_host_heap
                settle_host_heap()      # the collector's full pass over the new model's objects happens here, not inside a timed decode
        assert all(o.shape == (hp.num_mels, frames) for o in out), out[0].shape
        return sorted(times[1:])[len(times[1:]) // 2]
    dt = timed(repeats)
    # the same with every 32-step chunk of the decode loop replayed as a hipGraph (mtts_decoder_fwd_graphed; first call eager,
    # second captures, later ones replay)
    graph = None
    try:
        from multilingual_text_to_speech_amd import decoder_ops as D
        os.environ['MTTS_DECODE_GRAPH'] = '1'
        dt_g = timed(repeats + 2)
        sess = next(iter(D.GraphedDecode._cache.values()))
        graph = {'value': round(utterances * frames / dt_g, 1), 'seconds_per_batch': round(dt_g, 4), 'chunks_replayed_per_call': sess.replayed}
    except Exception as exc:
        graph = {'error': repr(exc)[:200]}
    finally:
        os.environ.pop('MTTS_DECODE_GRAPH', None)
    alg = step_algorithmic(hp, utterances, L)
    us_step = dt / frames * 1e6
    gbps = alg['bytes'] / (us_step * 1e-6) / 1e9
    return {'metric': 'mel-frames/sec (batched synthesis: encoder + free-running decoder + post-net)',
            'value': round(utterances * frames / dt, 1), 'unit': 'mel-frames/s', 'seconds_per_batch': round(dt, 4),
            'workload': f'params/{preset} synthesis (BASELINE configs[4]), {utterances} utterances x {L} tokens -> {frames} frames, '
                        'stop rule disabled, random-init weights, fp32',
            'hipgraph_replay': graph,
            'roofline': {'bound': 'hbm', 'what': 'free-running decoder step (whole-call time / frames: includes encoder and post-net)',
                         'achieved': round(gbps, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbps / 8000.0, 4),
                         'us_per_step': round(us_step, 2), 'bytes_per_step': alg['bytes']}}


# ---- live HBM traffic (rocprofv3 PMC) -------------------------------------------------------------------------------------
TRAFFIC_T = (48, 240)


def traffic_probe(args):
    """Child process run under `rocprofv3 --kernel-trace --pmc X`: marker | decode(T1) | marker | decode(T2) | marker."""
    from multilingual_text_to_speech_amd import _C
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    import multilingual_text_to_speech_amd.kernels as K
    device = torch.device('cuda', 0)
    if args.dtype == 'bf16':
        _C.set_precision('bf16')
    presets.apply(args.preset, speaker_number=91)
    torch.manual_seed(0)
    model = Tacotron().to(device).train()
    B, L = args.batch, args.chars
    lib = _C.lib()
    with torch.no_grad():
        batches = {T: synthetic_batch(hp, B, L, T, device) for T in TRAFFIC_T}
        b0 = batches[TRAFFIC_T[0]]
        langs = b0['languages']
        lang = langs.unsqueeze(1).expand(-1, L) if langs is not None else None
        spk = b0['speakers'].unsqueeze(1).expand(-1, L) if b0['speakers'] is not None else None
        enc = model._encoder(K.embedding(model._embedding.weight, b0['text'], 0), b0['text_length'], lang)
        model._decoder(enc, b0['text_length'], b0['target'], 1.0, spk, lang)          # warm-up (allocator, stream creation)
        torch.cuda.synchronize()
        for k, T in enumerate(TRAFFIC_T):
            _C.check(lib.mtts_prof_marker(k + 1, _C.stream_ptr()), 'marker')
            model._decoder(enc, b0['text_length'], batches[T]['target'], 1.0, spk, lang)
            torch.cuda.synchronize()
        _C.check(lib.mtts_prof_marker(len(TRAFFIC_T) + 1, _C.stream_ptr()), 'marker')
        torch.cuda.synchronize()


def measure_traffic(preset, B, dtype='f32', timeout=240, chars=L_CHARS):
    """HBM bytes per forward decoder step from two rocprofv3 PMC passes over a child process (FETCH_SIZE, WRITE_SIZE; both
    in KB; gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2, MI355X_MICROARCH.md HBM section).
    Per step = (bytes of the long decode - bytes of the short one) / (difference of frame counts)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not found'
    totals = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='mtts_pmc_', dir='/tmp')
        try:
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '-d', d, '-o', 'p', '--output-format', 'csv', '--',
                   sys.executable, os.path.abspath(__file__), '--traffic-probe', '--preset', preset, '--batch', str(B), '--dtype', dtype, '--chars', str(chars)]
            r = subprocess.run(cmd, cwd='/tmp', env={**os.environ, 'TMPDIR': '/tmp'}, capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return None, f'rocprofv3 {counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}'
            rows = []
            with open(files[0], newline='') as f:
                for row in csv.DictReader(f):
                    if row['Counter_Name'] == counter:
                        rows.append((int(row['Dispatch_Id']), row['Kernel_Name'], float(row['Counter_Value'])))
            rows.sort()
            marks = [i for i, (_, name, _) in enumerate(rows) if 'mtts_marker_kernel' in name]
            if len(marks) != len(TRAFFIC_T) + 1:
                return None, f'{len(marks)} marker dispatches in the {counter} pass'
            totals[counter] = [sum(v for _, _, v in rows[marks[k] + 1:marks[k + 1]]) * 1024.0 for k in range(len(TRAFFIC_T))]
        except Exception as exc:       # reporting only
            return None, repr(exc)[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    dT = TRAFFIC_T[1] - TRAFFIC_T[0]
    fetch = 2.0 * (totals['FETCH_SIZE'][1] - totals['FETCH_SIZE'][0]) / dT
    write = (totals['WRITE_SIZE'][1] - totals['WRITE_SIZE'][0]) / dT
    return {'bytes_per_step': round(fetch + write), 'fetch_bytes_x2': round(fetch), 'write_bytes': round(write),
            'method': f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), decode of {TRAFFIC_T[1]} minus {TRAFFIC_T[0]} frames'}, None


def measure_mfma(preset, B, dtype='f32', timeout=240):
    """Matrix-pipe occupancy of the forward decoder's kernels from ONE rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES: matrix-pipe busy
    cycles summed over the chip's 1024 SIMDs) over the traffic-probe child.  Per kernel: busy cycles / (1024 SIMDs x duration x 2.4 GHz)
    = fraction of the matrix peak at the nominal clock (bf16 MFMA 2.5 PF dense / fp32 MFMA 157 TF: the counter counts pipe cycles
    whatever the operand type).  North star: 'rocprof HBM GB/s AND MFMA utilisation against chip peak'."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not found'
    d = tempfile.mkdtemp(prefix='mtts_pmc_', dir='/tmp')
    try:
        cmd = ['rocprofv3', '--kernel-trace', '--pmc', 'SQ_VALU_MFMA_BUSY_CYCLES', '-d', d, '-o', 'p', '--output-format', 'csv', '--',
               sys.executable, os.path.abspath(__file__), '--traffic-probe', '--preset', preset, '--batch', str(B), '--dtype', dtype]
        r = subprocess.run(cmd, cwd='/tmp', env={**os.environ, 'TMPDIR': '/tmp'}, capture_output=True, text=True, timeout=timeout)
        cfiles = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        tfiles = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
        if r.returncode != 0 or not cfiles or not tfiles:
            return None, f'rocprofv3 MFMA pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}'
        busy, dur = collections.defaultdict(float), collections.defaultdict(float)
        name_of = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        with open(cfiles[0], newline='') as f:
            for row in csv.DictReader(f):
                if row['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
                    busy[name_of(row['Kernel_Name'])] += float(row['Counter_Value'])
        with open(tfiles[0], newline='') as f:
            for row in csv.DictReader(f):
                dur[name_of(row['Kernel_Name'])] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
        top = sorted(dur.items(), key=lambda kv: -kv[1])[:6]
        total_ns = sum(dur.values())
        per_kernel = {k: {'share_of_kernel_time': round(v / total_ns, 3), 'mfma_frac_of_peak': round(busy.get(k, 0.0) / (1024.0 * v * 2.4), 4)}
                      for k, v in top if v > 0}
        overall = sum(busy.values()) / (1024.0 * total_ns * 2.4)
        return {'mfma_frac_of_peak': round(overall, 4), 'kernels': per_kernel,
                'method': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES over the decoder-forward probe: busy cycles / (1024 SIMDs x kernel time x 2.4 GHz)'}, None
    except Exception as exc:       # reporting only
        return None, repr(exc)[:200]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def recorded_reference_baseline():
    """The reference itself, timed in the build container (scripts/cpu_reference_baseline.py -> profiles/cpu_reference.json)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'cpu_reference.json')) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None
    keep = [dict(config=r['config'], batch=r['batch'], frames=r['frames'], flush_denormal=r['flush_denormal'],
                 value=r['mel_frames_per_s'], seconds_per_step=r['seconds_per_step']) for r in doc.get('results', [])]
    return dict(kind='reference', unit=doc.get('unit'), cores=doc.get('cores'), cpu=doc.get('cpu'), where='build container (the '
                'reference does not exist on the GPU box)', results=keep, source='profiles/cpu_reference.json')


def cpu_baseline(seconds_budget=150.0):
    """Time the CPU oracle (port of the reference) on the benchmark's OWN configuration (batch 64, 120 chars -> 600 frames): one
    warm step, then one timed step when the first one left enough of the budget (a step is ~25-60 s on 16 host cores); the
    sample says which step was reported."""
    from oracle import tacotron_oracle as O
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    presets.apply(PRESET)
    cores = min(os.cpu_count() or 1, 16)      # small-op oracle: more threads only add synchronisation cost
    torch.set_num_threads(cores)
    torch.set_flush_denormal(True)
    B, L, T = PER_GPU_BATCH, L_CHARS, T_FRAMES
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
    for it in range(2):
        t0 = time.time()
        opt.zero_grad()
        out = O.tacotron_forward(sd, cfg, b['text'], b['text_length'], b['target'], b['target_length'], b['speakers'],
                                 b['languages'], teacher, masks, True)
        loss, _ = O.tacotron_loss(cfg, out, b['text_length'], b['target_length'], b['target'], b['stop'], b['speakers'], 0.25)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, hp.gradient_clipping)
        opt.step()
        times.append(time.time() - t0)
        if time.time() - t_start + times[-1] > seconds_budget:      # a second step would not fit
            break
    steady = times[-1]
    return dict(value=round(B * T / steady, 1), unit='mel-frames/s', cores=cores, kind='port', seconds_per_step=round(steady, 2),
                sample=f'oracle/tacotron_oracle.py train step (fwd+loss+bwd+clip+Adam) on the benchmark configuration: {PRESET}, batch {B}, '
                       f'{L} chars -> {T} frames; step {len(times)} of {len(times)} timed '
                       f'({"after one warm step" if len(times) > 1 else "first step, no warm-up: the budget allowed one"}), flush-denormal on')


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec this script under torch.distributed.run (one rank per GPU,
    rendezvous on 127.0.0.1 with a free port) with the same arguments; the children print the JSON line themselves (rank 0)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this platform (RCCL needs it)
    return subprocess.run(cmd, env=env).returncode


def stub_main(args):
    """MTTS_BENCH_STUB=1 (tests/test_bench_launch.py): the launch / barrier / max-over-ranks / one-JSON-line protocol of main()
    on a stub step over gloo on CPU, so that the N > 1 plumbing is exercised where there is no GPU."""
    import torch.distributed as dist
    from multilingual_text_to_speech_amd import dist as D
    rank, world, _ = D.init(backend='gloo')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    torch.manual_seed(0)
    model = torch.nn.Linear(8, 8)
    D.broadcast_parameters(model)
    buckets = D.GradientBuckets(model.parameters(), bucket_bytes=64) if world > 1 else None
    x = torch.randn(args.batch, 8, generator=torch.Generator().manual_seed(1 + rank))

    def step():
        model.zero_grad()
        model(x).pow(2).mean().backward()
        if buckets is not None:
            buckets.all_reduce()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t = torch.tensor([time.perf_counter() - t0])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        frames = args.batch * args.frames * world * args.steps
        print(json.dumps({'metric': 'stub', 'value': round(frames / float(t), 1), 'unit': 'mel-frames/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * float(t) / args.steps, 3),
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'stub',
                          'config': {'workload': 'stub step (launch protocol test)', 'global_batch': args.batch * world,
                                     'parallelism': f'dp{world}'}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=PER_GPU_BATCH, help='per-GPU batch')
    ap.add_argument('--frames', type=int, default=T_FRAMES)
    ap.add_argument('--preset', default=PRESET)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='bf16: BASELINE configs[3] (contraction operands bf16, fp32 accumulate)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the batch-240 step roofline, the inference object and the PMC traffic passes')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--traffic-probe', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--chars', type=int, default=L_CHARS, help=argparse.SUPPRESS)          # traffic probe only: encoder length
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if args.traffic_probe:
        return traffic_probe(args)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: launch the ranks ourselves (one process per GPU) and relay rank 0's JSON line
        raise SystemExit(self_launch(args.gpus))
    if os.environ.get('MTTS_BENCH_STUB') == '1':
        return stub_main(args)

    from multilingual_text_to_speech_amd import _C, dist as D
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
    rank, world, local = D.init()
    if args.dtype == 'bf16':
        _C.set_precision('bf16')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the hot path has no CPU fallback)')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: pass the same N to torch.distributed.run and to --gpus')
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
    # steady state reached: one full pass of Python's garbage collector now, survivors frozen - otherwise the collector's first
    # generation-2 pass (85 ms of host time, the GPU idle meanwhile) lands ~10 steps into the run, i.e. inside the timed steps
    from multilingual_text_to_speech_amd.utils import settle_host_heap
    settle_host_heap()
    lib = _C.lib()
    n_samples = 16
    _C.check(lib.mtts_prof_begin(n_samples * args.steps, max(1, T // n_samples)), 'prof_begin')
    barrier()
    t0 = time.perf_counter()
    sync_each = os.environ.get('MTTS_BENCH_SYNC', '0') == '1'
    for _ in range(args.steps):
        loss = train_step(model, crit, opt, buckets, batch, hp)
        if sync_each:
            torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    tot_ms, cnt = ctypes.c_float(0), ctypes.c_int(0)
    _C.check(lib.mtts_prof_end(ctypes.byref(tot_ms), ctypes.byref(cnt)), 'prof_end')
    t_max = torch.tensor([dt], device=device)
    if world > 1:
        torch.distributed.all_reduce(t_max, op=torch.distributed.ReduceOp.MAX)
    dt = float(t_max.item())
    from multilingual_text_to_speech_amd.kernels import check_device_errors
    check_device_errors(device)

    if rank == 0:
        frames = B * T * world * args.steps
        d = model_dims(hp)
        H, Dm = d['H'], d['Dm']
        # dominant kernel of the forward decoder: ONE persistent launch (pdec_kernel) runs the attention LSTM + attention of all T
        # steps (shapes it does not take - batch > 64 - run the per-step launches, where the sampled kernel is the attention-LSTM
        # step).  Every sampled launch sits in a HIP-event bracket on its launch stream preceded by an EMPTY bracket that measures
        # what an event pair costs by itself (subtracted; rocprofv3 reports the bare kernel, see profiles/).
        A_, L_, ks_ = d['A'], L, d['ks']
        lib.mtts_prof_empty_ms.restype = ctypes.c_float
        raw_s = (tot_ms.value / max(cnt.value, 1)) * 1e-3
        empty_s = (float(lib.mtts_prof_empty_ms()) / max(cnt.value, 1)) * 1e-3
        avg_s = max(raw_s - empty_s, 1e-9)
        roof = step_roofline(model, hp, batch, B, L, T, args.preset, args.dtype)
        try:
            roof_bwd = backward_roofline(model, hp, batch, B, L, T, args.preset, args.dtype)
        except Exception as exc:      # reporting only
            roof_bwd = {'error': repr(exc)[:200]}
        persistent = cnt.value > 0 and cnt.value <= 2 * args.steps          # one sample per train step = the persistent launch
        if persistent:
            # algorithmic operands of chain A per step: recurrent LSTM weights + query / location parameters read once, per sample the
            # memory transform, the memory, alignment state and the cell state (SURVEY 8d's list restricted to these two layers)
            w_a = 4 * H * (Dm + H) + 4 * H + A_ * H + A_ * ks_ + 2 * A_
            act_a = L_ * A_ + L_ * Dm + 3 * L_ + 4 * H + 2 * Dm
            bytes_step = 4.0 * (w_a + B * act_a)
            flop_step = 2.0 * B * (4 * H * (Dm + H) + A_ * H + L_ * (A_ * ks_ + A_ + Dm))
            us_step = avg_s * 1e6 / T
            roof['kernels'] = {'persistent_attention_decoder': {
                'kernel': 'pdec_kernel: attention LSTM (recurrent part, weights stationary in LDS) + query + location-sensitive attention of ALL '
                          f'{T} steps in one launch; two grid barriers per step', 'bound': 'hbm',
                'avg_launch_us': round(avg_s * 1e6, 1), 'steps_per_launch': T, 'us_per_step': round(us_step, 2),
                'bytes_per_launch': bytes_step * T, 'achieved_GBps': round(bytes_step / (us_step * 1e-6) / 1e9, 1),
                'frac': round(bytes_step / (us_step * 1e-6) / 8e12, 4), 'peak_GBps': 8000.0,
                'flop_per_launch': flop_step * T, 'fp32_mfma_frac_of_157TF': round(flop_step / (us_step * 1e-6) / 157.3e12, 4),
                'event_bracket_us': round(raw_s * 1e6, 1), 'empty_bracket_us': round(empty_s * 1e6, 2), 'samples': cnt.value,
                'sampled_in': 'the timed train steps',
                'note': 'algorithmic bytes count the recurrent weights once per step although they never leave the chip: the HBM traffic '
                        'measured by the PMC passes (roofline.traffic) is below the algorithmic figure'}}
        else:
            k_rec = Dm + H
            flop = 2.0 * B * k_rec * 4 * H
            bytes_alg = 4.0 * (4 * H * k_rec + B * k_rec + B * 4 * H + 2 * 4 * H + 3 * B * H + B * 4 * H + B * H)
            achieved = flop / avg_s / 1e12 if cnt.value else 0.0
            roof['kernels'] = {'attention_lstm_step': {
                'kernel': 'attention-LSTM step: [B,Dm+H]x[4H,Dm+H]^T + LSTM cell + query partials (K-split gate GEMM + cell kernel: 2 launches per frame)', 'bound': 'mfma',
                'achieved_TFLOPs': round(achieved, 2), 'peak_TFLOPs': 157.3, 'frac': round(achieved / 157.3, 4), 'flop_per_launch': flop,
                'bytes_per_launch': bytes_alg, 'avg_launch_us': round(avg_s * 1e6, 2), 'event_bracket_us': round(raw_s * 1e6, 2),
                'empty_bracket_us': round(empty_s * 1e6, 2), 'hbm_frac_of_8TBps': round(bytes_alg / avg_s / 8e12, 4) if cnt.value else 0.0,
                'samples': cnt.value, 'sampled_in': 'the timed train steps (side streams active)'}}
        line = {
            'metric': f'mel-frames/sec (train, fwd+bwd+optimizer, batch {B}/GPU, {L} chars -> {T} frames)',
            'value': round(frames / dt, 1), 'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'params/{args.preset}' + (' (BASELINE configs[1])' if args.preset == PRESET else '') + f' train step, per-GPU batch {B}, L={L} -> T={T}, '
                                   f'{args.dtype}, random-init weights', 'gemm_core': GEMM_CORE[args.dtype],
                       'global_batch': B * world, 'parallelism': f'dp{world}',
                       'loss': float(loss.item())},
            'roofline': roof, 'roofline_bwd': roof_bwd,
        }
        if world == 1 and not args.no_secondary:
            # free the training step's memory first; each leg is reporting only and must never cost the headline number
            del model, opt, crit, batch
            torch.cuda.empty_cache()
            try:
                traffic, why = measure_traffic(args.preset, B, args.dtype)
                roof['traffic'] = traffic['bytes_per_step'] if traffic else None
                roof['traffic_detail'] = traffic if traffic else {'error': why}
                if traffic:      # what actually crossed the fabric per step against the same peak (the recurrent weights stay on chip)
                    roof['traffic_frac'] = round(traffic['bytes_per_step'] / (roof['us_per_step'] * 1e-6) / 8e12, 4)
            except Exception as exc:
                roof['traffic_detail'] = {'error': repr(exc)[:200]}
            try:      # matrix-pipe occupancy of the same decoder forward (one more PMC pass)
                mfma, why = measure_mfma(args.preset, B, args.dtype)
                roof['mfma_busy'] = mfma['mfma_frac_of_peak'] if mfma else None
                roof['mfma_detail'] = mfma if mfma else {'error': why}
            except Exception as exc:
                roof['mfma_detail'] = {'error': repr(exc)[:200]}
            # batch 240 = the valid batch next to the north star's 256 on ONE GPU; batch 40 = what one rank of BASELINE configs[3]
            # (global batch 320 over 8 GPUs, bf16) actually sees
            for key, nb, dt_ in (('roofline_b240', 240, 'f32'), ('roofline_b240_bf16', 240, 'bf16'), ('roofline_b40_bf16', 40, 'bf16')):
                try:
                    line[key] = secondary_step_roofline('generated_switching', nb, L_CHARS, 300, device, dt_)
                    torch.cuda.empty_cache()
                    traffic, why = measure_traffic('generated_switching', nb, dt_, timeout=180)
                    line[key]['traffic'] = traffic['bytes_per_step'] if traffic else None
                    line[key]['traffic_detail'] = traffic if traffic else {'error': why}
                except Exception as exc:
                    line.setdefault(key, {})['error'] = repr(exc)[:200]
            try:      # the per-rank train step of the 8-GPU configuration (configs[3]), bf16 and fp32 on this box
                line['train_b40_bf16'] = secondary_train_step('generated_switching', 40, L_CHARS, T_FRAMES, device, 'bf16')
                torch.cuda.empty_cache()
            except Exception as exc:
                line['train_b40_bf16'] = {'error': repr(exc)[:200]}
            _C.set_precision('fp32')
            try:
                line['roofline_L200'] = long_input_roofline(device)
                torch.cuda.empty_cache()
                traffic, why = measure_traffic(PRESET, PER_GPU_BATCH, 'f32', timeout=180, chars=200)      # (full-length rows: the probe does not vary the lengths)
                line['roofline_L200']['traffic'] = traffic['bytes_per_step'] if traffic else None
                line['roofline_L200']['traffic_detail'] = traffic if traffic else {'error': why}
            except Exception as exc:
                line['roofline_L200'] = {'error': repr(exc)[:200]}
            _C.set_precision('bf16' if args.dtype == 'bf16' else 'fp32')
            try:
                line['inference'] = inference_bench(device)
            except Exception as exc:
                line['inference'] = {'error': repr(exc)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            try:      # separate process + hard timeout: the baseline is reporting only, never lose the GPU number over it
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only'], capture_output=True,
                                   text=True, timeout=240, env={**os.environ, 'HIP_VISIBLE_DEVICES': ''})
                line['cpu_baseline'] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as exc:
                line['cpu_baseline'] = {'error': repr(exc)[:200]}
            rec = recorded_reference_baseline()
            if rec is not None:
                line['cpu_baseline']['reference_recorded'] = rec
                same = [r for r in rec['results'] if r['batch'] == B and r['frames'] == T and args.preset in r['config']]
                if same:      # the reference itself on this very configuration (build container): the like-for-like ratio
                    best = max(r['value'] for r in same)
                    line['cpu_baseline']['gpu_over_reference_recorded'] = {
                        'ratio': round(line['value'] / best, 1), 'reference_value': best, 'reference_cores': rec['cores'],
                        'config': f'{args.preset}, batch {B}, {T} frames (the bench configuration)'}
            if 'value' in line['cpu_baseline']:
                line['cpu_baseline']['gpu_over_port'] = round(line['value'] / line['cpu_baseline']['value'], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
