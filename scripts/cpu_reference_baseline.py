"""CPU baseline from the REFERENCE ITSELF: imports /root/reference (only present in the build container)
and times its own train step - Tacotron.forward + TacotronLoss + backward + clip_grad_norm_(0.25) + Adam.step(), fp32,
train() mode, teacher forcing 1.0 - on SURVEY 8(d)'s synthetic inputs, all host cores.  Writes profiles/cpu_reference.json,
which bench.py attaches to its line as cpu_baseline.reference_recorded (kind "reference"; the live leg of bench.py times the
oracle port on the GPU box because /root/reference does not exist there).

    python scripts/cpu_reference_baseline.py [--reference /root/reference] [--out profiles/cpu_reference.json]
"""
import argparse
import json
import os
import platform
import sys
import time

import torch


def synthetic(hp, B, L, T, seed=1):
    g = torch.Generator().manual_seed(seed)
    V = hp.symbols_count() + 3
    text = torch.randint(3, V, (B, L), generator=g)
    target = torch.randn(B, hp.num_mels, T, generator=g)
    stop = torch.zeros(B, T)
    stop[:, T - hp.stop_frames:] = 1.0
    spk = torch.randint(0, max(hp.speaker_number, 1), (B,), generator=g) if hp.multi_speaker else None
    lang = (torch.arange(B) % hp.language_number) if hp.multi_language else None
    return text, torch.full((B,), L, dtype=torch.int64), target, torch.full((B,), T, dtype=torch.int64), stop, spk, lang


def run(name, json_path, B, L, T, steps, flush, mods, defaults):
    hp, Tacotron, TacotronLoss = mods
    hp.load_state_dict(defaults)
    if json_path:
        hp.load(json_path)
    hp.speaker_number = 91 if hp.multi_speaker else 0
    hp.language_number = len(hp.languages) if hp.multi_language else 0
    torch.set_flush_denormal(flush)
    torch.manual_seed(0)
    model = Tacotron().train()
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    opt = torch.optim.Adam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
    text, tl, target, tgl, stop, spk, lang = synthetic(hp, B, L, T)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        post, pre, st, align, spk_pred, enc = model(text, tl, target, tgl, spk, lang, 1.0)
        cls = model._reversal_classifier if hp.reversal_classifier else None
        loss, _ = crit(tl, tgl, pre, target, post, target, st, stop, align, spk, spk_pred, enc, cls)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), hp.gradient_clipping)
        opt.step()
        crit.update_states()
        times.append(time.perf_counter() - t0)
        print(f'  {name} B={B} flush={flush} step {it}: {times[-1]:.2f} s  loss {float(loss):.4f}', flush=True)
    timed = sorted(times[1:])
    med = timed[(len(timed) - 1) // 2]           # lower median: the box is shared, slow outliers are other tenants
    return dict(config=name, batch=B, chars=L, frames=T, flush_denormal=flush, warmup_steps=1, timed_steps=steps,
                seconds_per_step=round(med, 3), mel_frames_per_s=round(B * T / med, 1), best_seconds_per_step=round(timed[0], 3), all_steps_s=[round(t, 3) for t in times],
                params_M=round(sum(p.numel() for p in model.parameters()) / 1e6, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'cpu_reference.json'))
    ap.add_argument('--quick', action='store_true', help='B=8 case only')
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    import utils  # noqa: F401  (before modules.tacotron2: circular import in the reference)
    from modules.tacotron2 import Tacotron, TacotronLoss
    from params.params import Params as hp
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    defaults = dict(hp.state_dict())
    mods = (hp, Tacotron, TacotronLoss)
    shared = os.path.join(args.reference, 'params', 'shared_training.json')
    results = []
    results.append(run('Params defaults (LJ Speech, BASELINE configs[0])', None, 8, 120, 600, 3, True, mods, defaults))
    results.append(run('Params defaults (LJ Speech, BASELINE configs[0])', None, 8, 120, 600, 3, False, mods, defaults))
    if not args.quick:
        results.append(run('params/shared_training.json (BASELINE configs[1], the bench workload)', shared, 64, 120, 600, 2, True, mods, defaults))
    cpu = ''
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
    except Exception:
        pass
    doc = dict(kind='reference', what='reference modules.tacotron2.Tacotron + TacotronLoss train step (fwd + loss + bwd + clip_grad_norm_ + Adam), '
               'fp32, CPU, imported from /root/reference and executed in the build container', unit='mel-frames/s', cores=cores,
               torch_threads=torch.get_num_threads(), cpu=cpu, host=platform.node(), torch=torch.__version__, results=results,
               recorded=time.strftime('%Y-%m-%d %H:%M:%S'))
    with open(args.out, 'w') as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc)[:400])


if __name__ == '__main__':
    main()
