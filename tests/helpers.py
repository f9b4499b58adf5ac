"""Shared test helpers: fixture loading and oracle replay."""
import glob
import os

import torch

from oracle import tacotron_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def golden_names(kind=None):
    names = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, '*.pt')))
    if kind == 'trajectory':          # several consecutive optimizer steps of the reference's training loop (oracle/make_golden.py)
        return [n for n in names if '3step' in n]
    names = [n for n in names if '3step' not in n]
    if kind == 'train':
        return [n for n in names if not n.endswith('_infer')]
    if kind == 'infer':
        return [n for n in names if n.endswith('_infer')]
    return names


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)


def cfg_of(fx):
    """Oracle cfg dict = reference defaults + the fixture's overrides."""
    from multilingual_text_to_speech_amd.params import Params, reset_defaults
    reset_defaults()
    Params.load_state_dict(fx['hp'])
    cfg = O.cfg_from_params(Params)
    reset_defaults()
    return cfg


def oracle_replay(fx, with_grads=False, state_dict=None, cfg=None):
    cfg = cfg_of(fx) if cfg is None else cfg
    sd = {k: v.clone() for k, v in (fx['state_dict'] if state_dict is None else state_dict).items()}
    if with_grads:
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
                v.requires_grad_(True)
    out = O.tacotron_forward(sd, cfg, fx['text'], fx['text_length'], fx['target'], fx['target_length'],
                             fx['speakers'], fx['languages'], fx['teacher'], fx['masks'], fx['train'])
    loss, parts = O.tacotron_loss(cfg, out, fx['text_length'], fx['target_length'], fx['target'],
                                  fx['stop_target'], fx['speakers'], fx['guided_g'])
    grads = None
    if with_grads:
        loss.backward()
        grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    return out, loss, parts, grads


def assert_after_step_close(got, want, what, lr=1e-3, max_mult=0.1):
    """Parameters after an Adam step.  An element whose gradient is of the size of Adam's eps (1e-8) moves by anything in [-lr, lr]
    depending on the last bits of that gradient (update = lr g / (|g| + eps) on the first step), so elementwise 1e-5 is not a
    property of a correct implementation; per tensor: relative L2 <= 1e-4 and no element further off than a tenth of one update.
    (Relative to the tensor's norm or - zero-initialised biases ARE their first updates - to the norm of one full update, lr per element,
    times 10: 1e-4 of that is 1e-3 of an update per element in the root mean square.)"""
    d = (got.detach().double().cpu() - want.double())
    rel = d.norm().item() / max(want.double().norm().item(), 10.0 * lr * want.numel() ** 0.5, 1e-12)
    assert rel <= 1e-4 and d.abs().max().item() <= max_mult * lr, f'{what}: relative L2 {rel:.2e}, max |delta| {d.abs().max().item():.2e}'


def assert_after_adam_close(got, want, exp_avg, exp_avg_sq, step, what, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_tol=1e-3):
    """Parameters after `step` Adam updates, ELEMENTWISE, to first order in the gradient error the one-step parity tests allow
    (grad_tol x max |g| per tensor and step): Adam's update lr m^ / (sqrt(v^) + eps) turns a gradient error dg into lr dg / (sqrt(v^) + eps),
    i.e. it AMPLIFIES errors on elements whose gradient is small against the tensor's largest (an embedding row the batch barely
    touches flips the sign of its whole update on a 1e-3 relative difference between two correct gradient evaluations), and damps
    them elsewhere.  `exp_avg` / `exp_avg_sq` are the reference side's moments after the step."""
    d = (got.detach().double().cpu() - want.detach().double()).abs()
    vhat = exp_avg_sq.double() / (1.0 - betas[1] ** step)
    gscale = (exp_avg.double() / (1.0 - betas[0] ** step)).abs().max().item()
    tol = 1e-6 + lr * torch.clamp(step * 2.0 * grad_tol * gscale / (vhat.sqrt() + eps), max=3.0 * step)
    bad = d > tol
    # beyond first order the map is chaotic for exactly those small-gradient elements (step k + 1 evaluates its gradient at parameters
    # that already differ): a few percent of an embedding table sit there.  Held: at most 3 % of a tensor beyond the first-order
    # bound, nothing further off than the largest move `step` updates can make, and the tensor as a whole to 2e-3 relative L2.
    frac = float(bad.double().mean())
    assert frac <= 0.03 and d.max().item() <= 3.0 * lr * step + 1e-6, (
        f'{what}: {int(bad.sum())} of {d.numel()} elements beyond the first-order bound, worst |delta| {d.max().item():.2e}')
    rel = d.norm().item() / max(want.double().norm().item(), 10.0 * lr * want.numel() ** 0.5, 1e-12)
    assert rel <= 2e-3, f'{what}: relative L2 {rel:.2e}'


# ------------------------------------------------------------------------------------------------
# HIP-side helpers (GPU tests)
# ------------------------------------------------------------------------------------------------

def build_hip_model(fx, device='cuda'):
    """Instantiate the product model for a fixture's hyper-parameters and load the reference weights."""
    from multilingual_text_to_speech_amd.params import Params, reset_defaults
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    reset_defaults()
    Params.load_state_dict(fx['hp'])
    model = Tacotron()
    missing, unexpected = model.load_state_dict(fx['state_dict'], strict=True), None
    model.to(device)
    model.train(fx['train'])
    return model


def injected_masks(fx, device='cuda'):
    """Convert the reference's recorded dropout multipliers to the product's uint8 / channel-last layout."""
    m = fx['masks']
    out = {}
    keep = lambda t: (t != 0).to(torch.uint8)
    for k, v in m.items():
        if k.startswith(('enc', 'post.')):
            out[k] = keep(v).permute(0, 2, 1).contiguous().to(device)
    teacher = fx.get('teacher')
    if teacher is not None:
        out['teacher'] = [bool(x) for x in teacher]
        T = len(teacher)
        i = 0
        while f'prenet.{i}' in m:
            pm = keep(m[f'prenet.{i}'])[:, :T].transpose(0, 1).contiguous()          # [T,B,P]
            if f'prenet_step.{i}' in m:
                ps = keep(m[f'prenet_step.{i}'])                                       # [T,B,P]
                for t in range(T):
                    if not out['teacher'][t]:
                        pm[t] = ps[t]
            out[f'dec.prenet.{i}'] = pm.to(device)
            i += 1
    else:
        i = 0
        while f'prenet_step.{i}' in m:
            out[f'dec.prenet.{i}'] = keep(m[f'prenet_step.{i}']).contiguous().to(device)
            i += 1
    for src, dst in (('att_lstm', 'dec.att_lstm'), ('gen_lstm', 'dec.gen_lstm'), ('att_lstm.h', 'dec.att_lstm.h'),
                     ('att_lstm.c', 'dec.att_lstm.c'), ('gen_lstm.h', 'dec.gen_lstm.h'), ('gen_lstm.c', 'dec.gen_lstm.c')):
        if src in m:
            out[dst] = keep(m[src]).contiguous().to(device)
    return out


def hip_forward(fx, model, device='cuda', tf=1.0):
    from multilingual_text_to_speech_amd.masks import provider
    provider.injected = injected_masks(fx, device)
    try:
        to = lambda t: None if t is None else t.to(device)
        return model(to(fx['text']), fx['text_length'], to(fx['target']), fx['target_length'], to(fx['speakers']),
                     to(fx['languages']), tf)
    finally:
        provider.injected = None
