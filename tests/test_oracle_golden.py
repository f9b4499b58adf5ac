"""Pin the CPU oracle against fixtures recorded from the reference implementation itself
(oracle/make_golden.py).  CPU only."""
import os

import pytest
import torch

from oracle import tacotron_oracle as O
from tests.helpers import assert_after_step_close, cfg_of, golden_names, load_golden, oracle_replay

TOL = 2e-5     # fp32 CPU, same ATen kernels; differences are summation-order only


@pytest.mark.parametrize('name', golden_names('train'))
def test_forward_matches_reference(name):
    fx = load_golden(name)
    out, loss, parts, _ = oracle_replay(fx)
    for key in ('post', 'pre', 'stop', 'alignment', 'encoder_output'):
        torch.testing.assert_close(out[key], fx[key], atol=TOL, rtol=1e-4, msg=lambda m: f'{name}/{key}: {m}')
    if fx['speaker_prediction'] is not None:
        torch.testing.assert_close(out['speaker_prediction'], fx['speaker_prediction'], atol=TOL, rtol=1e-4)
    torch.testing.assert_close(loss, fx['loss'], atol=1e-5, rtol=1e-5)
    for k, v in fx['loss_parts'].items():
        assert abs(float(parts[k]) - v) < 1e-5, (name, k)


@pytest.mark.parametrize('name', [n for n in golden_names('train') if n != 'simple_eval'])
def test_gradients_and_bn_stats_match_reference(name):
    fx = load_golden(name)
    out, loss, parts, grads = oracle_replay(fx, with_grads=True)
    # aliased parameters (_prenet == _decoder._prenet, _attention == _decoder._attention) appear once in
    # named_parameters(); the oracle reads the un-prefixed names.
    for k, g in fx['grads'].items():
        assert k in grads, f'{name}: oracle has no gradient for {k}'
        torch.testing.assert_close(grads[k], g, atol=3e-5, rtol=2e-4, msg=lambda m: f'{name}/{k}: {m}')
    for k, v in fx['bn_stats'].items():
        torch.testing.assert_close(out['bn_stats'][k], v, atol=1e-5, rtol=1e-5, msg=lambda m: f'{name}/{k}: {m}')


@pytest.mark.parametrize('name', golden_names('infer'))
def test_inference_matches_reference(name):
    fx = load_golden(name)
    cfg = cfg_of(fx)
    sd = fx['state_dict']
    text = fx['text']
    L = text.shape[1]
    spk = fx['speakers']
    lang = fx['languages']
    if spk is not None:
        spk = spk.unsqueeze(1).expand(-1, L)
    emb = torch.nn.functional.embedding(text, sd['_embedding.weight'], padding_idx=0)
    enc = O.encode(sd, cfg, emb, torch.tensor([L]), lang, None, False)
    lang_ids = torch.argmax(lang, dim=2) if lang is not None else None
    mask = torch.ones(1, L, dtype=torch.bool)
    frames, _, _ = O.decode(sd, cfg, enc, mask, None, None, spk, lang_ids, fx['masks'], False,
                            max_frames=fx['n_frames'], stop_rule=True)
    post = O.postnet(sd, cfg, frames.transpose(1, 2), None, False)
    torch.testing.assert_close(post[0], fx['inference_output'], atol=TOL, rtol=1e-4)


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference checkout only exists in the build container')
@pytest.mark.parametrize('name', ['simple_train', 'generated_train'])
def test_committed_fixture_is_what_the_reference_produces(name, tmp_path):
    """Re-run oracle/make_golden.py (which imports and EXECUTES /root/reference) for one configuration and compare every tensor with
    the committed fixture: the golden vectors are the reference's own outputs, reproducibly."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'oracle', 'make_golden.py'), '--out', str(tmp_path), '--only', name],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    new = torch.load(tmp_path / f'{name}.pt', weights_only=False)
    old = load_golden(name)

    def same(a, b, path):
        if isinstance(a, torch.Tensor):
            assert a.shape == b.shape and torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6), path
        elif isinstance(a, dict):
            assert set(a) == set(b), path
            for k in a:
                same(a[k], b[k], f'{path}/{k}')
        elif isinstance(a, (list, tuple)):
            assert len(a) == len(b), path
            for i, (u, v) in enumerate(zip(a, b)):
                same(u, v, f'{path}[{i}]')
    same(old, new, name)


def _reference_run(tmp_path, *extra):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'oracle', 'make_golden.py'), '--out', str(tmp_path), *extra],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference checkout only exists in the build container')
@pytest.mark.parametrize('name', ['real_shared_training', 'real_generated_switching'])
def test_oracle_matches_the_reference_at_real_widths(name, tmp_path):
    """The committed fixtures are 16-wide toy models; every real-width GPU test trusts the oracle at attention kernel 31, A = 128,
    H = 1024, Dm = 544 / 288.  Here the REFERENCE runs at those widths (its own params/shared_training.json, batch 2 x 40 characters
    x 12 frames; params/generated_switching.json, batch 5 x 30 x 8 with the adversarial classifier) and the oracle replays its
    inputs and dropout draws: outputs 2e-5, every parameter gradient 3e-5 (+ 2e-4 relative), BatchNorm running statistics."""
    _reference_run(tmp_path, '--real', str(tmp_path), '--only', name)
    fx = torch.load(tmp_path / f'{name}.pt', weights_only=False)
    out, loss, parts, grads = oracle_replay(fx, with_grads=True)
    for key in ('post', 'pre', 'stop', 'alignment', 'encoder_output'):
        torch.testing.assert_close(out[key], fx[key], atol=TOL, rtol=1e-4, msg=lambda m: f'{name}/{key}: {m}')
    if fx['speaker_prediction'] is not None:
        torch.testing.assert_close(out['speaker_prediction'], fx['speaker_prediction'], atol=TOL, rtol=1e-4)
    torch.testing.assert_close(loss, fx['loss'], atol=1e-5, rtol=1e-5)
    assert set(fx['grads']) <= set(grads)
    for k, g in fx['grads'].items():
        torch.testing.assert_close(grads[k], g, atol=3e-5, rtol=2e-4, msg=lambda m: f'{name}/{k}: {m}')
    for k, v in fx['bn_stats'].items():
        torch.testing.assert_close(out['bn_stats'][k], v, atol=1e-5, rtol=1e-5, msg=lambda m: f'{name}/{k}: {m}')


def oracle_trajectory(fx, on_step=None):
    """Replay a recorded training trajectory: oracle forward / loss / autograd, torch's clip_grad_norm_ and torch.optim.Adam on the
    oracle's leaf tensors, BatchNorm running statistics carried from step to step, guided-attention state decayed like
    TacotronLoss.update_states (reference modules/tacotron2.py:439-441).  Yields what the reference recorded after every step."""
    cfg = cfg_of(fx)
    opt_cfg, crit = fx['optimizer'], fx['criterion']
    sd = {k: v.clone() for k, v in fx['state_dict'].items()}
    learn = [k for k, v in sd.items() if v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))]
    # aliased parameters appear under two names in the state dict (the oracle reads the un-prefixed ones): optimise each storage once
    ref_names = list(fx['steps'][0]['adam_after'])
    for k in ref_names:
        sd[k].requires_grad_(True)
    opt = torch.optim.Adam([sd[k] for k in ref_names], lr=opt_cfg['lr'], weight_decay=opt_cfg['weight_decay'])
    g, g_steps = crit['g0'], crit['steps']
    for i, st in enumerate(fx['steps']):
        assert abs(g - st['guided_g']) < 1e-9
        opt.zero_grad()
        out = O.tacotron_forward(sd, cfg, st['text'], st['text_length'], st['target'], st['target_length'], st['speakers'],
                                 st['languages'], st['teacher'], st['masks'], True)
        loss, parts = O.tacotron_loss(cfg, out, st['text_length'], st['target_length'], st['target'], st['stop_target'],
                                      st['speakers'], g, g_steps)
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_([sd[k] for k in ref_names], opt_cfg['clip'])
        opt.step()
        g *= crit['gamma']
        g_steps = max(0, g_steps - 1)
        with torch.no_grad():
            for k, v in out['bn_stats'].items():
                sd[k].copy_(v)
        yield i, st, out, loss, float(norm), sd, opt


@pytest.mark.parametrize('name', golden_names('trajectory'))
def test_three_training_steps_match_the_reference(name):
    """What one-step fixtures cannot see: the parameters, BatchNorm running statistics and Adam moments the reference holds AFTER each
    of three consecutive iterations of its loop (train.py:58-93), and the decayed guided-attention tolerance feeding the next loss."""
    fx = load_golden(name)
    for i, st, out, loss, norm, sd, opt in oracle_trajectory(fx):
        torch.testing.assert_close(loss.detach(), st['loss'], atol=2e-5, rtol=2e-5, msg=lambda m: f'{name} step {i} loss: {m}')
        for key in ('post', 'alignment'):
            torch.testing.assert_close(out[key].detach(), st[key], atol=5e-5, rtol=1e-3, msg=lambda m: f'{name} step {i} {key}: {m}')
        assert abs(norm - st['grad_norm']) <= 1e-4 * st['grad_norm'], (i, norm, st['grad_norm'])
        for k, v in st['adam_after'].items():
            s = opt.state[sd[k]]
            assert_after_step_close(sd[k], st['state_after'][k], f'{name} step {i} {k}', fx['optimizer']['lr'])
            torch.testing.assert_close(s['exp_avg'], v['exp_avg'], atol=1e-6, rtol=1e-3, msg=lambda m: f'{name} step {i} exp_avg {k}: {m}')
            torch.testing.assert_close(s['exp_avg_sq'], v['exp_avg_sq'], atol=1e-9, rtol=2e-3, msg=lambda m: f'{name} step {i} exp_avg_sq {k}: {m}')
        for k, v in st['state_after'].items():
            if k.endswith(('running_mean', 'running_var')):
                torch.testing.assert_close(sd[k], v, atol=1e-5, rtol=1e-5, msg=lambda m: f'{name} step {i} {k}: {m}')
    assert fx['steps'][-1]['criterion_after']['_g_steps'] == fx['criterion']['steps'] - len(fx['steps'])
