"""Several CONSECUTIVE training steps (every other parity test is one step): what a step leaves behind must be what the next step
of the reference starts from - updated parameters (and the product's re-packed stationary copies of them), BatchNorm running
statistics, both Adam moments, the decayed guided-attention tolerance (reference train.py:58-93, modules/tacotron2.py:439-441)."""
import pytest
import torch

from oracle import tacotron_oracle as O
from tests.helpers import assert_after_adam_close, assert_after_step_close, build_hip_model, golden_names, injected_masks, load_golden
from tests.test_gpu_more import _random_batch, make_draws

pytestmark = pytest.mark.gpu


def _hip_step(model, crit, opt, hp, batch, inj):
    from multilingual_text_to_speech_amd.masks import provider
    to = lambda t: None if t is None else t.cuda()
    provider.injected = inj
    try:
        opt.zero_grad(set_to_none=True)
        post, pre, stop, align, spk_pred, enc = model(to(batch['text']), batch['text_length'], to(batch['target']), batch['target_length'],
                                                      to(batch['speakers']), to(batch['languages']), 1.0)
    finally:
        provider.injected = None
    loss, _ = crit(batch['text_length'].cuda(), batch['target_length'].cuda(), pre, to(batch['target']), post, to(batch['target']), stop,
                   to(batch['stop_target']), align, to(batch['speakers']), spk_pred, enc, None)
    loss.backward()
    norm = opt.step(max_norm=hp.gradient_clipping)
    crit.update_states()
    torch.cuda.synchronize()
    return loss, norm, post, align


@pytest.mark.parametrize('name', golden_names('trajectory'))
def test_three_training_steps_match_the_reference_trajectory(name):
    """The HIP model + FusedAdam (mtts_clip_adam_step) + the fused loss reproduce what the REFERENCE recorded after each of three
    iterations of its own loop (oracle/make_golden.py run_trajectory): loss, gradient norm, every parameter, BatchNorm running
    statistics and both Adam moments, to 1e-4 relative."""
    from multilingual_text_to_speech_amd.modules.tacotron2 import TacotronLoss
    from multilingual_text_to_speech_amd.optim import FusedAdam
    from multilingual_text_to_speech_amd.params import Params as hp
    fx = load_golden(name)
    model = build_hip_model(fx)
    oc, cc = fx['optimizer'], fx['criterion']
    opt = FusedAdam(model.parameters(), lr=oc['lr'], weight_decay=oc['weight_decay'])
    crit = TacotronLoss(cc['steps'], cc['g0'], cc['gamma'])
    assert abs(hp.gradient_clipping - oc['clip']) < 1e-12
    params = dict(model.named_parameters())
    for i, st in enumerate(fx['steps']):
        assert abs(crit._g - st['guided_g']) < 1e-9
        loss, norm, post, align = _hip_step(model, crit, opt, hp, st, injected_masks(st, 'cuda'))
        assert abs(loss.item() - st['loss'].item()) <= 1e-4 * max(1.0, abs(st['loss'].item())), (i, loss.item(), st['loss'].item())
        assert (post.detach().cpu() - st['post']).abs().max().item() <= 1e-3
        assert (align.detach().cpu() - st['alignment']).abs().max().item() <= 1e-3
        assert abs(norm[0].item() - st['grad_norm']) <= 1e-4 * st['grad_norm'], (i, norm[0].item(), st['grad_norm'])
        sd = model.state_dict()
        for k, v in st['state_after'].items():
            if k.endswith(('running_mean', 'running_var')):
                torch.testing.assert_close(sd[k].cpu(), v, atol=1e-4, rtol=1e-4, msg=lambda m: f'{name} step {i} {k}: {m}')
            elif k.endswith('num_batches_tracked'):
                assert int(sd[k]) == int(v), k
            else:
                assert_after_step_close(sd[k], v, f'{name} step {i} {k}', oc['lr'], max_mult=0.25)
        for k, v in st['adam_after'].items():
            s = opt.state[params[k]]
            assert int(s['step']) == v['step']
            torch.testing.assert_close(s['exp_avg'].cpu(), v['exp_avg'], atol=2e-6, rtol=2e-3, msg=lambda m: f'{name} step {i} exp_avg {k}: {m}')
            torch.testing.assert_close(s['exp_avg_sq'].cpu(), v['exp_avg_sq'], atol=1e-9, rtol=4e-3, msg=lambda m: f'{name} step {i} exp_avg_sq {k}: {m}')
    assert crit.state_dict()['_g_steps'] == fx['steps'][-1]['criterion_after']['_g_steps']
    assert abs(crit.state_dict()['_g'] - fx['steps'][-1]['criterion_after']['_g']) < 1e-9


@pytest.mark.parametrize('preset,B,L,T', [('shared_training', 8, 30, 30), ('generated_switching', 10, 24, 20), (None, 8, 30, 30)])
def test_three_training_steps_at_real_widths_through_the_persistent_kernels(preset, B, L, T):
    """Real layer widths, batch <= 64: the teacher-forced decoder runs in the persistent weights-stationary kernels, whose packed
    copies of the recurrent weights must follow every optimizer step (a stale pack would reproduce step 1 and miss steps 2 and 3).
    Three steps of HIP model + FusedAdam against the oracle + torch.optim.Adam + clip_grad_norm_ on the CPU, fresh batch and
    draws per step, BatchNorm running statistics carried on both sides."""
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
    from multilingual_text_to_speech_amd.optim import FusedAdam
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    presets.apply(preset, speaker_number=7)
    torch.manual_seed(2)
    model = Tacotron().train()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    cfg = O.cfg_from_params(hp)
    ropt = torch.optim.Adam([sd[k] for k in names], lr=hp.learning_rate, weight_decay=hp.weight_decay)
    model.cuda()
    opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    g_tol, g_steps = hp.guided_attention_toleration, hp.guided_attention_steps
    params = dict(model.named_parameters())
    torch.set_flush_denormal(True)
    for step in range(3):
        text, tl, target, tgl, spk, lang = _random_batch(hp, B, L, T, seed=30 + step)
        stop_t = torch.zeros(B, T)
        for b in range(B):
            stop_t[b, max(int(tgl[b]) - hp.stop_frames, 0):] = 1.0
        inj, om = make_draws(hp, B, L, T, torch.Generator().manual_seed(40 + step), [True] * T)
        # ---- oracle side
        ropt.zero_grad()
        ref = O.tacotron_forward(sd, cfg, text, tl, target, tgl, spk, lang, torch.tensor([True] * T), om, True)
        rloss, _ = O.tacotron_loss(cfg, ref, tl, tgl, target, stop_t, spk, g_tol, g_steps)
        rloss.backward()
        rnorm = torch.nn.utils.clip_grad_norm_([sd[k] for k in names], hp.gradient_clipping)
        ropt.step()
        g_tol *= hp.guided_attention_gain
        g_steps = max(0, g_steps - 1)
        with torch.no_grad():
            for k, v in ref['bn_stats'].items():
                sd[k].copy_(v)
        # ---- product side
        batch = dict(text=text, text_length=tl, target=target, target_length=tgl, speakers=spk, languages=lang, stop_target=stop_t)
        loss, norm, post, align = _hip_step(model, crit, opt, hp, batch, {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inj.items()})
        # Step 0 is held to the one-step gates.  From step 1 on the two sides start from parameters that already differ to first order in
        # the allowed gradient error, batch norm over 8 x 30 rows divides such perturbations by small standard deviations, and the
        # difference grows step over step (observed at step 2: single mel elements 3e-2, first moments 4e-3 of their largest element).
        # What these later steps are for - a stale pack of the stationary recurrent weights (they move by ~3 % per Adam step), a wrong
        # moment or running-statistics carry - shows at the 1e-1 level; the bounds below sit between the two.
        first = step == 0
        assert abs(loss.item() - rloss.item()) <= (1e-4 if first else 1e-3) * max(1.0, abs(rloss.item())), (step, loss.item(), rloss.item())
        dpost = (post.detach().cpu() - ref['post'].detach()).double()
        rel_post = dpost.norm().item() / ref['post'].detach().double().norm().item()
        assert dpost.abs().max().item() <= (1e-3 if first else 1e-1) and rel_post <= (1e-3 if first else 1e-2), (step, dpost.abs().max().item(), rel_post)
        assert abs(norm[0].item() - float(rnorm)) <= (1e-3 if first else 5e-3) * float(rnorm), (step, norm[0].item(), float(rnorm))
        hsd = model.state_dict()
        bad = []
        for k in names:
            # encoder-side tensors (embedding, encoder convolutions / batch norms / BiLSTM): one ReLU unit of an encoder block within
            # rounding of zero takes different sides on the CPU and the GPU and moves that block's and everything upstream's gradients
            # by percents of their largest element (profiles/r06_defaults_seed_sweep.txt: the fp32 and fp64 CPU oracles do it to each
            # other); with three steps on fresh batches it happens somewhere in most runs: sanity bound only.
            enc_side = k.startswith(('_embedding', '_encoder'))
            s_, r = opt.state[params[k]], ropt.state[sd[k]]
            if first and not enc_side:
                try:
                    assert_after_adam_close(hsd[k], sd[k], r['exp_avg'], r['exp_avg_sq'], 1, f'{preset} step 0 {k}', hp.learning_rate, grad_tol=1e-3)
                except AssertionError as exc:
                    bad.append(str(exc))
            dp = hsd[k].detach().cpu().double() - sd[k].detach().double()
            relp = dp.norm().item() / max(sd[k].detach().double().norm().item(), 10.0 * hp.learning_rate * dp.numel() ** 0.5)
            dm = s_['exp_avg'].cpu().double() - r['exp_avg'].double()
            d, gmax = dm.abs().max().item(), r['exp_avg'].abs().max().item()
            relm = dm.norm().item() / max(r['exp_avg'].double().norm().item(), 1e-30)
            if relp > (2e-3 if first else 1e-2) * (5 if enc_side else 1):
                bad.append(f'{k}: parameter relative L2 {relp:.2e}')
            if (relm > 0.1) if enc_side else ((d > 2e-3 * gmax + 2e-9) if first else (relm > 3e-2)):
                bad.append(f'{k}: first moment off by {d:.2e} (largest element {gmax:.2e}, relative L2 {relm:.2e})')
        assert not bad, f'{preset} step {step}: ' + ' | '.join(bad[:10])
        for k, v in ref['bn_stats'].items():
            torch.testing.assert_close(hsd[k].cpu(), v, atol=1e-4 if first else 2e-3, rtol=1e-4 if first else 2e-3, msg=lambda m: f'{preset} step {step} {k}: {m}')
