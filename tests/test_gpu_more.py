"""Further GPU parity: inference vs the reference fixtures, HIP vs CPU oracle at the real layer widths, and
size-independent properties at the BASELINE shapes."""
import pytest
import torch

from oracle import tacotron_oracle as O
from tests.helpers import build_hip_model, golden_names, injected_masks, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', golden_names('infer'))
def test_inference_matches_reference_fixture(name):
    """Batch-1 free-running synthesis incl. the stop rule and (generated encoder) per-character language blending."""
    from multilingual_text_to_speech_amd.masks import provider
    fx = load_golden(name)
    model = build_hip_model(fx)
    T = fx['n_frames']
    m = injected_masks(fx)
    # the provider is asked for max_output_length steps; pad the recorded draws with ones
    for k in list(m):
        if k.startswith('dec.prenet'):
            full = torch.ones(model._decoder._max_frames, *m[k].shape[1:], dtype=torch.uint8, device='cuda')
            full[:m[k].shape[0]] = m[k]
            m[k] = full
    provider.injected = m
    try:
        to = lambda t: None if t is None else t.to('cuda')
        out = model.inference(fx['text'][0].clone().to('cuda'), to(fx['speakers']), to(fx['languages']))
    finally:
        provider.injected = None
    assert out.shape == fx['inference_output'].shape == (model._decoder._output_dim, T)
    err = (out.cpu() - fx['inference_output']).abs().max().item()
    assert err <= 1e-3, f'{name}: max |delta| = {err:.3e}'


def _random_batch(hp, B, L, T, seed=5, ragged=True):
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(3, hp.symbols_count() + 3, (B, L), generator=g)
    tl = torch.full((B,), L, dtype=torch.int64)
    tgl = torch.full((B,), T, dtype=torch.int64)
    if ragged:
        tl = torch.sort(torch.randint(L // 2, L + 1, (B,), generator=g), descending=True).values; tl[0] = L
        tgl = torch.randint(T // 2, T + 1, (B,), generator=g); tgl[0] = T
    target = torch.randn(B, hp.num_mels, T, generator=g)
    for b in range(B):
        text[b, tl[b]:] = 0
        target[b, :, tgl[b]:] = 0
    spk = torch.randint(0, hp.speaker_number, (B,), generator=g) if hp.multi_speaker else None
    lang = (torch.arange(B) % hp.language_number) if hp.multi_language else None
    return text, tl, target, tgl, spk, lang


@pytest.mark.parametrize('preset,B,L,T,over', [
    ('shared_training', 4, 24, 10, {}), ('generated_switching', 10, 24, 10, {}),
    ('shared_training', 1, 1, 1, {}),                                # one token, one frame, one sample
    ('shared_training', 17, 33, 2, {}),                              # odd batch (two 16-row MFMA tiles, one nearly empty)
    ('generated_switching', 5, 2, 3, {}),                            # inputs shorter than every convolution kernel
    ('shared_training', 3, 150, 3, {'attention_dimension': 64}),     # long input + narrow attention: <G=16, NE4=16, NMT=4> kernel
    ('shared_training', 2, 201, 2, {}),                              # synthesis-length input through the training path
    ('shared_training', 2, 40, 3, {'attention_dimension': 96, 'attention_kernel_size': 15})])   # generic attention kernels
def test_hip_matches_oracle_at_real_widths(preset, B, L, T, over):
    """Real layer widths (512/1024/...), small batch and lengths so the CPU oracle finishes in seconds; eval mode with the
    prenet dropout (always on) injected; BN running stats randomised so eval-mode activations stay bounded.  The tiny
    cases are the degenerate ends of the shape range (single token / frame / sample, ragged odd batches)."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    from multilingual_text_to_speech_amd.masks import provider
    presets.apply(preset, speaker_number=7, **over)
    torch.manual_seed(0)
    model = Tacotron()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('running_var'):
                v.copy_(torch.empty(v.shape).uniform_(300.0, 900.0, generator=g) if '_encoder' in k and preset != 'shared_training'
                        else torch.empty(v.shape).uniform_(0.5, 1.5, generator=g))
    model.eval()
    text, tl, target, tgl, spk, lang = _random_batch(hp, B, L, T, ragged=L >= 4)
    keep = {f'dec.prenet.{i}': (torch.rand(T, B, hp.prenet_dimension, generator=g) >= hp.dropout).to(torch.uint8) for i in range(2)}
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.cfg_from_params(hp)
    masks = {f'prenet.{i}': torch.cat((keep[f'dec.prenet.{i}'].transpose(0, 1).float() / (1 - hp.dropout),
                                       torch.ones(B, 1, hp.prenet_dimension)), 1) for i in range(2)}
    ref = O.tacotron_forward(sd, cfg, text, tl, target, tgl, spk, lang, torch.ones(T, dtype=torch.bool), masks, False)
    model.cuda()
    provider.injected = {**{k: v.cuda() for k, v in keep.items()}, 'teacher': [True] * T}
    try:
        to = lambda t: None if t is None else t.cuda()
        post, pre, stop, align, spk_pred, enc = model(to(text), tl, to(target), tgl, to(spk), to(lang), 1.0)
    finally:
        provider.injected = None
    for name, a, b in (('encoder', enc, ref['encoder_output']), ('alignment', align, ref['alignment']), ('pre', pre, ref['pre']),
                       ('post', post, ref['post'])):
        err = (a.cpu() - b).abs().max().item()
        assert err <= 1e-3, f'{preset}/{name}: max |delta| = {err:.3e}'


def test_full_size_properties_and_schedule_equivalence():
    """BASELINE shape (shared_training, batch 64, 120 -> 600): alignment rows are probability vectors supported on the valid
    characters, padded outputs are masked, and the hoisted ('fast') schedule equals the step-by-step one on a prefix."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    from multilingual_text_to_speech_amd import decoder_ops as D
    presets.apply('shared_training')
    torch.manual_seed(0)
    model = Tacotron().cuda().train()
    B, L, T = 64, 120, 600
    text, tl, target, tgl, spk, lang = _random_batch(hp, B, L, T)
    with torch.no_grad():
        post, pre, stop, align, _, enc = model(text.cuda(), tl, target.cuda(), tgl, None, lang.cuda(), 1.0)
    assert torch.isfinite(post).all() and torch.isfinite(align).all()
    rows = align.sum(-1)
    assert (rows - 1).abs().max().item() < 1e-4
    lm = (torch.arange(L)[None, :] < tl[:, None]).cuda()
    assert align.masked_select(~lm[:, None, :].expand_as(align)).abs().max().item() == 0.0
    tm = (torch.arange(T)[None, :] < tgl[:, None]).cuda()
    assert post.masked_select(~tm[:, None, :].expand_as(post)).abs().max().item() == 0.0
    assert (stop.masked_select(~tm) == 1000).all()

    # fast vs general schedule on a 40-frame prefix with identical masks (eval: only the prenet dropout is live)
    model.eval()
    dec = model._decoder
    Ts = 40
    w = D.decoder_weights(dec, model._attention, model._prenet)
    masks = dec._step_masks(Ts, B, 'cuda')
    with torch.no_grad():
        memory = dec._memory(enc, None, lang.cuda().unsqueeze(1).expand(-1, L))
        tgt = target[:, :, :Ts].transpose(1, 2).contiguous().cuda()
        outs = []
        for allow in (True, False):
            cfg = dict(dec._cfg(), allow_fast=allow)
            outs.append(D.decode_train(memory, tgt, tl, [True] * Ts, masks, cfg, w))
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2e-4


@pytest.mark.parametrize('preset,B,L,T,over', [
    ('shared_training', 4, 20, 9, {}), ('generated_switching', 5, 20, 9, {}),
    ('shared_training', 3, 150, 3, {'attention_dimension': 64}),     # long input, narrow attention (other template instances)
    ('shared_training', 2, 40, 3, {'attention_dimension': 96, 'attention_kernel_size': 15}),     # generic attention kernels
    ('shared_training', 2, 5, 2, {})])                               # tiny sizes through the full backward (BatchNorm over 10 rows)
def test_train_step_gradients_match_oracle_at_real_widths(preset, B, L, T, over):
    """Full train step (loss + backward) at the real layer widths against the CPU oracle's autograd: exercises the
    MFMA attention kernels, the packed-operand step kernels and the two-stream schedules that the small fixtures bypass."""
    run_train_step_case(preset, B, L, T, over)


# bf16 gradients against the same-rounding oracle: relative L2 per parameter tensor.  The error is NOT made in the backward: the post-net
# output of the two runs already differs by ~1 % (tolerance 2e-2 above: rounding flips amplified by five batch norms), so d(loss)/d(post)
# and with it EVERY gradient carries that relative difference in a random direction (observed: norm ratio 1.000 +- 0.003,
# 1 - cosine 1.1e-4 for every tensor = 1.5e-2 relative L2).  3e-2 bounds it; a wrong kernel is off by O(1) in some tensor.
BF16_GRAD_TOL = 3e-2
BF16_ORACLE_SITES = frozenset({'conv', 'bilstm_in', 'lstm', 'memory', 'loc', 'prenet', 'proj', 'linear'})


def make_draws(hp, B, L, T, g, teacher):
    """Every dropout / zoneout draw of one train-mode step as the product's injected uint8 keep flags (`inj`, channel-last) and as
    the oracle's multipliers (`om`, reference layout), from generator `g`."""
    keep = lambda *shape, p: (torch.rand(*shape, generator=g) >= p).to(torch.uint8)
    H, P = hp.decoder_dimension, hp.prenet_dimension
    inj = {'teacher': teacher, 'dec.att_lstm': keep(T, B, H, p=hp.dropout_hidden), 'dec.gen_lstm': keep(T, B, H, p=hp.dropout_hidden)}
    zone = hp.decoder_regularization == 'zoneout'
    if zone:
        for cell in ('att_lstm', 'gen_lstm'):
            inj[f'dec.{cell}.h'] = keep(T, B, H, p=hp.zoneout_hidden)
            inj[f'dec.{cell}.c'] = keep(T, B, H, p=hp.zoneout_cell)
    inj.update({f'dec.prenet.{i}': keep(T, B, P, p=hp.dropout) for i in range(2)})
    grouped = hp.encoder_type in ('generated', 'convolutional')       # reference modules/tacotron2.py:299-303: block dropout 0.05
    G = (hp.language_number if hp.multi_language else 1) if grouped else 1
    if grouped:
        from multilingual_text_to_speech_amd.modules.encoder import _LAYERS
        for i, (k, d, hw) in enumerate(_LAYERS):
            inj[f'enc.{i}'] = keep(B // G, L, G * hp.encoder_dimension * (2 if hw else 1), p=0.05)
    else:
        inj.update({f'enc.{i}': keep(B, L, hp.encoder_dimension, p=hp.dropout) for i in range(hp.encoder_blocks)})
    nb = hp.postnet_blocks
    inj.update({f'post.{i}': keep(B, T, hp.postnet_dimension if i < nb - 1 else hp.num_mels, p=hp.dropout) for i in range(nb)})

    mult = lambda m, p: m.float() / (1 - p)
    om = {'att_lstm': mult(inj['dec.att_lstm'], hp.dropout_hidden), 'gen_lstm': mult(inj['dec.gen_lstm'], hp.dropout_hidden)}
    if zone:
        for cell in ('att_lstm', 'gen_lstm'):
            om[f'{cell}.h'] = mult(inj[f'dec.{cell}.h'], hp.zoneout_hidden)
            om[f'{cell}.c'] = mult(inj[f'dec.{cell}.c'], hp.zoneout_cell)
    for i in range(2):
        om[f'prenet.{i}'] = torch.cat((mult(inj[f'dec.prenet.{i}'], hp.dropout).transpose(0, 1), torch.ones(B, 1, P)), 1)
        om[f'prenet_step.{i}'] = mult(inj[f'dec.prenet.{i}'], hp.dropout)          # free-running steps draw per step ([T,B,P])
    for k, v in inj.items():
        if k.startswith('enc.'):
            om[k] = mult(v, 0.05 if grouped else hp.dropout).permute(0, 2, 1)
        if k.startswith('post.'):
            om[k] = mult(v, hp.dropout).permute(0, 2, 1)
    return inj, om


def run_train_step_case(preset, B, L, T, over, check_grads=True, ragged=None, teacher=None, bf16=False, seed=9, fp64_spread=False):
    """One train-mode step of the product on the GPU against the CPU oracle with identical dropout draws: outputs, loss and
    (check_grads) the gradient of every parameter.  Shared by the chunk-boundary tests in test_gpu_chunks.py."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
    from multilingual_text_to_speech_amd.masks import provider
    presets.apply(preset, speaker_number=7, **over)
    torch.manual_seed(1)
    model = Tacotron().train()
    text, tl, target, tgl, spk, lang = _random_batch(hp, B, L, T, seed=seed, ragged=(L >= 4) if ragged is None else ragged)
    stop_t = torch.zeros(B, T)
    for b in range(B):
        stop_t[b, max(int(tgl[b]) - hp.stop_frames, 0):] = 1.0
    g = torch.Generator().manual_seed(21)
    teacher = [True] * T if teacher is None else [bool(x) for x in teacher]
    inj, om = make_draws(hp, B, L, T, g, teacher)
    zone = hp.decoder_regularization == 'zoneout'
    # ---- oracle (CPU autograd)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
            v.requires_grad_(True)
    cfg = O.cfg_from_params(hp)
    torch.set_flush_denormal(True)               # CPU speed only: identical output (SURVEY 8c recipe 5)
    O.BF16_SITES = BF16_ORACLE_SITES if bf16 else frozenset()      # bf16 path: the oracle rounds the same contraction operands
    if bf16 and check_grads:      # ... and its backward rounds what the product's backward rounds (see oracle._RoundedLinear): the batched
        # GEMMs and - since round 5 - the per-step input-gradient products dG W^T of both decoder LSTMs (bf16 pair tiles, skinny_body.h
        # PK = 3: dG and the transposed recurrent weights RNE-rounded, fp32 accumulation).  A mixed-teacher-forcing step keeps fp32
        # per-step products (general schedule): there only the weight gradients of the LSTMs round.
        fast = all(teacher)
        O.BF16_BWD_SITES = BF16_ORACLE_SITES - ({'loc'} if fast else {'lstm', 'loc'})
        O.BF16_BWD_WGRAD_ONLY = frozenset() if fast else frozenset({'lstm'})
    spread64 = None
    try:
        with torch.set_grad_enabled(check_grads):
            ref = O.tacotron_forward(sd, cfg, text, tl, target, tgl, spk, lang, torch.tensor(teacher), om, True)
            rloss, _ = O.tacotron_loss(cfg, ref, tl, tgl, target, stop_t, spk, hp.guided_attention_toleration)
        if check_grads:
            rloss.backward()
        if fp64_spread and check_grads:
            # the SAME oracle, the SAME operand rounding, carried in fp64: two correct evaluations of one function that differ only in
            # the precision / order of their sums.  Their per-tensor gradient distance is what rounding-decision flips cost in THIS
            # model; the product is held to a small multiple of it (below) instead of to an asserted constant.
            sd64 = {k: v.detach().double() for k, v in sd.items()}
            for k, v in sd64.items():
                if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
                    v.requires_grad_(True)
            om64 = {k: v.double() for k, v in om.items()}
            ref64 = O.tacotron_forward(sd64, cfg, text, tl, target.double(), tgl, spk, lang, torch.tensor(teacher), om64, True)
            loss64, _ = O.tacotron_loss(cfg, ref64, tl, tgl, target.double(), stop_t.double(), spk, hp.guided_attention_toleration)
            loss64.backward()
            spread64 = {k: ((sd[k].grad.double() - v.grad).norm() / v.grad.norm().clamp_min(1e-300)).item() for k, v in sd64.items() if v.grad is not None}
            spread64['__post__'] = ((ref['post'].detach().double() - ref64['post'].detach()).norm() / ref64['post'].detach().norm()).item()
            grads64 = {k: v.grad for k, v in sd64.items() if v.grad is not None}
            print('fp32 oracle vs fp64 oracle (same bf16 operand rounding), worst per-tensor relative L2 of the gradients:',
                  [(k, round(v, 5)) for k, v in sorted(spread64.items(), key=lambda kv: -kv[1])[:6]])
    finally:
        O.BF16_SITES = O.BF16_BWD_SITES = O.BF16_BWD_WGRAD_ONLY = frozenset()

    # ---- HIP
    model.cuda()
    provider.injected = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inj.items()}
    from multilingual_text_to_speech_amd import _C
    if bf16:
        _C.set_precision('bf16')
    try:
        to = lambda t: None if t is None else t.cuda()
        post, pre, stop, align, spk_pred, enc = model(to(text), tl, to(target), tgl, to(spk), to(lang), 1.0 if all(teacher) else 0.5)
    finally:
        provider.injected = None
        _C.set_precision('fp32')
    if bf16:
        # stated bf16 tolerance against the oracle WITH THE SAME OPERAND ROUNDING: what is left is summation order, which flips an
        # occasional rounding decision downstream (one flipped bf16 operand = 2^-9 relative on that element)
        errs = {}
        for name, a, b in (('encoder', enc, ref['encoder_output']), ('pre', pre, ref['pre']), ('alignment', align, ref['alignment']), ('post', post, ref['post'])):
            d = (a.detach().cpu() - b.detach()).double()
            errs[name] = (round((d.norm() / b.detach().double().norm()).item(), 6), round(d.abs().max().item(), 6))
        print('bf16 vs bf16-operand oracle (relative L2, max |delta|):', errs)
        # encoder output, decoder mels and alignments: 2e-3.  The post-net output sits behind five more conv + batch-norm (batch
        # statistics) + tanh layers of a random-init model, which amplify the decoder's residual ~10x (observed 5.7e-3): 2e-2.
        tol = {'encoder': 2e-3, 'pre': 2e-3, 'alignment': 2e-3, 'post': 2e-2}
        if hp.encoder_type == 'generated':
            # 14 generated conv blocks, each with a batch norm over (B / G) * L rows: a flipped operand rounding is amplified layer by
            # layer like behind the post-net (observed 4.8e-3 on the encoder output, 2.2e-3 on the decoder mels at B = 40, L = 30)
            tol.update(encoder=8e-3, pre=4e-3)
        assert all(v[0] <= tol[k] for k, v in errs.items()), f'{preset} B={B} T={T} bf16: {errs}'
        assert (post.detach().cpu() - ref['post'].detach()).abs().max().item() > 0      # (not bit-identical: different summation order)
        if not check_grads:
            return
        # bf16 GRADIENTS against the oracle's autograd through the same operand rounding (straight-through: the derivative of the
        # rounding is 1).  What differs on the product side: its batched backward GEMMs round dY / W / X to bf16 as well (the
        # per-step recurrence products stay fp32), i.e. one more 2^-9 relative perturbation per contraction operand, so the bound is
        # a norm-wise one per tensor: relative L2 <= BF16_GRAD_TOL.  A wrong kernel (missing term, wrong mask, wrong scale) is off by
        # O(1) in at least one tensor; the cosine >= 0.98 check this replaces allowed a 20 % error.
        crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
        _C.set_precision('bf16')
        try:
            loss, _ = crit(tl.cuda(), tgl.cuda(), pre, target.cuda(), post, target.cuda(), stop, stop_t.cuda(), align, to(spk), spk_pred, enc, None)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            _C.set_precision('fp32')
        assert abs(loss.item() - rloss.item()) <= 2e-3 * max(1.0, abs(rloss.item()))
        worst, shape = {}, {}
        for k, p in model.named_parameters():
            r = sd[k].grad
            assert r is not None and p.grad is not None, k
            g64, r64 = p.grad.cpu().double().flatten(), r.double().flatten()
            worst[k] = ((g64 - r64).norm() / r64.norm().clamp_min(1e-12)).item()
            shape[k] = (round((g64.norm() / r64.norm().clamp_min(1e-12)).item(), 5),
                        round(1.0 - (torch.dot(g64, r64) / (g64.norm() * r64.norm()).clamp_min(1e-30)).item(), 7))
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
        print('bf16 gradients vs bf16-operand oracle, worst relative L2 (norm ratio, 1 - cosine):', [(k, round(v, 5), shape[k]) for k, v in top])
        # generated encoder: the gradients of the parameter GENERATORS (tens of values that steer whole convolution kernels through 14
        # batch-normed layers) move by percents between two CORRECT evaluations of this function (the same-rounding oracle carried in
        # fp32 and in fp64: `spread64`).  No flat constant (round 5 used 1.5e-1, which cannot see a 10 % error in a generator gradient):
        # per tensor, the product may be as far from the fp32 oracle as max(BF16_GRAD_TOL, 3x spread) + spread, spread = the distance
        # between the two oracles on that tensor (product -> fp64 oracle <= max(BF16_GRAD_TOL, 3x spread), fp64 -> fp32 oracle = spread); the norm ratio is held to
        # the same number and 1 - cosine to its square.
        if hp.encoder_type == 'generated':
            assert spread64 is not None, 'generated-encoder bf16 gradient cases need fp64_spread=True (the bound is per tensor)'
        gtol = {k: max(BF16_GRAD_TOL, 3.0 * spread64[k]) + spread64[k] if spread64 is not None else BF16_GRAD_TOL for k in worst}
        bad = {k: (v, shape[k], gtol[k]) for k, v in worst.items()
               if v > gtol[k] or abs(shape[k][0] - 1.0) > max(2e-2, gtol[k]) or shape[k][1] > max(5e-4, gtol[k] ** 2)}
        assert not bad, f'{preset} B={B} T={T} bf16 gradients (relative L2, (norm ratio, 1 - cosine), bound): {bad}'
        if spread64 is not None:
            loose = {k: round(t, 4) for k, t in gtol.items() if t > BF16_GRAD_TOL}
            print(f'{len(loose)} of {len(gtol)} tensors carry a bound above {BF16_GRAD_TOL} (from the fp32 / fp64 oracles\' own distance):', loose)
        if spread64 is not None:
            # DEMONSTRATION that the loose bound above is the model's, not the kernels': the fp32 oracle and the fp64 oracle (same
            # rounding sites, both correct) are as far from each other as the product is from either.  Per tensor the product's
            # distance to the fp64 oracle must stay within 3x the two oracles' distance (or within BF16_GRAD_TOL where the oracles
            # happen to agree better than that), and the tensors that exceed BF16_GRAD_TOL must be tensors on which the oracles
            # disagree by more than a third of it as well.
            prod64 = {k: ((p.grad.cpu().double() - grads64[k]).norm() / grads64[k].norm().clamp_min(1e-300)).item() for k, p in model.named_parameters()}
            post64 = ((post.detach().cpu().double() - ref64['post'].detach()).norm() / ref64['post'].detach().norm()).item()
            top64 = sorted(prod64.items(), key=lambda kv: -kv[1])[:6]
            print(f'post-net output: product vs fp64 oracle {post64:.3e}, fp32 oracle vs fp64 oracle {spread64["__post__"]:.3e}')
            print('gradients, relative L2 to the fp64 same-rounding oracle (product, fp32 oracle):', [(k, round(v, 5), round(spread64[k], 5)) for k, v in top64])
            off = {k: (v, spread64[k]) for k, v in prod64.items() if v > max(BF16_GRAD_TOL, 3.0 * spread64[k])}
            assert not off, f'{preset} B={B} T={T}: product further from the fp64 oracle than 3x the fp32 oracle is: {off}'
            assert max(spread64[k] for k in prod64) >= BF16_GRAD_TOL / 3 or max(prod64.values()) <= BF16_GRAD_TOL
        return
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    loss, _ = crit(tl.cuda(), tgl.cuda(), pre, target.cuda(), post, target.cuda(), stop, stop_t.cuda(), align, to(spk), spk_pred, enc, None)
    if check_grads:
        loss.backward()
    torch.cuda.synchronize()
    for name, a, b in (('post', post, ref['post']), ('pre', pre, ref['pre']), ('alignment', align, ref['alignment'])):
        err = (a.detach().cpu() - b.detach()).abs().max().item()
        assert err <= 1e-3, f'{preset} B={B} T={T} {name}: max |delta| = {err:.3e}'
    assert abs(loss.item() - rloss.item()) <= 1e-4 * max(1.0, abs(rloss.item()))
    if not check_grads:
        return
    bad = []
    for k, p in model.named_parameters():
        r = sd[k].grad
        assert r is not None and p.grad is not None, k
        err = (p.grad.cpu() - r).abs().max().item()
        tol = 1e-3 * r.abs().max().item() + 2e-5
        if err > tol:
            bad.append(f'{k}: max |delta| {err:.3e} > {tol:.3e}')
    assert not bad, f'{preset} B={B} L={L} T={T}: {len(bad)} gradients off: ' + '; '.join(bad[:12])


@pytest.mark.parametrize('preset,B', [('generated_switching', 40), ('shared_training', 72)])
def test_packed_operands_equal_row_major_for_odd_batches(preset, B, monkeypatch):
    """Batches that are not multiples of 16 / exceed one 64-row tile: the MFMA-tile-order ('packed') operand path must give
    the same outputs and gradients as the plain row-major path (regression test for an out-of-range row tile)."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    from multilingual_text_to_speech_amd.masks import provider
    presets.apply(preset, speaker_number=7)
    torch.manual_seed(3)
    model = Tacotron().cuda().train()
    L, T = 30, 12
    text, tl, target, tgl, spk, lang = _random_batch(hp, B, L, T, seed=4)
    results = []
    for no_pack in ('0', '1'):
        monkeypatch.setenv('MTTS_NO_PACK', no_pack)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(100)                      # identical dropout draws
        to = lambda t: None if t is None else t.cuda()
        post, pre, stop, align, _, enc = model(to(text), tl, to(target), tgl, to(spk), to(lang), 1.0)
        (post.square().mean() + align.square().sum() * 1e-3 + stop.clamp(-5, 5).mean()).backward()
        torch.cuda.synchronize()
        results.append((post.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    (p0, g0), (p1, g1) = results
    assert torch.isfinite(p0).all()
    assert (p0 - p1).abs().max().item() <= 1e-5
    for k in g0:
        assert (g0[k] - g1[k]).abs().max().item() <= 1e-4 * g0[k].abs().max().item() + 1e-6, k


def test_fused_loss_and_adam_match_torch():
    """mtts_tacotron_loss vs the torch formulas of the reference loss; mtts_clip_adam_step vs clip_grad_norm_ + torch Adam."""
    from multilingual_text_to_speech_amd.optim import TacotronLossFn, FusedAdam
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    B, M, T, L = 5, 8, 37, 21
    pre = torch.randn(B, M, T, generator=g).cuda().requires_grad_(True)
    post = torch.randn(B, M, T, generator=g).cuda().requires_grad_(True)
    stop = (3 * torch.randn(B, T, generator=g)).cuda().requires_grad_(True)
    align = torch.softmax(torch.randn(B, T, L, generator=g), -1).cuda().requires_grad_(True)
    tgt = torch.randn(B, M, T, generator=g).cuda()
    st = (torch.rand(B, T, generator=g) > 0.8).float().cuda()
    tl = torch.tensor([21, 18, 15, 11, 9]); fl = torch.tensor([37, 30, 22, 37, 19])
    v = TacotronLossFn.apply(pre, post, stop, align, tgt, st, tl, fl, 0.25, True, 100.0)
    v[4].backward()
    got = [x.grad.clone() for x in (pre, post, stop, align)]
    for x in (pre, post, stop, align):
        x.grad = None
    ref_terms = [2 * F.mse_loss(pre, tgt), F.mse_loss(post, tgt),
                 F.binary_cross_entropy_with_logits(stop, st, pos_weight=torch.tensor([100.0]).cuda()) / (M + 2),
                 O.guided_attention(align.cpu(), tl, fl, 0.25).cuda()]
    sum(ref_terms).backward()
    for a, b in zip(v[:4], ref_terms):
        assert abs(a.item() - b.item()) <= 1e-5 * max(1.0, abs(b.item()))
    for a, x in zip(got, (pre, post, stop, align)):
        assert (a - x.grad).abs().max().item() <= 1e-6 + 1e-4 * x.grad.abs().max().item()
    # a post-net target that is not the decoder's target (TacotronLoss.forward takes both; the reference passes the same tensor)
    for x in (pre, post, stop, align):
        x.grad = None
    tgt2 = torch.randn(B, M, T, generator=g).cuda()
    v2 = TacotronLossFn.apply(pre, post, stop, align, tgt, st, tl, fl, 0.25, True, 100.0, tgt2)
    v2[4].backward()
    got_post = post.grad.clone()
    post.grad = None
    ref_pos = F.mse_loss(post, tgt2)
    ref_pos.backward()
    assert abs(v2[1].item() - ref_pos.item()) <= 1e-5 * max(1.0, abs(ref_pos.item())) and abs(v2[0].item() - ref_terms[0].item()) <= 1e-5
    assert (got_post - post.grad).abs().max().item() <= 1e-6 + 1e-4 * post.grad.abs().max().item()

    torch.manual_seed(0)
    ps = [torch.randn(300, 70).cuda(), torch.randn(5).cuda(), torch.randn(70000).cuda()]
    p1 = [torch.nn.Parameter(p.clone()) for p in ps]
    p2 = [torch.nn.Parameter(p.clone()) for p in ps]
    o1 = FusedAdam(p1, lr=1e-3, weight_decay=1e-6)
    o2 = torch.optim.Adam(p2, lr=1e-3, weight_decay=1e-6)
    for it in range(3):
        gs = [torch.randn_like(p) * (3.0 if it == 0 else 0.01) for p in ps]
        for a, b, gg in zip(p1, p2, gs):
            a.grad, b.grad = gg.clone(), gg.clone()
        norm = o1.step(max_norm=0.25)
        ref_norm = torch.nn.utils.clip_grad_norm_(p2, 0.25)
        o2.step()
        assert abs(norm[0].item() - ref_norm.item()) <= 1e-4 * ref_norm.item()
        for a, b in zip(p1, p2):
            assert (a - b).abs().max().item() <= 1e-6
    assert set(o1.state_dict()['state'][0].keys()) == set(o2.state_dict()['state'][0].keys())


@pytest.mark.parametrize('variant', ['nt', 'nn', 'tt', 'tn'])
def test_split_bf16_gemm_core_is_fp32_accurate(variant):
    """The default GEMM core evaluates fp32 products as six bf16 partial products (csrc/gemm.hip).  Its error against an
    fp64 product must stay at the level of an fp32 accumulation: the yardstick is torch.matmul in fp32 (rocBLAS, exact fp32
    products) on the same data, measured in units of sum|a_k b_k| per output element, on wide-dynamic-range data, ragged
    sizes and all four storage variants.  (An operand rounded to ONE bf16 would sit near 2^-9 = 2e-3 on this scale.)"""
    from multilingual_text_to_speech_amd import kernels as K
    torch.manual_seed(3)
    M, N, Kd = 333, 517, 1234          # none a multiple of the 128x128x32 tile
    dev = torch.device('cuda')
    # values spanning ~12 binades so that the low split planes matter
    A = (torch.randn(M, Kd, device=dev) * torch.exp2(torch.randint(-6, 6, (M, Kd), device=dev).float()))
    Bm = (torch.randn(N, Kd, device=dev) * torch.exp2(torch.randint(-6, 6, (N, Kd), device=dev).float()))
    C = torch.empty(M, N, device=dev)
    tA, tB = variant[0] == 't', variant[1] == 'n'
    a_st = A.t().contiguous() if tA else A          # transA: stored [K, M]
    b_st = Bm.t().contiguous() if tB else Bm        # transB: stored [K, N]
    K.gemm(a_st, b_st, C, M, N, Kd, M if tA else Kd, N if tB else Kd, N, transA=tA, transB=tB)
    ref = A.double() @ Bm.double().t()
    scale = A.double().abs() @ Bm.double().abs().t()
    err = ((C.double() - ref).abs() / scale).max().item()
    err_torch = (((A @ Bm.t()).double() - ref).abs() / scale).max().item()
    assert err <= 1.25 * err_torch + 2.0 ** -24, f'{variant}: split-bf16 GEMM error {err:.3e} (torch fp32 matmul: {err_torch:.3e})'


@pytest.mark.parametrize('name', golden_names('infer'))
def test_batched_inference_equals_batch1_per_utterance(name):
    """Tacotron.inference_batch (bucketed encoder / post-net, ONE batched decoder run with per-sample lengths and stop rule)
    must reproduce Tacotron.inference - the reference's batch-1 semantics, pinned by the *_infer fixtures - for every
    utterance of a ragged batch, given the same prenet dropout draws."""
    from multilingual_text_to_speech_amd.masks import provider
    from multilingual_text_to_speech_amd.params import Params as hp
    fx = load_golden(name)
    model = build_hip_model(fx)
    dev = torch.device('cuda')
    base = fx['text'][0].clone()
    L0 = base.numel()
    g = torch.Generator().manual_seed(11)
    texts = [base, base.flip(0).contiguous(), base[:max(3, L0 - 2)].clone()]
    n_utt = len(texts)
    n_lang = len(hp.languages) if hp.multi_language else 0
    langs = None
    if fx['languages'] is not None:
        fl = fx['languages']
        langs = []
        for i, t in enumerate(texts):
            if fl.dim() == 3:        # per-character weights: reuse the fixture's (rolled for variety), cut to length
                w = fl[0].roll(i, 0)[:t.numel()].clone()
            else:                    # a language id -> one-hot rows
                w = torch.zeros(t.numel(), n_lang); w[:, (int(fl.reshape(-1)[0]) + i) % n_lang] = 1.0
            langs.append(w)
    spks = None
    if fx['speakers'] is not None:
        spks = [(int(fx['speakers'].reshape(-1)[0]) + i) % max(1, hp.speaker_number) for i in range(n_utt)]
    Tmax, P = model._decoder._max_frames, hp.prenet_dimension
    n_pre = len(model._decoder._prenet._layers)
    draws = [(torch.rand(Tmax, n_utt, P, generator=g) >= model._decoder._prenet._dropout_rate).to(torch.uint8) for _ in range(n_pre)]
    try:
        provider.injected = {f'dec.prenet.{k}': draws[k].to(dev) for k in range(n_pre)}
        batch = model.inference_batch(texts, spks, langs)
        singles = []
        for i, t in enumerate(texts):
            provider.injected = {f'dec.prenet.{k}': draws[k][:, i:i + 1].contiguous().to(dev) for k in range(n_pre)}
            lw = langs[i].unsqueeze(0).to(dev) if langs is not None else None
            sp = torch.tensor([spks[i]], dtype=torch.int64, device=dev) if spks is not None else None
            singles.append(model.inference(t.clone().to(dev), sp, lw))
    finally:
        provider.injected = None
    for i in range(n_utt):
        assert batch[i].shape == singles[i].shape, (name, i, batch[i].shape, singles[i].shape)
        err = (batch[i] - singles[i]).abs().max().item()
        assert err <= 1e-3, f'{name} utterance {i}: max |delta| = {err:.3e}'


def _ddp_gpu_worker(rank, world, port, out, fixture):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      MTTS_PERSIST='0')       # both ranks share ONE GPU here: the persistent kernels need the whole chip for themselves
    import torch
    import torch.distributed as dist
    import bench
    from multilingual_text_to_speech_amd import dist as D
    from multilingual_text_to_speech_amd.optim import FusedAdam
    from multilingual_text_to_speech_amd.params import Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import TacotronLoss
    from tests.helpers import build_hip_model, load_golden
    D.init(backend='gloo')                      # both ranks share cuda:0; gloo stages device tensors through the host
    fx = load_golden(fixture)
    model = build_hip_model(fx).train()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.01)
    D.broadcast_parameters(model, 0)
    opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=hp.weight_decay)
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    buckets = D.GradientBuckets(model.parameters(), bucket_bytes=1 << 16, overlap=True)
    G = len(hp.languages) if hp.encoder_type in ('generated', 'convolutional') else 1
    losses = []
    for step in range(2):
        batch = bench.synthetic_batch(hp, 2 * G, 9, 12, torch.device('cuda'), seed=10 * step + rank)
        losses.append(float(bench.train_step(model, crit, opt, buckets, batch, hp)))
    torch.cuda.synchronize()
    torch.save(dict(losses=losses, params=[p.detach().cpu() for p in model.parameters()], nb=len(buckets.buckets)), f'{out}/g{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_train_steps_two_ranks_one_gpu(tmp_path):
    """Two processes (gloo, both on cuda:0) run two full train steps with overlapped bucket all-reduce and the fused
    clip+Adam: replicas must stay bit-identical although every rank sees different data."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_gpu_worker, args=(2, port, str(tmp_path), 'generated_train'), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'g0.pt'), torch.load(tmp_path / 'g1.pt')
    assert r0['nb'] > 1 and all(map(lambda v: v == v and abs(v) < 1e4, r0['losses'] + r1['losses']))
    assert r0['losses'] != r1['losses']                         # different shards
    for a, b in zip(r0['params'], r1['params']):
        assert torch.equal(a, b)


def _ddp_guard_worker(rank, world, port, out):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from multilingual_text_to_speech_amd import dist as D
    from multilingual_text_to_speech_amd.kernels import _err_flag
    from multilingual_text_to_speech_amd.optim import FusedAdam
    D.init(backend='gloo')                      # both ranks share cuda:0; gloo stages device tensors through the host
    torch.manual_seed(0)
    w = [torch.nn.Parameter(torch.randn(257, 33, device='cuda'))]
    opt = FusedAdam(w, lr=1e-2)
    g = torch.Generator().manual_seed(5)
    grads = [torch.randn(257, 33, generator=g).cuda() for _ in range(3)]      # identical on both ranks (as after an all-reduce)
    res = {}
    w[0].grad = grads[0]
    opt.step(max_norm=0.25)
    opt.poll_skipped()
    res['after0'] = w[0].detach().cpu().clone()
    if rank == 1:
        _err_flag('cuda')[1] = 2                # this rank's persistent decoder "gave up"
    w[0].grad = grads[1]
    opt.step(max_norm=0.25)
    opt.poll_skipped()
    torch.cuda.synchronize()
    res['after1'] = w[0].detach().cpu().clone()
    res['flag'] = _err_flag('cuda').tolist()
    _err_flag('cuda').zero_()
    w[0].grad = grads[2]
    opt.step(max_norm=0.25)
    torch.cuda.synchronize()
    res['after2'] = w[0].detach().cpu().clone()
    res['skipped'] = opt.poll_skipped(wait=True)
    torch.save(res, f'{out}/guard{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_guarded_adam_step_is_skipped_on_every_rank_when_one_rank_reports_an_error(tmp_path):
    """Two ranks (gloo, one GPU): rank 1's device error word is set before the second step.  FusedAdam.step MAX-all-reduces the guard
    words (dist.agree_on_guard), so BOTH ranks skip that update on the device and both see the word; the replicas stay identical
    through the skipped step and the next real one (VERDICT r5 weak 8 ii; reference semantics train.py:84-85,173-179)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_guard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'guard0.pt'), torch.load(tmp_path / 'guard1.pt')
    for k in ('after0', 'after1', 'after2'):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.equal(r0['after0'], r0['after1'])              # the step with the error word moved nothing, on either rank
    assert not torch.equal(r0['after1'], r0['after2'])
    assert r0['flag'][1] == 2 and r1['flag'][1] == 2            # every rank will raise at its next error poll
    assert r0['skipped'] == 1 and r1['skipped'] == 1


def _eval_collated(hp, G, n_batches=3):
    """Collate-format validation batches (8-tuples of data.Collate) with RAGGED text lengths, so that the ranks hold different
    numbers of valid characters."""
    import bench
    out = []
    for i in range(n_batches):
        b = bench.synthetic_batch(hp, 2 * G, 9, 12, torch.device('cpu'), seed=40 + i)
        tl = b['text_length'].clone(); tl[-1] = 5 + i
        b['text'][-1, int(tl[-1]):] = 0
        out.append((b['text'], tl, b['target'], None, b['target_length'], b['stop'], b['speakers'], b['languages']))
    return out


def _ddp_eval_worker(rank, world, port, out, fixture):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      MTTS_PERSIST='0')       # both ranks share ONE GPU
    import torch.distributed as dist
    import train
    from multilingual_text_to_speech_amd import dist as D
    from multilingual_text_to_speech_amd.params import Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import TacotronLoss
    from tests.helpers import build_hip_model, load_golden
    D.init(backend='gloo')
    model = build_hip_model(load_golden(fixture)).train()
    assert hp.reversal_classifier
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    res = train.evaluate(hp, _eval_collated(hp, len(hp.languages)), model, crit, torch.device('cuda'), rank, world)
    torch.save(res, f'{out}/ev{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluation_with_the_real_loss_and_an_odd_batch_count(tmp_path):
    """train.evaluate under data parallelism with the REAL TacotronLoss and hp.reversal_classifier: three validation batches over
    two ranks (2 + 1).  The classifier term must not issue a collective outside the sharded training step (the ranks make different
    numbers of loss calls: a collective there pairs with the final all-reduce of the other rank - hang / garbage), both ranks return
    the same means, and the classifier term (deterministic in eval mode: encoder + classifier only) equals the single-process value."""
    import socket
    import torch.multiprocessing as mp
    import train
    from multilingual_text_to_speech_amd.params import Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import TacotronLoss
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_eval_worker, args=(2, port, str(tmp_path), 'generated_train'), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'ev0.pt'), torch.load(tmp_path / 'ev1.pt')
    assert set(r0) == set(r1) == set(train.EVAL_TERMS) | set(train.EVAL_EXTRAS)
    for k in r0:
        assert r0[k] == r1[k] and r0[k] == r0[k], k
    model = build_hip_model(load_golden('generated_train')).train()
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    single = train.evaluate(hp, _eval_collated(hp, len(hp.languages)), model, crit, torch.device('cuda'))
    assert abs(single['lang_class'] - r0['lang_class']) <= 1e-5 * max(1.0, abs(single['lang_class']))


def _micro_batch_grads(model, crit, hp, rank, G):
    """Gradients of one micro-batch exactly as data-parallel rank `rank` computes them (same data seed, same dropout keys)."""
    import os
    import bench
    os.environ['RANK'] = str(rank)                    # MaskProvider mixes the rank into its Philox key
    torch.manual_seed(77)
    batch = bench.synthetic_batch(hp, 2 * G, 9, 12, torch.device('cuda'), seed=100 + rank)
    post, pre, stop, align, spk, enc = model(batch['text'], batch['text_length'], batch['target'], batch['target_length'],
                                             batch['speakers'], batch['languages'], 1.0)
    loss, _ = crit(batch['text_length'].cuda(), batch['target_length'].cuda(), pre, batch['target'], post, batch['target'], stop,
                   batch['stop'], align, batch['speakers'], spk, enc, None)
    loss.backward()
    return float(loss)


def _ddp_grad_worker(rank, world, port, out, fixture):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      MTTS_PERSIST='0')       # both ranks share ONE GPU here: the persistent kernels need the whole chip for themselves
    import torch.distributed as dist
    from multilingual_text_to_speech_amd import dist as D
    from multilingual_text_to_speech_amd.params import Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import TacotronLoss
    from tests.helpers import build_hip_model, load_golden
    D.init(backend='gloo')
    model = build_hip_model(load_golden(fixture)).train()
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    buckets = D.GradientBuckets(model.parameters(), bucket_bytes=1 << 16, overlap=True)
    G = len(hp.languages) if hp.encoder_type in ('generated', 'convolutional') else 1
    loss = _micro_batch_grads(model, crit, hp, rank, G)
    buckets.all_reduce()
    torch.cuda.synchronize()
    torch.save(dict(loss=loss, grads={k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}), f'{out}/dp{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_the_mean_over_micro_batches(tmp_path, monkeypatch):
    """the all-reduced gradient of a 2-rank step must equal what ONE rank gets from the same two
    micro-batches run one after the other and averaged (the reference's loss is a mean over the gathered global batch;
    equal per-rank B and T make the mean of local means that mean).  Whole Tacotron incl. the adversarial classifier."""
    import socket
    import torch.multiprocessing as mp
    from multilingual_text_to_speech_amd.params import Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import TacotronLoss
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_grad_worker, args=(2, port, str(tmp_path), 'generated_train'), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'dp0.pt'), torch.load(tmp_path / 'dp1.pt')
    for k in r0['grads']:
        assert torch.equal(r0['grads'][k], r1['grads'][k]), k            # both ranks hold the same reduced gradient
    monkeypatch.setenv('RANK', '0')
    model = build_hip_model(load_golden('generated_train')).train()
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    G = len(hp.languages)
    acc, losses = None, []
    for k in range(2):
        model.zero_grad(set_to_none=True)
        losses.append(_micro_batch_grads(model, crit, hp, k, G))
        g = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    assert abs(losses[0] - r0['loss']) <= 1e-6 * max(1, abs(losses[0])) and abs(losses[1] - r1['loss']) <= 1e-6 * max(1, abs(losses[1]))
    assert losses[0] != losses[1]
    for n, ref in acc.items():
        ref = ref / 2
        err = (r0['grads'][n] - ref).abs().max().item()
        assert err <= 1e-5 * ref.abs().max().item() + 1e-7, f'{n}: {err:.3e}'


def test_exact_f32_gemm_core_selectable():
    """MTTS_GEMM_EXACT_F32=1 (read once per process) switches every GEMM to the v_mfma_f32_32x32x2_f32 core; the fixture
    parity of a forward and a backward case must hold on it as well."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MTTS_GEMM_EXACT_F32='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', 'tests/test_gpu_forward.py', 'tests/test_gpu_backward.py',
                        '-k', 'shared_train or simple_train'], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


def test_philox_keep_masks_have_the_right_rate_and_do_not_repeat():
    """mtts_dropout_keep_mask: P(keep) = 1 - p within sampling error, consecutive calls draw fresh bits, reseeding replays."""
    from multilingual_text_to_speech_amd.masks import MaskProvider
    pr = MaskProvider()
    torch.manual_seed(123)
    for p in (0.1, 0.5):
        m = pr.keep('x', (257, 1031), p, 'cuda')          # odd size: exercises the unaligned tail
        assert m.dtype == torch.uint8 and m.shape == (257, 1031) and int(m.max()) == 1
        rate = m.float().mean().item()
        assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / m.numel()) ** 0.5 + 1e-4, (p, rate)
    a = pr.keep('x', (4096,), 0.5, 'cuda')
    b = pr.keep('x', (4096,), 0.5, 'cuda')
    assert 0.4 < (a == b).float().mean().item() < 0.6     # independent draws agree about half the time
    torch.manual_seed(123)
    pr2 = MaskProvider()
    pr2.keep('x', (257, 1031), 0.1, 'cuda'); pr2.keep('x', (257, 1031), 0.5, 'cuda')     # same sequence of calls after reseeding
    assert torch.equal(pr2.keep('x', (4096,), 0.5, 'cuda'), a)
    assert pr.keep('x', (8,), 0.0, 'cuda') is None


@pytest.mark.parametrize('seed', range(10))
def test_conv1d_kernels_match_torch_on_random_shapes(seed):
    """Implicit-GEMM convolution (forward, input gradient, weight gradient) against torch.nn.functional.conv1d in fp64 for random
    channel counts / kernel sizes / dilations / groups / lengths, including lengths shorter than the receptive field."""
    from multilingual_text_to_speech_amd import kernels as K
    g = torch.Generator().manual_seed(100 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    groups = [1, 1, 2, 5][ri(0, 3)]
    Cg, Og = 4 * ri(1, 40), 4 * ri(1, 40)
    k = [1, 3, 5, 31, 2, 4][ri(0, 5)]
    dil = [1, 1, 3, 9][ri(0, 3)] if k % 2 else 1          # even kernels pad (p, p+1) like modules/layers.py:72-73
    N_, L = ri(1, 5), ri(1, 70)
    Cin, O = Cg * groups, Og * groups
    x = torch.randn(N_, L, Cin, generator=g).cuda()
    w = (torch.randn(O, Cg, k, generator=g) / (Cg * k) ** 0.5).cuda()
    dy = torch.randn(N_, L, O, generator=g).cuda()
    wp = K.pack_conv_weight(w)
    y = K.conv1d_fwd(x, wp, k, dil, groups)
    dx, dwp = K.conv1d_bwd(x, wp, dy, k, dil, groups)
    dw = K.unpack_conv_weight(dwp, O, Cg, k)
    xr = x.double().transpose(1, 2).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    pl = (k - 1) * dil // 2
    yr = torch.nn.functional.conv1d(torch.nn.functional.pad(xr, (pl, (k - 1) * dil - pl)), wr, dilation=dil, groups=groups)
    yr.backward(dy.double().transpose(1, 2))
    tol = lambda ref: 2e-5 * max(1.0, ref.abs().max().item())
    tag = f'groups={groups} Cg={Cg} Og={Og} k={k} dil={dil} N={N_} L={L}'
    assert (y.double() - yr.transpose(1, 2)).abs().max().item() <= tol(yr), tag
    assert (dx.double() - xr.grad.transpose(1, 2)).abs().max().item() <= tol(xr.grad), tag
    assert (dw.double() - wr.grad).abs().max().item() <= tol(wr.grad) * (N_ * L) ** 0.5, tag


@pytest.mark.parametrize('seed', range(6))
def test_skinny_gemm_matches_torch_on_random_shapes(seed):
    """mtts_skinny_gemm (the per-step GEMM of the decoder loop) as a plain multi-segment product out = sum_s x_s W_s^T + bias:
    random batch sizes (all three row-tile variants, ragged last tile), 1-3 K segments, K split on / off."""
    import ctypes
    from multilingual_text_to_speech_amd import _C
    from multilingual_text_to_speech_amd._C import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(200 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    B, N, nseg = [1, 7, 16, 23, 40, 64, 100][ri(0, 6)], 4 * ri(1, 80), ri(1, 3)
    ks = [1, 1, 4][ri(0, 2)]
    xs = [torch.randn(B, 4 * ri(1, 90), generator=g).cuda() for _ in range(nseg)]
    ws = [(torch.randn(N, x.shape[1], generator=g) / x.shape[1] ** 0.5).cuda() for x in xs]
    bias = torch.randn(N, generator=g).cuda()
    a = _C.SkinnyArgs()
    a.nseg, a.B, a.N, a.ksplit = nseg, B, N, ks
    for i, (x, w) in enumerate(zip(xs, ws)):
        a.seg[i].x, a.seg[i].w, a.seg[i].K, a.seg[i].ldx, a.seg[i].ldw = ptr(x), ptr(w), x.shape[1], x.shape[1], x.shape[1]
    out = torch.zeros(ks, B, N, device='cuda')
    a.out, a.ldo = ptr(out), N
    if ks > 1:
        a.out_ks = B * N                       # raw partial slabs, summed by the consumer
    else:
        a.bias = ptr(bias)
    check(lib().mtts_skinny_gemm(ctypes.byref(a), stream_ptr()), 'mtts_skinny_gemm')
    ref = sum(x.double() @ w.double().t() for x, w in zip(xs, ws))
    got = out.double().sum(0) if ks > 1 else out[0].double() - bias.double()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), (B, N, [x.shape[1] for x in xs], ks)


@pytest.mark.parametrize('B,N,Ks,ldo', [(128, 81, (1024, 288), 84), (240, 81, (1024, 544), 84), (1, 81, (1024, 288), 84), (37, 20, (4096,), 20),
                                         (128, 256, (80,), 256)])
def test_skinny_gemm_one_round_trip_variant(B, N, Ks, ldo):
    """skinny_kernel_wide (16-row workgroups, every K chunk of a wave requested before the first product): the frame / stop
    projection of the free-running loop (reference modules/tacotron2.py:191-193; M + 1 = 81 columns into rows of 84) at synthesis
    batch sizes, a K long enough to loop (4096 > 16 chunks x 8 waves x 16), and a prenet-layer shape; bias fused, row stride > N."""
    import ctypes
    from multilingual_text_to_speech_amd import _C
    from multilingual_text_to_speech_amd._C import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(77)
    xs = [torch.randn(B, K, generator=g).cuda() for K in Ks]
    Kt = sum(Ks)
    w = (torch.randn(N, Kt, generator=g) / Kt ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    a = _C.SkinnyArgs()
    a.nseg, a.B, a.N, a.ksplit = len(Ks), B, N, 1
    k0 = 0
    for i, x in enumerate(xs):
        a.seg[i].x, a.seg[i].w, a.seg[i].K, a.seg[i].ldx, a.seg[i].ldw = ptr(x), w.data_ptr() + 4 * k0, x.shape[1], x.shape[1], Kt
        k0 += x.shape[1]
    out = torch.full((B, ldo), 7.0, device='cuda')
    a.out, a.ldo, a.bias = ptr(out), ldo, ptr(bias)
    check(lib().mtts_skinny_gemm(ctypes.byref(a), stream_ptr()), 'mtts_skinny_gemm')
    ref = torch.cat(xs, 1).double() @ w.double().t() + bias.double()
    assert (out[:, :N].double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert (out[:, N:] == 7.0).all()                      # padding columns of the row are not touched


def test_fused_adam_two_groups_mixed_steps_match_torch():
    """hp.encoder_optimizer layout (reference train.py:261-270): two parameter groups with their own learning rates, one global
    clip coefficient over ALL parameters, and a parameter that receives no gradient in the first step (its bias-correction step
    lags by one: the update then runs as one launch per (group, step value) table).  Against torch.optim.Adam + clip_grad_norm_."""
    from multilingual_text_to_speech_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(300, 70), (5,), (70000,), (64, 33), (17,)]
    base = [torch.randn(*s).cuda() for s in shapes]
    p1 = [torch.nn.Parameter(p.clone()) for p in base]
    p2 = [torch.nn.Parameter(p.clone()) for p in base]
    groups = lambda ps: [{'params': ps[:3]}, {'params': ps[3:], 'lr': 3e-4}]
    o1 = FusedAdam(groups(p1), lr=1e-3, weight_decay=1e-6)
    o2 = torch.optim.Adam(groups(p2), lr=1e-3, weight_decay=1e-6)
    for it in range(4):
        gs = [torch.randn_like(p) * (3.0 if it % 2 == 0 else 0.01) for p in base]
        for k, (a, b, gg) in enumerate(zip(p1, p2, gs)):
            skip = it == 0 and k in (1, 4)                      # late starters in BOTH groups
            a.grad, b.grad = (None, None) if skip else (gg.clone(), gg.clone())
        norm = o1.step(max_norm=0.25)
        ref_norm = torch.nn.utils.clip_grad_norm_([p for p in p2 if p.grad is not None], 0.25)
        o2.step()
        assert abs(norm[0].item() - ref_norm.item()) <= 1e-4 * ref_norm.item()
        for k, (a, b) in enumerate(zip(p1, p2)):
            assert (a - b).abs().max().item() <= 2e-6, (it, k)
    for a, b in zip(p1, p2):
        assert float(o1.state[a]['step']) == float(o2.state[b]['step'])
        assert (o1.state[a]['exp_avg_sq'] - o2.state[b]['exp_avg_sq']).abs().max().item() <= 1e-7


def test_fused_adam_skips_a_non_finite_step_and_reports_it():
    """Guarded optimizer step (csrc/loss_optim.hip, AdamArgs.guard): a non-finite gradient norm leaves weights and moments untouched on
    the device; FusedAdam.poll_skipped() surfaces it one poll later without stalling the stream (round 5) - the reference's
    clip_grad_norm_ + Adam (train.py:84-85) would have written NaNs into every weight."""
    import warnings
    from multilingual_text_to_speech_amd.optim import FusedAdam
    torch.manual_seed(0)
    w = [torch.nn.Parameter(torch.randn(300, 17, device='cuda')), torch.nn.Parameter(torch.randn(1000, device='cuda'))]
    opt = FusedAdam(w, lr=1e-2)
    for p in w:
        p.grad = torch.randn_like(p)
    opt.step(max_norm=0.25)
    assert opt.poll_skipped() == 0
    before = [p.detach().clone() for p in w]
    w[1].grad[123] = float('inf')
    opt.step(max_norm=0.25)
    torch.cuda.synchronize()
    for p, b in zip(w, before):
        assert torch.equal(p.detach(), b)                     # the update was skipped on the device
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        opt.poll_skipped()                                     # enqueues the copy of [norm, coefficient] of the skipped step
        torch.cuda.synchronize()
        n = opt.poll_skipped()                                 # accounts for it
    assert n == 1 and any('skipped' in str(c.message) for c in caught)
    for p in w:
        p.grad = torch.randn_like(p)
    opt.step(max_norm=0.25)
    torch.cuda.synchronize()
    assert not torch.equal(w[0].detach(), before[0])          # a finite step updates again


def test_fused_adam_counts_every_skipped_step_when_the_host_runs_ahead():
    """The host queues several steps before the GPU has finished the first (normal asynchronous training): every step's [norm,
    coefficient] is copied into its own pinned slot between that step's kernels and the next step's, so a skipped step in the middle
    is counted although later steps have overwritten the device-side pair (the one-slot poll lost it)."""
    import warnings
    from multilingual_text_to_speech_amd.optim import FusedAdam
    torch.manual_seed(0)
    w = [torch.nn.Parameter(torch.randn(2048, 2048, device='cuda'))]
    opt = FusedAdam(w, lr=1e-3)
    grads = [torch.randn_like(w[0]) for _ in range(6)]
    grads[1][5, 5] = float('nan')
    grads[4][7, 7] = float('inf')
    big = torch.randn(8192, 8192, device='cuda')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for g in grads:
            for _ in range(3):
                big = torch.tanh(big)               # keep the stream busy: the host is several steps ahead by the end of the loop
            w[0].grad = g
            opt.step(max_norm=0.25)
            opt.poll_skipped()
        assert opt.poll_skipped(wait=True) == 2
    assert opt.poll_skipped() == 2                  # a poll without a new step only accounts


def test_to_device_async_copies_values_without_synchronising():
    """kernels.to_device_async: small host tensors through the pinned ring - values, dtype conversion, shapes (0-d, empty, 2-d), more copies
    than the ring has slots, a tensor too large for a slot (falls back to the ordinary copy), device tensors passed through - and no
    synchronising call while doing so (torch's sync-debug mode raises on one)."""
    from multilingual_text_to_speech_amd import kernels as K
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(0)
    cases = [torch.randint(0, 1000, (64,), generator=g), torch.randint(0, 1000, (7, 3), generator=g), torch.tensor(5), torch.randn(33, generator=g),
             torch.zeros(0, dtype=torch.int64), torch.randint(0, 2, (100,), generator=g).bool()]
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        outs = [K.to_device_async(t, dev) for t in cases if t.numel()]
        many = [K.to_device_async(torch.full((5,), i), dev, torch.int32) for i in range(3 * K._PinnedRing.SLOTS)]
        conv = K.to_device_async(cases[0], dev, torch.int32)
        same = K.to_device_async(outs[0], dev)
    finally:
        torch.cuda.set_sync_debug_mode('default')
    for t, o in zip([t for t in cases if t.numel()], outs):
        assert o.device.type == 'cuda' and o.shape == t.shape and o.dtype == t.dtype and torch.equal(o.cpu(), t)
    assert all(int(m[0]) == i and m.dtype == torch.int32 for i, m in enumerate(many))
    assert conv.dtype == torch.int32 and torch.equal(conv.cpu().long(), cases[0]) and same.data_ptr() == outs[0].data_ptr()
    assert K.to_device_async(cases[4], dev).numel() == 0 and K.to_device_async(None, dev) is None
    big = torch.arange(1 << 15, dtype=torch.int64)                   # 256 KB: larger than a slot
    assert torch.equal(K.to_device_async(big, dev).cpu(), big)


def test_train_step_makes_no_synchronising_call():
    """A whole train step (forward, loss, backward, fused clip + Adam, the two non-blocking polls) without ONE host-side stream
    synchronisation: round 5 made nine per step (sequence lengths moved with cpu_tensor.to(device), the optimizer's tables rebuilt whenever
    zero_grad(set_to_none=True) had re-allocated the gradients) and each one cost the host its run-ahead - 67.6 -> 66.2 ms per step
    (profiles/r06_host_sync_ab.txt).  torch's sync-debug mode raises on any synchronising torch call."""
    import bench
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
    from multilingual_text_to_speech_amd.optim import FusedAdam
    for preset, B in (('shared_training', 8), ('generated_switching', 10)):
        presets.apply(preset, speaker_number=7)
        torch.manual_seed(0)
        dev = torch.device('cuda')
        model = Tacotron().to(dev).train()
        crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
        opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
        batch = bench.synthetic_batch(hp, B, 30, 40, dev)
        for _ in range(2):                                            # first-touch allocations, table builds
            bench.train_step(model, crit, opt, None, batch, hp)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode('error')
        try:
            for _ in range(2):
                bench.train_step(model, crit, opt, None, batch, hp)
        finally:
            torch.cuda.set_sync_debug_mode('default')
        torch.cuda.synchronize()
