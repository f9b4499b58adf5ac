"""Generate golden fixtures by EXECUTING THE REFERENCE (imported from /root/reference) on CPU.

    python oracle/make_golden.py [--out tests/golden] [--reference /root/reference]

Test infrastructure only.  Runs in this container (the reference cannot travel to the GPU box);
the produced `tests/golden/*.pt` files are committed.  Each fixture stores: the hyper-parameter
overrides, the reference model's `state_dict`, the inputs, every dropout multiplier the reference
drew (named by call site), the teacher-forcing draw, the six forward outputs, the updated BatchNorm
running statistics, the reference `TacotronLoss` value and the gradient of that loss w.r.t. every
parameter.  Nothing from this repo's product package is imported here (the reference and the
product share the module names `modules`, `params`, `utils`).

Two more modes (round 6):
  * `--real DIR`: the reference at its REAL layer widths (its own `params/*.json`, attention kernel 31, A = 128, H = 1024) on small
    batches - 100+ MB each, written to DIR and NOT committed; `tests/test_oracle_golden.py::test_oracle_matches_the_reference_at_real_widths`
    runs this mode into a temp directory and replays the result through the oracle (skipped where /root/reference is absent).
  * trajectory cases (`*_3step_train`, committed): THREE consecutive training steps of the reference's loop (train.py:58-93:
    forward, TacotronLoss, backward, clip_grad_norm_(0.25), Adam(lr 1e-3, weight_decay 1e-6).step(), criterion.update_states()), a
    fresh batch and fresh dropout draws per step; per step: inputs, draws, loss, the clip's gradient norm, every parameter, the
    BatchNorm running statistics and both Adam moments AFTER the step, and the guided-attention state.

Dropout control (SURVEY.md §8c recipe 2): `torch.nn.functional.dropout` is replaced by a recorder
that draws its own Bernoulli mask from a seeded generator and stores the multiplier; `torch.rand`
is wrapped to capture the `teacher` draw of modules/tacotron2.py:171.
"""
import argparse
import os
import sys

import torch

SMALL = dict(embedding_dimension=16, encoder_dimension=16, prenet_dimension=16, attention_dimension=8,
             attention_kernel_size=5, attention_location_dimension=4, decoder_dimension=32,
             postnet_dimension=16, num_mels=8, characters="abcdefghij ", punctuations_out=".,", punctuations_in="-",
             reversal_classifier_dim=8, speaker_embedding_dimension=8, stop_frames=2)

CASES = {
    # name: (hp overrides, B, L, T, flags)
    'simple_train': (dict(), 3, 9, 7, dict()),
    'simple_mixed_tf': (dict(), 3, 9, 7, dict(tf=0.5)),
    'simple_zoneout': (dict(decoder_regularization='zoneout'), 2, 8, 6, dict()),
    'simple_eval': (dict(), 3, 9, 7, dict(train=False)),
    'multi_simple_train': (dict(multi_language=True, language_number=3, languages=['a', 'b', 'c'],
                                language_embedding_dimension=4), 4, 10, 6, dict()),
    'shared_train': (dict(encoder_type='shared', multi_language=True, language_number=3,
                          languages=['a', 'b', 'c'], language_embedding_dimension=4, input_language_embedding=4),
                     3, 8, 5, dict(onehot_lang=True)),
    'separate_train': (dict(encoder_type='separate', multi_language=True, language_number=2,
                            languages=['a', 'b'], language_embedding_dimension=4),
                       1, 8, 5, dict(onehot_lang=True, mixed_lang=True)),   # the reference's MultiEncoder only broadcasts at B=1
    'generated_train': (dict(encoder_type='generated', multi_language=True, language_number=2, languages=['a', 'b'],
                             language_embedding_dimension=0, generator_dim=6, generator_bottleneck_dim=3,
                             multi_speaker=True, speaker_number=5, reversal_classifier=True,
                             reversal_classifier_w=0.125), 4, 12, 6, dict()),
    'convolutional_train': (dict(encoder_type='convolutional', multi_language=True, language_number=2,
                                 languages=['a', 'b'], language_embedding_dimension=4), 4, 12, 6, dict()),
    'generated_infer': (dict(encoder_type='generated', multi_language=True, language_number=2, languages=['a', 'b'],
                             language_embedding_dimension=0, generator_dim=6, generator_bottleneck_dim=3,
                             multi_speaker=True, speaker_number=5, max_output_length=12),
                        1, 10, 0, dict(infer=True, train=False)),
    'simple_infer': (dict(max_output_length=12), 1, 10, 0, dict(infer=True, train=False)),
    # T >= 100: the product's decoder schedules work in chunks of 48 steps (csrc/decoder.hip); these two cross 2 chunk
    # boundaries with a ragged last chunk, so the reference itself pins chunk hand-off and per-chunk gradient accumulation
    'simple_long_train': (dict(), 3, 12, 110, dict()),
    'generated_long_train': (dict(encoder_type='generated', multi_language=True, language_number=2, languages=['a', 'b'],
                                  language_embedding_dimension=0, generator_dim=6, generator_bottleneck_dim=3,
                                  multi_speaker=True, speaker_number=5, reversal_classifier=True,
                                  reversal_classifier_w=0.125), 4, 12, 100, dict()),
}


_GEN = dict(encoder_type='generated', multi_language=True, language_number=2, languages=['a', 'b'],
            language_embedding_dimension=0, generator_dim=6, generator_bottleneck_dim=3,
            multi_speaker=True, speaker_number=5, reversal_classifier=True, reversal_classifier_w=0.125)

# name: (hp overrides, B, L, T, number of optimizer steps)
TRAJECTORIES = {
    'simple_3step_train': (dict(), 3, 9, 7, 3),
    'generated_3step_train': (_GEN, 4, 12, 6, 3),
}

# name: (reference json under params/, extra overrides, B, L, T)   -- real widths; B a multiple of the language count for the grouped encoder
REAL_CASES = {
    'real_shared_training': ('shared_training.json', dict(), 2, 40, 12),
    'real_generated_switching': ('generated_switching.json', dict(multi_speaker=True, speaker_number=7), 5, 30, 8),
}


class Recorder:
    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.dropouts = []
        self.rands = []

    def dropout(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        keep = (torch.empty(input.shape).uniform_(generator=self.gen) >= p).float() / (1.0 - p)
        self.dropouts.append(keep)
        return input * keep

    def rand(self, *size, **kw):
        kw.pop('device', None)
        r = self._rand(*size, generator=self.gen, **kw)
        self.rands.append(r)
        return r


def name_masks(rec, hp, T, teacher, train, infer, n_steps=None):
    """Turn the recorder's call-ordered list into named tensors (see the call order in
    reference modules/tacotron2.py:355-385 and :148-209)."""
    q = list(rec.dropouts)
    masks = {}

    def take():
        return q.pop(0)

    if train:
        n_enc = {'simple': hp.encoder_blocks, 'shared': hp.encoder_blocks, 'generated': 14, 'convolutional': 14}
        if hp.encoder_type == 'separate':
            for l in range(hp.language_number):
                for i in range(hp.encoder_blocks):
                    masks[f'enc{l}.{i}'] = take()
        else:
            for i in range(n_enc[hp.encoder_type]):
                masks[f'enc.{i}'] = take()
    if not infer:
        for i in range(hp.prenet_layers):
            masks[f'prenet.{i}'] = take()
    steps = T if not infer else n_steps
    per = {}
    for i in range(steps):
        if infer or not bool(teacher[i]):
            for l in range(hp.prenet_layers):
                per.setdefault(f'prenet_step.{l}', {})[i] = take()
        if train:
            names = ['att_lstm', 'gen_lstm']
            if hp.decoder_regularization == 'zoneout':
                names = ['att_lstm.h', 'att_lstm.c', 'gen_lstm.h', 'gen_lstm.c']
            for n in names:
                per.setdefault(n, {})[i] = take()
    for n, d in per.items():
        shape = next(iter(d.values())).shape
        masks[n] = torch.stack([d.get(i, torch.ones(shape)) for i in range(steps)], 0)
    if train:
        for i in range(hp.postnet_blocks):
            masks[f'post.{i}'] = take()
    assert not q, f'{len(q)} unassigned dropout draws'
    return masks


def run_case(name, overrides, B, L, T, flags, ref_mods):
    hp, Tacotron, TacotronLoss, defaults = ref_mods
    hp.load_state_dict(defaults)
    for k in list(hp.state_dict()):
        if k not in defaults:
            delattr(hp, k)
    hp.load_state_dict(SMALL)
    hp.load_state_dict(overrides)
    train = flags.get('train', True)
    infer = flags.get('infer', False)
    tf = flags.get('tf', 1.0)

    torch.manual_seed(list(CASES).index(name) + 1)
    model = Tacotron()
    # non-trivial BN affine + running stats so that eval-mode parity is meaningful
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('running_var'):
                v.copy_(torch.empty(v.shape).uniform_(0.5, 1.5, generator=g))
            elif k.endswith('running_mean'):
                v.copy_(torch.empty(v.shape).uniform_(-0.3, 0.3, generator=g))
            elif k.endswith(('_block.2.weight',)):
                v.copy_(torch.empty(v.shape).uniform_(0.6, 1.4, generator=g))
            elif k.endswith(('_block.2.bias',)):
                v.copy_(torch.empty(v.shape).uniform_(-0.2, 0.2, generator=g))
    model.train(train)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    V = hp.symbols_count() + 3
    text = torch.randint(3, V, (B, L), generator=g)
    text_length = torch.sort(torch.randint(max(L // 2, 1), L + 1, (B,), generator=g), descending=True).values
    text_length[0] = L
    for b in range(B):
        text[b, text_length[b]:] = 0
    NL = hp.language_number if hp.multi_language else 0
    speakers = torch.randint(0, hp.speaker_number, (B,), generator=g) if hp.multi_speaker else None
    languages = None
    if hp.multi_language:
        languages = torch.arange(B) % NL
        if flags.get('onehot_lang'):
            languages = torch.nn.functional.one_hot(languages.unsqueeze(1).expand(-1, L), NL).float()
            if flags.get('mixed_lang'):
                languages[:, L // 2:, :] = 0.0
                languages[:, L // 2:, 1] = 1.0
    fx = dict(name=name, hp={**SMALL, **overrides}, train=train, state_dict=sd0, text=text,
              text_length=text_length, speakers=speakers, languages=languages)

    rec = Recorder(seed=1234)
    rec._rand = torch.rand
    real_dropout, real_rand = torch.nn.functional.dropout, torch.rand
    torch.nn.functional.dropout = rec.dropout
    torch.rand = rec.rand
    try:
        if infer:
            lang_w = None
            if hp.multi_language:
                lang_w = torch.zeros(1, L, NL)
                lang_w[0, :L // 2, 0] = 1.0            # code-switch in the middle of the utterance
                lang_w[0, L // 2:, 1] = 1.0
                lang_w[0, L // 2, :] = torch.tensor([0.25, 0.75])
            spk = speakers[:1] if speakers is not None else None
            with torch.no_grad():
                out = model.inference(text[0, :].clone(), spk, lang_w)
            fx.update(languages=lang_w, speakers=spk, text=text[:1], inference_output=out.clone())
            n_steps = out.shape[1]
            # free-running frames beyond the returned ones may have been computed; count prenet draws
            fx['masks'] = name_masks(rec, hp, 0, None, False, True, n_steps=len(rec.dropouts) // hp.prenet_layers)
            fx['n_frames'] = n_steps
        else:
            target_length = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
            target_length[0] = T
            target = torch.randn(B, hp.num_mels, T, generator=g)
            stop_target = torch.zeros(B, T)
            for b in range(B):
                target[b, :, target_length[b]:] = 0
                stop_target[b, max(int(target_length[b]) - hp.stop_frames, 0):] = 1.0
            outs = model(text, text_length, target, target_length, speakers, languages, tf)
            teacher = (rec.rands[0] > (1 - tf))
            fx.update(target=target, target_length=target_length, stop_target=stop_target, teacher=teacher,
                      masks=name_masks(rec, hp, T, teacher, train, False))
            post, pre, stop, align, spk_pred, enc = outs
            crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
            cls = model._reversal_classifier if hp.reversal_classifier else None
            loss, parts = crit(text_length, target_length, pre, target, post, target, stop, stop_target, align,
                               speakers, spk_pred, enc, cls)
            fx.update(post=post.detach().clone(), pre=pre.detach().clone(), stop=stop.detach().clone(),
                      alignment=align.detach().clone(), encoder_output=enc.detach().clone(),
                      speaker_prediction=None if spk_pred is None else spk_pred.detach().clone(),
                      loss=loss.detach().clone(), loss_parts={k: float(v) for k, v in parts.items()},
                      guided_g=hp.guided_attention_toleration)
            if train:
                loss.backward()
                fx['grads'] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
                sd1 = model.state_dict()
                fx['bn_stats'] = {k: sd1[k].detach().clone() for k in sd1
                                  if k.endswith(('running_mean', 'running_var'))}
    finally:
        torch.nn.functional.dropout, torch.rand = real_dropout, real_rand
    return fx


def _reset_hp(hp, defaults, base, overrides):
    hp.load_state_dict(defaults)
    for k in list(hp.state_dict()):
        if k not in defaults:
            delattr(hp, k)
    hp.load_state_dict(base)
    hp.load_state_dict(overrides)


def _randomize_bn(model, g):
    """Non-trivial BatchNorm affine parameters and running statistics (same recipe as run_case)."""
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('running_var'):
                v.copy_(torch.empty(v.shape).uniform_(0.5, 1.5, generator=g))
            elif k.endswith('running_mean'):
                v.copy_(torch.empty(v.shape).uniform_(-0.3, 0.3, generator=g))
            elif k.endswith(('_block.2.weight',)):
                v.copy_(torch.empty(v.shape).uniform_(0.6, 1.4, generator=g))
            elif k.endswith(('_block.2.bias',)):
                v.copy_(torch.empty(v.shape).uniform_(-0.2, 0.2, generator=g))


def _train_batch(hp, B, L, T, g):
    """One synthetic training batch (ragged, lengths sorted descending as the reference's packed BiLSTM requires)."""
    V = hp.symbols_count() + 3
    text = torch.randint(3, V, (B, L), generator=g)
    text_length = torch.sort(torch.randint(max(L // 2, 1), L + 1, (B,), generator=g), descending=True).values
    text_length[0] = L
    for b in range(B):
        text[b, text_length[b]:] = 0
    NL = hp.language_number if hp.multi_language else 0
    speakers = torch.randint(0, hp.speaker_number, (B,), generator=g) if hp.multi_speaker else None
    languages = torch.arange(B) % NL if hp.multi_language else None
    target_length = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
    target_length[0] = T
    target = torch.randn(B, hp.num_mels, T, generator=g)
    stop_target = torch.zeros(B, T)
    for b in range(B):
        target[b, :, target_length[b]:] = 0
        stop_target[b, max(int(target_length[b]) - hp.stop_frames, 0):] = 1.0
    return dict(text=text, text_length=text_length, speakers=speakers, languages=languages, target=target,
                target_length=target_length, stop_target=stop_target)


def _recorded_step(model, crit, hp, batch, T, seed):
    """Forward + loss of the reference with every dropout draw and the teacher draw recorded; returns (loss, record)."""
    rec = Recorder(seed=seed)
    rec._rand = torch.rand
    real_dropout, real_rand = torch.nn.functional.dropout, torch.rand
    torch.nn.functional.dropout = rec.dropout
    torch.rand = rec.rand
    try:
        outs = model(batch['text'], batch['text_length'], batch['target'], batch['target_length'], batch['speakers'],
                     batch['languages'], 1.0)
        teacher = (rec.rands[0] > 0.0)
        post, pre, stop, align, spk_pred, enc = outs
        cls = model._reversal_classifier if hp.reversal_classifier else None
        loss, parts = crit(batch['text_length'], batch['target_length'], pre, batch['target'], post, batch['target'], stop,
                           batch['stop_target'], align, batch['speakers'], spk_pred, enc, cls)
    finally:
        torch.nn.functional.dropout, torch.rand = real_dropout, real_rand
    r = dict(batch)
    r.update(teacher=teacher, masks=name_masks(rec, hp, T, teacher, True, False),
             post=post.detach().clone(), pre=pre.detach().clone(), stop=stop.detach().clone(), alignment=align.detach().clone(),
             encoder_output=enc.detach().clone(), speaker_prediction=None if spk_pred is None else spk_pred.detach().clone(),
             loss=loss.detach().clone(), loss_parts={k: float(v) for k, v in parts.items()})
    return loss, r


def run_real_case(name, json_name, overrides, B, L, T, ref_mods, reference_root):
    """The reference at the layer widths of one of its own params/*.json (SURVEY 8d config mapping): one train-mode step, outputs,
    loss and every parameter gradient.  Large (the state dict alone is 100+ MB): written outside the repository."""
    import json
    hp, Tacotron, TacotronLoss, defaults = ref_mods
    with open(os.path.join(reference_root, 'params', json_name), encoding='utf-8') as f:
        from_json = json.load(f)
    _reset_hp(hp, defaults, from_json, overrides)
    if hp.multi_language:
        hp.language_number = len(hp.languages)               # reference train.py:240
    torch.manual_seed(100 + list(REAL_CASES).index(name))
    model = Tacotron()
    g = torch.Generator().manual_seed(7)
    _randomize_bn(model, g)
    model.train(True)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = _train_batch(hp, B, L, T, g)
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    loss, fx = _recorded_step(model, crit, hp, batch, T, seed=1234)
    loss.backward()
    hp_over = dict(from_json)
    hp_over.update(overrides)
    if hp.multi_language:
        hp_over['language_number'] = hp.language_number
    fx.update(name=name, hp=hp_over, train=True, state_dict=sd0, guided_g=hp.guided_attention_toleration,
              grads={k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    sd1 = model.state_dict()
    fx['bn_stats'] = {k: sd1[k].detach().clone() for k in sd1 if k.endswith(('running_mean', 'running_var'))}
    return fx


def run_trajectory(name, overrides, B, L, T, n_steps, ref_mods):
    """n_steps consecutive iterations of the reference's training loop (train.py:58-93) on fresh batches."""
    hp, Tacotron, TacotronLoss, defaults = ref_mods
    _reset_hp(hp, defaults, SMALL, overrides)
    torch.manual_seed(50 + list(TRAJECTORIES).index(name))
    model = Tacotron()
    g = torch.Generator().manual_seed(11)
    _randomize_bn(model, g)
    model.train(True)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    optimizer = torch.optim.Adam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)      # train.py:260
    crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
    names = {id(p): k for k, p in model.named_parameters()}
    steps = []
    for s in range(n_steps):
        batch = _train_batch(hp, B, L, T, g)
        guided_g = crit._g
        optimizer.zero_grad()
        loss, r = _recorded_step(model, crit, hp, batch, T, seed=1234 + s)
        loss.backward()                                                                                      # train.py:83
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), hp.gradient_clipping)                      # train.py:84
        optimizer.step()                                                                                     # train.py:85
        crit.update_states()                                                                                 # train.py:93
        sd = model.state_dict()
        r.update(guided_g=guided_g, grads=grads, grad_norm=float(norm),
                 state_after={k: v.detach().clone() for k, v in sd.items()},
                 adam_after={names[id(p)]: dict(exp_avg=st['exp_avg'].clone(), exp_avg_sq=st['exp_avg_sq'].clone(), step=int(st['step']))
                             for p, st in optimizer.state.items()},
                 criterion_after=dict(crit.state_dict()))
        steps.append(r)
    return dict(name=name, hp={**SMALL, **overrides}, train=True, state_dict=sd0, steps=steps,
                optimizer=dict(lr=hp.learning_rate, weight_decay=hp.weight_decay, clip=hp.gradient_clipping),
                criterion=dict(steps=hp.guided_attention_steps, g0=hp.guided_attention_toleration, gamma=hp.guided_attention_gain))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--reference', default='/root/reference')
    ap.add_argument('--only', default=None)
    ap.add_argument('--real', default=None, metavar='DIR', help='write the real-width cases (large, not committed) to DIR and nothing else')
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    import utils  # noqa: F401  (must precede modules.tacotron2: circular import in the reference)
    from modules.tacotron2 import Tacotron, TacotronLoss
    from params.params import Params as hp
    torch.set_num_threads(4)
    defaults = dict(hp.state_dict())
    ref_mods = (hp, Tacotron, TacotronLoss, defaults)
    if args.real:
        os.makedirs(args.real, exist_ok=True)
        for name, (json_name, ov, B, L, T) in REAL_CASES.items():
            if args.only and name not in args.only.split(','):
                continue
            fx = run_real_case(name, json_name, ov, B, L, T, ref_mods, args.reference)
            path = os.path.join(args.real, name + '.pt')
            torch.save(fx, path)
            print(f'{name}: {os.path.getsize(path) / 2 ** 20:.0f} MiB')
        return
    os.makedirs(args.out, exist_ok=True)
    for name, (ov, B, L, T, n_steps) in TRAJECTORIES.items():
        if args.only and name not in args.only.split(','):
            continue
        fx = run_trajectory(name, ov, B, L, T, n_steps, ref_mods)
        path = os.path.join(args.out, name + '.pt')
        torch.save(fx, path)
        print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')
    for name, (ov, B, L, T, flags) in CASES.items():
        if args.only and name not in args.only.split(','):
            continue
        fx = run_case(name, ov, B, L, T, flags, (hp, Tacotron, TacotronLoss, defaults))
        path = os.path.join(args.out, name + '.pt')
        torch.save(fx, path)
        print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
