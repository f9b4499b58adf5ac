"""Generate golden fixtures by EXECUTING THE REFERENCE (imported from /root/reference) on CPU.

    python oracle/make_golden.py [--out tests/golden] [--reference /root/reference]

Test infrastructure only.  Runs in this container (the reference cannot travel to the GPU box);
the produced `tests/golden/*.pt` files are committed.  Each fixture stores: the hyper-parameter
overrides, the reference model's `state_dict`, the inputs, every dropout multiplier the reference
drew (named by call site), the teacher-forcing draw, the six forward outputs, the updated BatchNorm
running statistics, the reference `TacotronLoss` value and the gradient of that loss w.r.t. every
parameter.  Nothing from this repo's product package is imported here (the reference and the
product share the module names `modules`, `params`, `utils`).

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--reference', default='/root/reference')
    ap.add_argument('--only', default=None)
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    import utils  # noqa: F401  (must precede modules.tacotron2: circular import in the reference)
    from modules.tacotron2 import Tacotron, TacotronLoss
    from params.params import Params as hp
    torch.set_num_threads(4)
    defaults = dict(hp.state_dict())
    os.makedirs(args.out, exist_ok=True)
    for name, (ov, B, L, T, flags) in CASES.items():
        if args.only and name not in args.only.split(','):
            continue
        fx = run_case(name, ov, B, L, T, flags, (hp, Tacotron, TacotronLoss, defaults))
        path = os.path.join(args.out, name + '.pt')
        torch.save(fx, path)
        print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
