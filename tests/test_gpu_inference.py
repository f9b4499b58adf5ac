"""Batched synthesis (Tacotron.inference_batch: K-split step kernels on the general schedule, fused two-layer prenet step,
per-sample lengths and stop rule) against the CPU ORACLE looping batch-1 - the reference's own inference semantics
(modules/tacotron2.py:201-207,216-219,387-408) - at the real layer widths."""
import pytest
import torch

from oracle import tacotron_oracle as O

pytestmark = pytest.mark.gpu


def _oracle_batch1(sd, cfg, hp, text, spk, lang_w, keep, max_frames, stop_rule=True):
    """One utterance through the oracle exactly like Tacotron.inference: batch 1, per-character language weights."""
    L = text.numel()
    emb = torch.nn.functional.embedding(text.view(1, L), sd['_embedding.weight'], padding_idx=0)
    lw = lang_w.view(1, L, -1) if lang_w is not None else None
    enc = O.encode(sd, cfg, emb, torch.tensor([L]), lw, None, False)
    lang_ids = torch.argmax(lw, dim=2) if lw is not None else None
    spk_ids = torch.full((1, L), int(spk), dtype=torch.int64) if spk is not None else None
    masks = {f'prenet_step.{i}': (k.float() / (1 - hp.dropout)) for i, k in enumerate(keep)}       # [T,1,P] multipliers
    frames, stops, _ = O.decode(sd, cfg, enc, torch.ones(1, L, dtype=torch.bool), None, None, spk_ids, lang_ids, masks, False,
                                max_frames=max_frames, stop_rule=stop_rule)
    post = O.postnet(sd, cfg, frames.transpose(1, 2), None, False)
    return post[0], stops[0]


@pytest.mark.parametrize('preset,lens', [('generated_switching', [30, 25, 18, 30, 7]), ('shared_training', [26, 26, 11])])
def test_batched_inference_matches_oracle_batch1_loop(preset, lens):
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    from multilingual_text_to_speech_amd.masks import provider
    max_frames = 36
    presets.apply(preset, speaker_number=7, max_output_length=max_frames)
    n_lang = len(hp.languages)
    done = False
    for seed in range(8):
        torch.manual_seed(seed)
        model = Tacotron()
        g = torch.Generator().manual_seed(100 + seed)
        with torch.no_grad():
            for k, v in model.state_dict().items():         # eval-mode BatchNorm needs plausible running statistics (SURVEY 8c recipe 3)
                if k.endswith('running_var'):
                    v.copy_(torch.empty(v.shape).uniform_(300.0, 900.0, generator=g) if '_encoder' in k and preset != 'shared_training'
                            else torch.empty(v.shape).uniform_(0.5, 1.5, generator=g))
            # a stop head that fires within the window, with margins fp32 noise cannot flip
            model._decoder._stop_prediction.weight.mul_(6.0)
        model.eval()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        cfg = O.cfg_from_params(hp)
        texts = [torch.cat((torch.randint(3, hp.symbols_count() + 3, (n - 1,), generator=g), torch.tensor([1]))) for n in lens]
        langs = None
        if hp.multi_language:
            langs = []
            for i, n in enumerate(lens):
                w = torch.zeros(n, n_lang); w[:, i % n_lang] = 1.0
                if i == 0 and hp.encoder_type == 'generated':
                    w[n // 2:] = 0.0; w[n // 2:, (i + 1) % n_lang] = 1.0          # code switching inside one utterance
                langs.append(w)
        spks = [i % hp.speaker_number for i in range(len(lens))] if hp.multi_speaker else None
        P, n_pre = hp.prenet_dimension, hp.prenet_layers
        draws = [(torch.rand(max_frames, len(lens), P, generator=g) >= hp.dropout).to(torch.uint8) for _ in range(n_pre)]
        refs, margin = [], 1e9
        for i, t in enumerate(texts):
            post, stops = _oracle_batch1(sd, cfg, hp, t, None if spks is None else spks[i], None if langs is None else langs[i],
                                         [d[:, i:i + 1] for d in draws], max_frames)
            refs.append(post)
            margin = min(margin, stops.abs().min().item())
        if margin < 2e-2 or all(r.shape[1] == max_frames for r in refs):
            continue                     # a stop logit too close to the threshold, or the stop rule never fired: next seed
        model.cuda()
        provider.injected = {f'dec.prenet.{k}': draws[k].cuda() for k in range(n_pre)}
        try:
            outs = model.inference_batch(texts, spks, langs)
        finally:
            provider.injected = None
        for i, (o, r) in enumerate(zip(outs, refs)):
            assert o.shape == r.shape, (preset, seed, i, o.shape, r.shape)
            err = (o.cpu() - r).abs().max().item()
            assert err <= 1e-3, f'{preset} seed {seed} utterance {i}: max |delta| = {err:.3e}'
        assert len({r.shape[1] for r in refs}) > 1 or min(r.shape[1] for r in refs) < max_frames
        done = True
        break
    assert done, 'no seed produced a stop trajectory with safe margins'


def _synthesis_case(preset, lens, lang_of, max_frames, seed=0):
    """Model (eval, plausible BatchNorm running statistics), utterances with ONE language each (lang_of[i]) and prenet draws."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    presets.apply(preset, speaker_number=7, max_output_length=max_frames)
    n_lang = len(hp.languages)
    torch.manual_seed(seed)
    model = Tacotron()
    g = torch.Generator().manual_seed(100 + seed)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('running_var'):
                v.copy_(torch.empty(v.shape).uniform_(300.0, 900.0, generator=g) if '_encoder' in k and preset != 'shared_training'
                        else torch.empty(v.shape).uniform_(0.5, 1.5, generator=g))
    model.eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    texts = [torch.cat((torch.randint(3, hp.symbols_count() + 3, (n - 1,), generator=g), torch.tensor([1]))) for n in lens]
    langs = []
    for i, n in enumerate(lens):
        w = torch.zeros(n, n_lang); w[:, lang_of[i]] = 1.0
        langs.append(w)
    spks = [(3 * i + 1) % hp.speaker_number for i in range(len(lens))]
    draws = [(torch.rand(max_frames, len(lens), hp.prenet_dimension, generator=g) >= hp.dropout).to(torch.uint8) for _ in range(hp.prenet_layers)]
    return hp, model, sd, texts, langs, spks, draws


def _synthesize(model, texts, spks, langs, draws, **kw):
    from multilingual_text_to_speech_amd.masks import provider
    provider.injected = {f'dec.prenet.{k}': d.cuda() for k, d in enumerate(draws)}
    try:
        return model.inference_batch(texts, spks, langs, **kw)
    finally:
        provider.injected = None


def test_large_pure_language_batch_free_running_matches_oracle_batch1_loop():
    """BASELINE configs[4]'s path as bench.py's `inference` leg runs it: >= 128 utterances of up to 201 tokens, every utterance in ONE
    language -> `_pure_languages` -> `_compact_groups` (modules/encoder.py; reference expansion + blend modules/encoder.py:196-221),
    then the free-running loop of a large batch - fused prenet step, `lstm_fused2_kernel` for both LSTMs, `attn_step_big_kernel<8>`
    (two workgroups per sample at L = 201), `skinny_proj_kernel` - on the model's OWN frames for 7 frames (reference
    modules/tacotron2.py:180-207).  Compared with the oracle looping batch 1 (the reference's inference semantics) on a
    deterministic subset: every utterance of the short length buckets and every fourth of the 201-token bucket."""
    n_utt, max_frames = 132, 7
    lens = [201] * 104 + [176] * 14 + [150] * 9 + [97] * 5            # four encoder buckets, the 201-token one above 64 rows per group
    lang_of = [(i * i + i // 3) % 5 for i in range(n_utt)]            # uneven language counts inside every bucket
    hp, model, sd, texts, langs, spks, draws = _synthesis_case('generated_switching', lens, lang_of, max_frames)
    assert len(set(lang_of[:104])) == 5 and len({lang_of[:104].count(l) for l in range(5)}) > 1
    cfg = O.cfg_from_params(hp)
    model.cuda()
    outs = _synthesize(model, texts, spks, langs, draws, stop_threshold=2.0)          # stop rule off: all 7 frames for everyone
    check = list(range(0, 104, 4)) + list(range(104, n_utt))
    torch.set_flush_denormal(True)
    worst = 0.0
    with torch.no_grad():
        for i in check:
            ref, _ = _oracle_batch1(sd, cfg, hp, texts[i], spks[i], langs[i], [d[:, i:i + 1] for d in draws], max_frames, stop_rule=False)
            assert outs[i].shape == ref.shape == (hp.num_mels, max_frames)
            err = (outs[i].cpu() - ref).abs().max().item()
            worst = max(worst, err)
            assert err <= 1e-3, f'utterance {i} (L = {lens[i]}, language {lang_of[i]}): max |delta| = {err:.3e}'
    print(f'large pure-language batch: {len(check)} utterances checked, worst max |delta| = {worst:.2e}')


def test_compact_language_groups_with_uneven_counts_and_against_the_expansion(monkeypatch):
    """Seven pure-language utterances of one length with language counts 3 / 0 / 1 / 2 / 1 (zero-padded slots in the compact layout,
    one EMPTY group) against the oracle's batch-1 loop; then the same batch with the compaction switched off (every utterance
    expanded to all G groups and blended, reference modules/encoder.py:120-127,203-221): equal per utterance.  Finally a MIXED batch
    (one code-switching utterance -> `_pure_languages` is None -> expansion) whose pure utterances must equal their compact results."""
    from multilingual_text_to_speech_amd.modules import encoder as E
    max_frames = 6
    lens = [23] * 7
    lang_of = [0, 2, 0, 3, 4, 3, 0]
    hp, model, sd, texts, langs, spks, draws = _synthesis_case('generated_switching', lens, lang_of, max_frames, seed=1)
    cfg = O.cfg_from_params(hp)
    model.cuda()
    calls = []
    real = E._compact_groups
    monkeypatch.setattr(E, '_compact_groups', lambda *a: (calls.append(1), real(*a))[1])
    compact = _synthesize(model, texts, spks, langs, draws, stop_threshold=2.0)
    assert calls, 'the pure-language batch did not take the compact layout'
    torch.set_flush_denormal(True)
    with torch.no_grad():
        for i in range(7):
            ref, _ = _oracle_batch1(sd, cfg, hp, texts[i], spks[i], langs[i], [d[:, i:i + 1] for d in draws], max_frames, stop_rule=False)
            err = (compact[i].cpu() - ref).abs().max().item()
            assert err <= 1e-3, f'compact, utterance {i}: max |delta| = {err:.3e}'
    n_calls = len(calls)
    monkeypatch.setattr(E, '_pure_languages', lambda x_langs: None)
    expanded = _synthesize(model, texts, spks, langs, draws, stop_threshold=2.0)
    assert len(calls) == n_calls
    for i, (a, b) in enumerate(zip(compact, expanded)):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-5, (i, (a - b).abs().max().item())
    monkeypatch.undo()
    mixed = [w.clone() for w in langs]
    mixed[2][11:] = 0.0; mixed[2][11:, 1] = 1.0                       # utterance 2 switches language in the middle
    assert E._pure_languages(torch.stack(mixed)) is None and E._pure_languages(torch.stack(langs)) is not None
    out = _synthesize(model, texts, spks, mixed, draws, stop_threshold=2.0)
    for i in (0, 1, 3, 4, 5, 6):
        assert (out[i] - compact[i]).abs().max().item() <= 2e-5, i
    with torch.no_grad():
        ref, _ = _oracle_batch1(sd, cfg, hp, texts[2], spks[2], mixed[2], [d[:, 2:3] for d in draws], max_frames, stop_rule=False)
    assert (out[2].cpu() - ref).abs().max().item() <= 1e-3


def test_graph_replayed_decode_equals_eager(monkeypatch):
    """MTTS_DECODE_GRAPH=1 (BASELINE configs[4]: hipGraph-captured decode steps): the third call with the same shapes replays
    captured graphs for every chunk and must reproduce, bit for bit, the first call of the session (which executes the same kernels
    with the same arguments eagerly).  The graph path pads the memory to its 32-position length bucket (one session per bucket
    instead of one per utterance length), so against the un-padded eager decode only the summation order of the context
    differs: equal to fp32 rounding."""
    from multilingual_text_to_speech_amd.params import presets, Params as hp
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    from multilingual_text_to_speech_amd.masks import provider
    from multilingual_text_to_speech_amd import decoder_ops as D
    presets.apply('generated_switching', speaker_number=7, max_output_length=70)
    torch.manual_seed(0)
    model = Tacotron().cuda().eval()
    g = torch.Generator().manual_seed(3)
    lens = [21, 21, 13, 8]
    texts = [torch.cat((torch.randint(3, hp.symbols_count() + 3, (n - 1,), generator=g), torch.tensor([1]))) for n in lens]
    n_lang = len(hp.languages)
    langs = []
    for i, n in enumerate(lens):
        w = torch.zeros(n, n_lang); w[:, i % n_lang] = 1.0
        langs.append(w)
    spks = [i % hp.speaker_number for i in range(len(lens))]
    draws = {f'dec.prenet.{k}': (torch.rand(70, len(lens), hp.prenet_dimension, generator=g) >= hp.dropout).to(torch.uint8).cuda() for k in range(2)}

    def run():
        provider.injected = dict(draws)
        try:
            return model.inference_batch(texts, spks, langs, stop_threshold=2.0)       # stop rule off: all 70 frames, 3 chunks
        finally:
            provider.injected = None
    eager = run()
    monkeypatch.setenv('MTTS_DECODE_GRAPH', '1')
    outs = [run() for _ in range(3)]
    sess = next(iter(D.GraphedDecode._cache.values()))
    assert sess.replayed == 3, sess.replayed                 # 70 frames = chunks of 32, 32, 6: all replayed on the third call
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)
    for a, b in zip(outs[0], eager):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-5
