"""Pin the CPU oracle against fixtures recorded from the reference implementation itself
(oracle/make_golden.py).  CPU only."""
import os

import pytest
import torch

from oracle import tacotron_oracle as O
from tests.helpers import cfg_of, golden_names, load_golden, oracle_replay

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
