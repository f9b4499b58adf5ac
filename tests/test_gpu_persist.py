"""The three forms of the teacher-forced decoder forward agree (csrc/persist.hip, csrc/decoder.hip):
  * persistent kernels with the polled hand-off of h (default) vs the barrier-only hand-off (MTTS_PDEC_POLL=0): the same bits - only
    the synchronisation differs;
  * persistent kernels vs the per-step launch schedule (MTTS_PERSIST=0): the same arithmetic in another summation order - fp32 noise.
The switches are read once per process, so every form runs in a child (reference path: Decoder._decode, modules/tacotron2.py:148-209)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import sys, torch
sys.path.insert(0, %(root)r)
import bench
from multilingual_text_to_speech_amd.params import presets, Params as hp
from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
presets.apply('shared_training', speaker_number=91)
torch.manual_seed(11)
dev = torch.device('cuda:0')
model = Tacotron().to(dev).train()
B, L, T = 64, 40, 61
batch = bench.synthetic_batch(hp, B, L, T, dev, seed=5)
torch.manual_seed(12)                                   # dropout draws
with torch.no_grad():
    post, pre, stop, align, spk, enc = model(batch['text'], batch['text_length'], batch['target'], batch['target_length'],
                                             batch['speakers'], batch['languages'], 1.0)
torch.cuda.synchronize()
from multilingual_text_to_speech_amd import kernels
kernels.check_device_errors()
torch.save({'post': post.cpu(), 'pre': pre.cpu(), 'stop': stop.cpu(), 'align': align.cpu()}, sys.argv[1])
'''


def _run(tmp_path, name, env_add):
    path = str(tmp_path / f'{name}.pt')
    r = subprocess.run([sys.executable, '-c', _CHILD % {'root': ROOT}, path], env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return torch.load(path)


def test_decoder_forward_forms_agree(tmp_path):
    polled = _run(tmp_path, 'polled', {})
    barrier = _run(tmp_path, 'barrier', {'MTTS_PDEC_POLL': '0'})
    steps = _run(tmp_path, 'steps', {'MTTS_PERSIST': '0'})
    for k, v in polled.items():
        assert torch.isfinite(v).all(), k
        assert torch.equal(v, barrier[k]), f'{k}: polled and barrier-only hand-off differ'
        ref = steps[k]
        rel = ((v.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30)).item()
        assert rel <= 2e-5, f'{k}: persistent vs per-step launches, relative L2 {rel:.3e}'


# Shape edges of the persistent kernels (csrc/persist.hip: 1..4 row tiles of 16, ragged last tile, a single sample, the maximal
# encoder length 128 and a very short one, both memory widths Dm = 544 / 288, ragged text lengths): forward in train mode with
# injected dropout draws against the CPU oracle (mel outputs, stop logits, alignments).
@pytest.mark.parametrize('preset,B,L,T', [('shared_training', 1, 7, 10), ('shared_training', 17, 128, 6), ('shared_training', 33, 40, 12),
                                          ('shared_training', 49, 25, 9), ('shared_training', 63, 128, 5), ('generated_switching', 40, 30, 8)])
def test_persistent_kernels_shape_edges_match_oracle(preset, B, L, T):
    from tests.test_gpu_more import run_train_step_case
    run_train_step_case(preset, B, L, T, {}, check_grads=False)
