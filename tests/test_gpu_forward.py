"""GPU parity: HIP forward (through the C-ABI) vs fixtures recorded from the reference implementation.
Tolerance: the north-star bound |delta| <= 1e-3 per mel value (fp32); observed errors are ~1e-5."""
import pytest
import torch

from tests.helpers import build_hip_model, golden_names, hip_forward, load_golden

pytestmark = pytest.mark.gpu
MEL_TOL = 1e-3


@pytest.mark.parametrize('name', golden_names('train'))
def test_forward_matches_reference_fixture(name):
    fx = load_golden(name)
    model = build_hip_model(fx)
    post, pre, stop, align, spk, enc = hip_forward(fx, model)
    torch.cuda.synchronize()
    for key, val in (('encoder_output', enc), ('alignment', align), ('pre', pre), ('post', post), ('stop', stop)):
        ref = fx[key]
        err = (val.cpu() - ref).abs().max().item()
        assert err <= MEL_TOL, f'{name}/{key}: max |delta| = {err:.3e}'
    if fx['speaker_prediction'] is not None:
        assert (spk.cpu() - fx['speaker_prediction']).abs().max().item() <= MEL_TOL
    if fx['train']:
        sd = model.state_dict()
        for k, v in fx['bn_stats'].items():
            assert (sd[k].cpu() - v).abs().max().item() <= 1e-4, f'{name}/{k}'
