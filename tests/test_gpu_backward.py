"""GPU parity of the full training step: HIP forward + loss + HIP backward vs gradients recorded from the
reference implementation (fixtures).  Tolerance: 1e-3 relative to the largest gradient entry of each tensor
(plus 2e-5 absolute), i.e. the same 1e-3 class as the mel bound; observed errors are ~1e-5."""
import pytest
import torch

from tests.helpers import build_hip_model, golden_names, hip_forward, load_golden

pytestmark = pytest.mark.gpu

# every training fixture: fully teacher forced (two-chain schedule), mixed teacher forcing (general schedule), zoneout
CASES = [n for n in golden_names('train') if n != 'simple_eval']


def run_step(fx, device='cuda'):
    from multilingual_text_to_speech_amd.modules.tacotron2 import TacotronLoss
    from multilingual_text_to_speech_amd.params import Params as hp
    model = build_hip_model(fx, device)
    post, pre, stop, align, spk, enc = hip_forward(fx, model, device, tf=0.5 if 'mixed_tf' in fx['name'] else 1.0)
    crit = TacotronLoss(hp.guided_attention_steps, fx['guided_g'], hp.guided_attention_gain)
    to = lambda t: None if t is None else t.to(device)
    loss, parts = crit(fx['text_length'].to(device), fx['target_length'].to(device), pre, to(fx['target']), post, to(fx['target']),
                       stop, to(fx['stop_target']), align, to(fx['speakers']), spk, enc, None)
    loss.backward()
    torch.cuda.synchronize()
    return model, loss, parts


@pytest.mark.parametrize('name', CASES)
def test_loss_and_gradients_match_reference(name):
    fx = load_golden(name)
    model, loss, parts = run_step(fx)
    assert abs(loss.item() - fx['loss'].item()) <= 1e-4 * max(1.0, abs(fx['loss'].item())), (loss.item(), fx['loss'].item())
    grads = {k: p.grad for k, p in model.named_parameters()}
    worst = []
    for k, ref in fx['grads'].items():
        assert grads.get(k) is not None, f'{name}: no gradient for {k}'
        got = grads[k].cpu()
        err = (got - ref).abs().max().item()
        tol = 1e-3 * ref.abs().max().item() + 2e-5
        worst.append((err / tol, k, err))
        assert err <= tol, f'{name}/{k}: max |delta| {err:.3e} > {tol:.3e} (|ref|max {ref.abs().max().item():.3e})'
    missing = [k for k, p in model.named_parameters() if p.grad is not None and k not in fx['grads']]
    assert not missing, f'{name}: unexpected gradients {missing}'
