"""mtts_gen_params_fwd / _bwd (csrc/generator.hip): the generated-encoder's kernel generator writing the implicit-GEMM layout
directly, against torch (reference Conv1dGenerated.forward modules/generated.py:34-42: Linear -> view(O, I/G, k))."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('G,bott,Og,Cg,k', [(5, 4, 512, 256, 3), (10, 8, 256, 256, 1), (2, 3, 16, 8, 3), (1, 1, 3, 5, 2), (5, 4, 7, 300, 5)])
def test_generated_kernel_matches_linear_view_and_its_autograd(G, bott, Og, Cg, k):
    from multilingual_text_to_speech_amd import kernels as K
    g = torch.Generator().manual_seed(G * 100 + k)
    per = Og * Cg * k
    hidden = torch.randn(G, bott, generator=g).cuda().requires_grad_(True)
    wk = (torch.randn(per, bott, generator=g) / bott ** 0.5).cuda().requires_grad_(True)
    bk = torch.randn(per, generator=g).cuda().requires_grad_(True)
    upstream = torch.randn(G * Og, k, Cg, generator=g).cuda()
    wp = K.generated_kernel(hidden, wk, bk, Og, Cg, k)
    (wp * upstream).sum().backward()
    got = [hidden.grad.clone(), wk.grad.clone(), bk.grad.clone()]
    for t in (hidden, wk, bk):
        t.grad = None
    flat = torch.nn.functional.linear(hidden.double(), wk.double(), bk.double())            # [G, per]
    ref = flat.view(G * Og, Cg, k).permute(0, 2, 1)                                         # reference view, then [O, k, I/G]
    assert wp.shape == ref.shape
    assert (wp.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    (ref * upstream.double()).sum().backward()
    for a, t, name in zip(got, (hidden, wk, bk), ('hidden', 'w_kernel', 'b_kernel')):
        r = t.grad.double()
        assert (a.double() - r).abs().max().item() <= 2e-5 * max(1.0, r.abs().max().item()), name


@pytest.mark.parametrize('B,L,S', [(4, 9, 5), (10, 120, 91), (3, 7, 130)])
def test_masked_cross_entropy_matches_the_reference_loss(B, L, S):
    """mtts_masked_cross_entropy vs the reference's ReversalClassifier.loss (modules/classifier.py:62-69: F.cross_entropy with
    padding as ignore_index) - value and gradient."""
    import torch.nn.functional as F
    from multilingual_text_to_speech_amd.modules.classifier import ReversalClassifier
    g = torch.Generator().manual_seed(B * 7 + S)
    pred = (3 * torch.randn(B, L, S, generator=g)).cuda().requires_grad_(True)
    lengths = torch.randint(1, L + 1, (B,), generator=g); lengths[0] = L
    speakers = torch.randint(0, S, (B,), generator=g)
    loss = ReversalClassifier.loss(lengths.cuda(), speakers.cuda(), pred)
    (loss * 0.7).backward()
    got = pred.grad.clone()
    ref_in = pred.detach().double().cpu().requires_grad_(True)
    target = speakers.repeat(L, 1).transpose(0, 1).clone()
    target[~(torch.arange(L)[None, :] < lengths[:, None])] = -100
    ref = F.cross_entropy(ref_in.transpose(1, 2), target, ignore_index=-100)
    (ref * 0.7).backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    assert (got.double().cpu() - ref_in.grad).abs().max().item() <= 1e-6
