"""Debug: ConvBlock (conv + BatchNorm + activation) forward / backward vs torch fp64 over batch sizes around 63."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from multilingual_text_to_speech_amd.modules.layers import ConvBlock


def case(N_, L, C, act):
    torch.manual_seed(4)
    blk = ConvBlock(C, C, 5, 0.0, act).cuda().train()
    A = {'relu': torch.nn.ReLU(), 'tanh': torch.nn.Tanh(), 'identity': torch.nn.Identity()}[act]
    ref = torch.nn.Sequential(torch.nn.Conv1d(C, C, 5, padding=2, bias=False), torch.nn.BatchNorm1d(C), A).double().train()
    ref[0].weight.data.copy_(blk._block[1].weight.data.double().cpu()); ref[1].weight.data.uniform_(0.5, 1.5); ref[1].bias.data.normal_()
    blk._block[2].weight.data.copy_(ref[1].weight.data.float()); blk._block[2].bias.data.copy_(ref[1].bias.data.float())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N_, L, C, generator=g); dy = torch.randn(N_, L, C, generator=g)
    xg = x.cuda().requires_grad_(True)
    y = blk(xg); y.backward(dy.cuda())
    xr = x.double().transpose(1, 2).requires_grad_(True)
    yr = ref(xr); yr.backward(dy.double().transpose(1, 2))
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
    dxe = (xg.grad.double().cpu() - xr.grad.transpose(1, 2)).abs()
    nbad = int((dxe > 1e-4 * xr.grad.abs().max()).sum())
    print(f'convblock {act:8s} N={N_} L={L} C={C}: y {rel(y, yr.detach().transpose(1, 2)):.2e} dx {rel(xg.grad, xr.grad.transpose(1, 2)):.2e} (bad elems {nbad}) '
          f'dW {rel(blk._block[1].weight.grad, ref[0].weight.grad):.2e} dgamma {rel(blk._block[2].weight.grad, ref[1].weight.grad):.2e} '
          f'dbeta {rel(blk._block[2].bias.grad, ref[1].bias.grad):.2e}', flush=True)


for act in ('tanh', 'relu'):
    for N_ in (61, 62, 63, 64, 65):
        for L in (40, 128):
            case(N_, L, 512, act)
case(63, 128, 256, 'tanh'); case(63, 128, 64, 'tanh'); case(126, 64, 512, 'tanh'); case(21, 384, 512, 'tanh')
