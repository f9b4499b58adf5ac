"""Debug: conv1d / BiLSTM kernels at the encoder shapes of the failing B = 63 gradient case, against torch fp64."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from multilingual_text_to_speech_amd import kernels as K


def conv_case(N_, L, C, O, k):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N_, L, C, generator=g).cuda(); w = (torch.randn(O, C, k, generator=g) / (C * k) ** 0.5).cuda(); dy = torch.randn(N_, L, O, generator=g).cuda()
    wp = K.pack_conv_weight(w)
    y = K.conv1d_fwd(x, wp, k, 1, 1)
    dx, dwp = K.conv1d_bwd(x, wp, dy, k, 1, 1)
    dw = K.unpack_conv_weight(dwp, O, C, k)
    xr = x.double().transpose(1, 2).requires_grad_(True); wr = w.double().requires_grad_(True)
    pl = (k - 1) // 2
    yr = torch.nn.functional.conv1d(torch.nn.functional.pad(xr, (pl, k - 1 - pl)), wr)
    yr.backward(dy.double().transpose(1, 2))
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()
    print(f'conv N={N_} L={L} C={C} O={O} k={k}: y {rel(y, yr.transpose(1, 2)):.2e} dx {rel(dx, xr.grad.transpose(1, 2)):.2e} dw {rel(dw, wr.grad):.2e}')


def bilstm_case(B, L, C, H):
    g = torch.Generator().manual_seed(2)
    lstm = torch.nn.LSTM(C, H, batch_first=True, bidirectional=True).double()
    x = torch.randn(B, L, C, generator=g)
    lens = torch.sort(torch.randint(L // 2, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    dy = torch.randn(B, L, 2 * H, generator=g)
    xr = x.double().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lens, batch_first=True)
    out, _ = lstm(packed)
    yr, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=L)
    yr.backward(dy.double())
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l0_reverse', 'weight_hh_l0_reverse', 'bias_ih_l0_reverse', 'bias_hh_l0_reverse']
    ps = [getattr(lstm, n).detach().float().cuda().requires_grad_(True) for n in names]
    xg = x.cuda().requires_grad_(True)
    y = K.bilstm(xg, lens, ps)
    y.backward(dy.cuda())
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
    msg = f'bilstm B={B} L={L} C={C} H={H}: y {rel(y, yr.detach()):.2e} dx {rel(xg.grad, xr.grad):.2e}'
    for n, p in zip(names, ps):
        msg += f' {n[:9]}{"r" if n.endswith("reverse") else ""} {rel(p.grad, getattr(lstm, n).grad):.1e}'
    print(msg)


for (N_, L) in ((63, 40), (64, 40), (47, 40), (63, 128)):
    conv_case(N_, L, 512, 512, 5)
for (B, L) in ((63, 40), (64, 40), (47, 40), (63, 128), (62, 40), (49, 40)):
    bilstm_case(B, L, 512, 256)


def gemm_case(M, N, Kd, beta):
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, Kd, generator=g).cuda(); Bm = torch.randn(N, Kd, generator=g).cuda().t().contiguous()      # B stored [K, N]: transB
    C0 = torch.randn(M, N, generator=g).cuda(); C = C0.clone()
    K.gemm(A, Bm, C, M, N, Kd, Kd, N, N, transB=True, beta=beta)
    ref = A.double() @ Bm.double() + beta * C0.double()
    print(f'gemm M={M} N={N} K={Kd} transB beta={beta}: {((C.double() - ref).abs().max() / ref.abs().max()).item():.2e}')


def bilstm_masked_case(B, L, C, H):
    g = torch.Generator().manual_seed(2)
    lstm = torch.nn.LSTM(C, H, batch_first=True, bidirectional=True).double()
    x = torch.randn(B, L, C, generator=g)
    lens = torch.sort(torch.randint(L // 2, L + 1, (B,), generator=g), descending=True).values; lens[0] = L
    dy = torch.randn(B, L, 2 * H, generator=g)
    xr = x.double().requires_grad_(True)
    out, _ = lstm(torch.nn.utils.rnn.pack_padded_sequence(xr, lens, batch_first=True))
    yr, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=L)
    yr.backward(dy.double())
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l0_reverse', 'weight_hh_l0_reverse', 'bias_ih_l0_reverse', 'bias_hh_l0_reverse']
    ps = [getattr(lstm, n).detach().float().cuda().requires_grad_(True) for n in names]
    xg = x.cuda().requires_grad_(True)
    y = K.bilstm(xg, lens, ps)
    y.backward(dy.cuda())                      # dy NOT masked: the kernels must ignore it at padded positions
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
    print(f'bilstm (unmasked dy) B={B} L={L}: y {rel(y, yr.detach()):.2e} dx {rel(xg.grad, xr.grad):.2e} ' +
          ' '.join(f'{rel(p.grad, getattr(lstm, n).grad):.1e}' for n, p in zip(names, ps)))


def convblock_case(N_, L, C):
    from multilingual_text_to_speech_amd.modules.layers import ConvBlock
    torch.manual_seed(4)
    blk = ConvBlock(C, C, 5, 0.0, 'relu').cuda().train()
    ref = torch.nn.Sequential(torch.nn.Conv1d(C, C, 5, padding=2, bias=False), torch.nn.BatchNorm1d(C), torch.nn.ReLU()).double().train()
    ref[0].weight.data.copy_(blk._block[1].weight.data.double().cpu()); ref[1].weight.data.uniform_(0.5, 1.5); ref[1].bias.data.normal_()
    blk._block[2].weight.data.copy_(ref[1].weight.data.float()); blk._block[2].bias.data.copy_(ref[1].bias.data.float())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N_, L, C, generator=g); dy = torch.randn(N_, L, C, generator=g)
    xg = x.cuda().requires_grad_(True)
    y = blk(xg); y.backward(dy.cuda())
    xr = x.double().transpose(1, 2).requires_grad_(True)
    yr = ref(xr); yr.backward(dy.double().transpose(1, 2))
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
    print(f'convblock N={N_} L={L} C={C}: y {rel(y, yr.detach().transpose(1, 2)):.2e} dx {rel(xg.grad, xr.grad.transpose(1, 2)):.2e} '
          f'dW {rel(blk._block[1].weight.grad, ref[0].weight.grad):.2e} dgamma {rel(blk._block[2].weight.grad, ref[1].weight.grad):.2e} '
          f'dbeta {rel(blk._block[2].bias.grad, ref[1].bias.grad):.2e}')


for M in (2520, 2560, 1880, 8064):
    for beta in (0.0, 1.0):
        gemm_case(M, 512, 1024, beta)
for (B, L) in ((63, 40), (64, 40)):
    bilstm_masked_case(B, L, 512, 256)
for (N_, L) in ((63, 40), (64, 40), (47, 40), (63, 128)):
    convblock_case(N_, L, 512)
