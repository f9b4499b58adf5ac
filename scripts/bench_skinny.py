"""Micro-benchmark of the step kernels (run under rocprofv3; durations are read from the trace)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multilingual_text_to_speech_amd import _C
from multilingual_text_to_speech_amd._C import lib, ptr, stream_ptr, check

dev = 'cuda'
def run(B, H, Ks, lstm, ksplit=1, N=None, reps=50, tag=''):
    a = _C.SkinnyArgs()
    xs = [torch.randn(B, K, device=dev) for K in Ks]
    Nw = 4 * H if lstm else N
    ws = [torch.randn(Nw, K, device=dev) * 0.05 for K in Ks]
    a.nseg = len(Ks)
    for i, K in enumerate(Ks):
        a.seg[i].x, a.seg[i].w, a.seg[i].K, a.seg[i].ldx, a.seg[i].ldw = xs[i].data_ptr(), ws[i].data_ptr(), K, K, K
    a.B, a.N, a.ksplit = B, Nw, ksplit
    out = torch.zeros(max(ksplit, 1), B, Nw, device=dev)
    a.out, a.ldo, a.out_ks = ptr(out), Nw, B * Nw
    if lstm:
        a.lstm, a.H = 1, H
        hp, cp = torch.randn(B, H, device=dev), torch.randn(B, H, device=dev)
        ho, co = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
        gates = torch.empty(B, 4 * H, device=dev)
        bi, bh = torch.randn(4 * H, device=dev), torch.randn(4 * H, device=dev)
        pre = torch.randn(B, 4 * H, device=dev)
        a.pre, a.ldpre, a.b_ih, a.b_hh = ptr(pre), 4 * H, ptr(bi), ptr(bh)
        a.h_prev, a.c_prev, a.h_out, a.c_out, a.gates_out = ptr(hp), ptr(cp), ptr(ho), ptr(co), ptr(gates)
    for _ in range(reps):
        check(lib().mtts_skinny_gemm(ctypes.byref(a), stream_ptr()), 'skinny')
    torch.cuda.synchronize()

run(64, 1024, [544, 1024], True, tag='att fast')
run(64, 1024, [1024], True, tag='gen fast')
run(64, 1024, [256, 544, 1024], True, tag='att general')
run(64, 256, [256], True, tag='bilstm')
run(64, 0, [1024], False, ksplit=8, N=128, tag='qproj')
run(64, 0, [16], False, ksplit=1, N=16, tag='floor')
run(256, 1024, [288, 1024], True, tag='att B256')
