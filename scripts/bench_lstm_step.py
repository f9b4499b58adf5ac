"""Micro-benchmark of the attention-LSTM step: K-split path (lstm_gates_kernel + lstm_cell_q_kernel) vs the skinny kernel,
back-to-back launches on one stream, HIP events.  python scripts/bench_lstm_step.py [--batch 64] [--dm 544]"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multilingual_text_to_speech_amd import _C
from multilingual_text_to_speech_amd._C import check, lib, ptr, stream_ptr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--dm', type=int, default=544)
    ap.add_argument('--iters', type=int, default=200)
    args = ap.parse_args()
    B, H, Dm, A = args.batch, 1024, args.dm, 128
    dev = 'cuda'
    L = lib()
    x0, x1 = torch.randn(B, Dm, device=dev), torch.randn(B, H, device=dev)
    w0, w1 = torch.randn(4 * H, Dm, device=dev) * 0.02, torch.randn(4 * H, H, device=dev) * 0.02
    b = torch.zeros(4 * H, device=dev)
    pre = torch.randn(B, 4 * H, device=dev)
    c_prev = torch.randn(B, H, device=dev)
    wq = torch.randn(A, H, device=dev) * 0.03
    out = {}
    for prec in (0, 1):
        packed = torch.empty(int(L.mtts_lstm_packed_weight_bytes(H, Dm + H, prec)), dtype=torch.uint8, device=dev)
        bias_u = torch.empty(4 * H, device=dev)
        pk = _C.LstmPackArgs()
        pk.w[0], pk.K[0], pk.ldw[0] = w0.data_ptr(), Dm, Dm
        pk.w[1], pk.K[1], pk.ldw[1] = w1.data_ptr(), H, H
        pk.nseg, pk.H, pk.precision, pk.dst, pk.b_ih, pk.b_hh, pk.bias_u = 2, H, prec, ptr(packed), ptr(b), ptr(b), ptr(bias_u)
        check(L.mtts_lstm_pack_weights(ctypes.byref(pk), stream_ptr()), 'pack')
        a = _C.LstmStepArgs()
        a.x[0], a.K[0], a.ldx[0] = x0.data_ptr(), Dm, Dm
        a.x[1], a.K[1], a.ldx[1] = x1.data_ptr(), H, H
        a.nseg, a.w_packed, a.precision, a.B, a.H = 2, ptr(packed), prec, B, H
        part = torch.empty(int(L.mtts_lstm_step_partial_floats(B, H, Dm + H)), device=dev)
        h_out, c_out, gates = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, 4 * H, device=dev)
        qpart = torch.empty(H // 16, B, A, device=dev)
        a.partials, a.pre, a.ldpre, a.bias_u, a.c_prev, a.h_out, a.c_out, a.gates_out = ptr(part), ptr(pre), 4 * H, ptr(bias_u), ptr(c_prev), ptr(h_out), ptr(c_out), ptr(gates)
        a.w_query, a.A, a.qpart = ptr(wq), A, ptr(qpart)
        for _ in range(10):
            check(L.mtts_lstm_step_fwd(ctypes.byref(a), stream_ptr()), 'step')
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            check(L.mtts_lstm_step_fwd(ctypes.byref(a), stream_ptr()), 'step')
        e1.record(); e1.synchronize()
        out['ksplit_prec%d_us' % prec] = round(e0.elapsed_time(e1) * 1e3 / args.iters, 2)
    # skinny kernel (row-major operands) + query projection, the round-1 path
    k = _C.SkinnyArgs()
    k.nseg, k.B, k.N, k.ksplit, k.lstm, k.H = 2, B, 4 * H, 1, 1, H
    k.seg[0].x, k.seg[0].w, k.seg[0].K, k.seg[0].ldx, k.seg[0].ldw = ptr(x0), ptr(w0), Dm, Dm, Dm
    k.seg[1].x, k.seg[1].w, k.seg[1].K, k.seg[1].ldx, k.seg[1].ldw = ptr(x1), ptr(w1), H, H, H
    pre_g = torch.randn(B, 4 * H, device=dev)
    h2, c2 = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    k.pre, k.ldpre, k.b_ih, k.b_hh, k.c_prev, k.h_out, k.c_out = ptr(pre_g), 4 * H, ptr(b), ptr(b), ptr(c_prev), ptr(h2), ptr(c2)
    q = _C.SkinnyArgs()
    q.nseg, q.B, q.N, q.ksplit = 1, B, A, 8
    q.seg[0].x, q.seg[0].w, q.seg[0].K, q.seg[0].ldx, q.seg[0].ldw = ptr(h2), ptr(wq), H, H, H
    qp = torch.empty(8, B, A, device=dev)
    q.out, q.ldo, q.out_ks = ptr(qp), A, B * A
    for _ in range(10):
        check(L.mtts_skinny_gemm(ctypes.byref(k), stream_ptr()), 'sk'); check(L.mtts_skinny_gemm(ctypes.byref(q), stream_ptr()), 'q')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        check(L.mtts_skinny_gemm(ctypes.byref(k), stream_ptr()), 'sk'); check(L.mtts_skinny_gemm(ctypes.byref(q), stream_ptr()), 'q')
    e1.record(); e1.synchronize()
    out['skinny_plus_query_us'] = round(e0.elapsed_time(e1) * 1e3 / args.iters, 2)
    out.update(batch=B, Dm=Dm)
    print(out)


if __name__ == '__main__':
    main()
