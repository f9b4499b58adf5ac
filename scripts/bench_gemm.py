"""Time mtts_gemm_ex on the decoder's big shapes (NT forward, NN dgrad, TT wgrad): python scripts/bench_gemm.py [bf16]
(per call, pack passes of the pre-split core included; MTTS_GEMM_PLANES=0 for the cores that split on the fly)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multilingual_text_to_speech_amd import kernels as K, _C
dev = torch.device('cuda')
if len(sys.argv) > 1 and sys.argv[1] == 'bf16':
    _C.set_precision('bf16')
R = 38400
cases = [('fwd  y=xW^T', 4096, 4096, 4096), ('fwd  y=xW^T', R, 4096, 1536), ('fwd  y=xW^T', R, 4096, 256), ('fwd  y=xW^T', R, 512, 2560), ('fwd  y=xW^T', 3072, 4096, 1536),
         ('dgrad dx=dyW', R, 1536, 4096), ('dgrad dx=dyW', 3072, 1024, 4096), ('wgrad dW=dy^Tx', 4096, 1536, R), ('wgrad dW=dy^Tx', 4096, 1024, R),
         # per-chunk weight gradients of the decoder backward (48 steps x 64 rows)
         ('wgrad dW=dy^Tx', 4096, 1024, 3072), ('wgrad dW=dy^Tx', 4096, 544, 3072), ('wgrad dW=dy^Tx', 4096, 256, 3072)]
for name, M, N, Kd in cases:
    if name.startswith('fwd'):
        A, B = torch.randn(M, Kd, device=dev), torch.randn(N, Kd, device=dev) * 0.1
        C = torch.empty(M, N, device=dev)
        f = lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N)
        ref = lambda: A.double() @ B.double().t()
    elif name.startswith('dgrad'):
        A, B = torch.randn(M, Kd, device=dev), torch.randn(Kd, N, device=dev) * 0.1    # dy [R,N_out], W [N_out, K_in]
        C = torch.empty(M, N, device=dev)
        f = lambda: K.gemm(A, B, C, M, N, Kd, Kd, N, N, transB=True)
        ref = lambda: A.double() @ B.double()
    else:
        A, B = torch.randn(Kd, M, device=dev), torch.randn(Kd, N, device=dev) * 0.1    # dy [R,M], x [R,N]
        C = torch.empty(M, N, device=dev)
        f = lambda: K.gemm(A, B, C, M, N, Kd, M, N, N, transA=True, transB=True)
        ref = lambda: A.double().t() @ B.double()
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    r = ref()
    err = ((C.double() - r).abs().max() / r.abs().max()).item()
    print('%-15s M=%5d N=%5d K=%5d  %.3f ms  %6.1f TFLOP/s  max|err|/max|ref| %.2e' % (name, M, N, Kd, ms, 2.0 * M * N * Kd / ms * 1e-9, err))
print('pre-split core launches:', int(_C.lib().mtts_gemm_planes_count()))
