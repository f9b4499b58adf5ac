"""NT (both operands K-contiguous) GEMM with and without padded leading dimensions: power-of-two row strides vs channel spread."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multilingual_text_to_speech_amd import kernels as K
dev = torch.device('cuda')
def run(M, N, Kd, pada, padb, nosplit=0, transB=False):
    A = torch.randn(M, Kd + pada, device=dev)
    if transB:
        B = torch.randn(Kd, N + padb, device=dev) * 0.1
        ldb = N + padb
    else:
        B = torch.randn(N, Kd + padb, device=dev) * 0.1
        ldb = Kd + padb
    C = torch.empty(M, N, device=dev)
    f = lambda: K.gemm(A, B, C, M, N, Kd, Kd + pada, ldb, N, transB=transB, nosplit=nosplit)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('M=%5d N=%5d K=%5d pad A %2d B %2d nosplit %d transB %d: %.3f ms %6.1f TF' % (M, N, Kd, pada, padb, nosplit, transB, ms, 2.0 * M * N * Kd / ms * 1e-9))
for (M, N, Kd) in [(4096, 4096, 4096), (38400, 4096, 1536), (38400, 512, 2560)]:
    for pada, padb in [(0, 0), (32, 0), (0, 32), (32, 32), (64, 64)]:
        run(M, N, Kd, pada, padb)
run(38400, 4096, 1536, 0, 0, nosplit=1)
run(38400, 1536, 4096, 0, 0, transB=True)
run(38400, 1536, 4096, 32, 0, transB=True)
run(38400, 1536, 4096, 0, 0, nosplit=1, transB=True)
