"""Tile time vs K at fixed M, N (2048 tiles = 8 full rounds of 256 CUs): fit time_per_tile = a + b * (K / 32)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multilingual_text_to_speech_amd import kernels as K
dev = torch.device('cuda')
if '--helper' in sys.argv:       # one workgroup per CU, the way the decoder's helper streams launch their GEMMs
    for (M, N, Kd, tb) in ((38400, 4096, 1536, False), (38400, 1536, 4096, True)):
        A = torch.randn(M, Kd, device=dev); B = (torch.randn(Kd, N, device=dev) if tb else torch.randn(N, Kd, device=dev)) * 0.1; C = torch.empty(M, N, device=dev)
        f = lambda: K.gemm(A, B, C, M, N, Kd, Kd, N if tb else Kd, N, transB=tb, nosplit=1)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('nosplit=1 M=%5d N=%5d K=%5d transB=%d  %.3f ms  %6.1f TF-eq' % (M, N, Kd, tb, ms, 2.0 * M * N * Kd / ms * 1e-9))
    sys.exit(0)
M, N = 8192, 4096
pts = []
for Kd in (96, 288, 544, 1056, 1568, 2080, 3104, 4128):
    A = torch.randn(M, Kd, device=dev); B = torch.randn(N, Kd, device=dev) * 0.1; C = torch.empty(M, N, device=dev)
    f = lambda: K.gemm(A, B, C, M, N, Kd, Kd, Kd, N)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    per_tile = ms * 1e3 / 8
    pts.append((Kd // 32, per_tile))
    print('K=%5d  %.3f ms  %6.1f TF  per tile-round %.1f us' % (Kd, ms, 2.0 * M * N * Kd / ms * 1e-9, per_tile))
n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
print('fit: per tile %.2f us + %.3f us per K block' % (a, b))
