"""Per-launch time of the skinny step kernels at the shapes the train step / the synthesis loop launch them with (back-to-back
launches through the C-ABI entry mtts_skinny_gemm, HIP events; operands L2-warm).  Used for same-box A/Bs of library builds
(scripts/ab_lib.sh): python scripts/bench_skinny.py [--reps 300]"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multilingual_text_to_speech_amd import _C                                    # noqa: E402
from multilingual_text_to_speech_amd._C import check, lib, ptr, stream_ptr        # noqa: E402

dev = 'cuda'
R = lambda *s: torch.randn(*s, device=dev) * 0.1
keep = []


def seg(a, i, x, w, K, ldx, ldw, xp=0, wp=0):
    keep.extend([x, w])
    a.seg[i].x, a.seg[i].w, a.seg[i].K, a.seg[i].ldx, a.seg[i].ldw, a.seg[i].xpack, a.seg[i].wpack = ptr(x), ptr(w), K, ldx, ldw, xp, wp


def product(B, N, K, ks, packed):
    a = _C.SkinnyArgs(); a.nseg, a.B, a.N, a.ksplit = 1, B, N, ks
    Bp = (B + 15) & ~15
    seg(a, 0, R(Bp, K), R(((N + 15) & ~15), K), K, K, K, int(packed), int(packed))
    out = R(ks, B, N); keep.append(out)
    a.out, a.ldo, a.out_ks = ptr(out), N, B * N if ks > 1 else 0
    return a


def cell_bwd(B, H, Kq, n_part, lengths=False):
    a = _C.SkinnyArgs(); a.B, a.H, a.lstm, a.ksplit, a.N = B, H, 2, 1, H
    if Kq:
        a.nseg = 1; seg(a, 0, R(B, Kq), R(H, Kq), Kq, Kq, Kq)
    bufs = dict(dh_a=R(B, H), part=R(max(n_part, 1), B, H), gates=torch.rand(B, 4 * H, device=dev), c_prev=R(B, H), dc_in=R(B, H),
                dc_out=R(B, H), dgates_out=R(B, 4 * H), dg_pack_out=R(((B + 15) & ~15) * 4 * H), dh_b=R(B, H), dh_carry_out=R(B, H))
    keep.append(bufs)
    a.dh_a, a.ld_dh_a = ptr(bufs['dh_a']), H
    if n_part:
        a.part, a.n_part, a.part_ks, a.part_ld, a.part_col0 = ptr(bufs['part']), n_part, B * H, H, 0
    a.gates, a.c_prev, a.dc_in, a.dc_out = ptr(bufs['gates']), ptr(bufs['c_prev']), ptr(bufs['dc_in']), ptr(bufs['dc_out'])
    a.dgates_out, a.ld_dgates = ptr(bufs['dgates_out']), 4 * H
    if lengths:
        ln = torch.randint(1, 100, (B,), dtype=torch.int32, device=dev); keep.append(ln)
        a.lengths, a.t, a.dh_b, a.dh_carry_out = ptr(ln), 50, ptr(bufs['dh_b']), ptr(bufs['dh_carry_out'])
    else:
        a.dg_pack_out = ptr(bufs['dg_pack_out'])
    return a


def cell_fwd(B, H, K):
    a = _C.SkinnyArgs(); a.B, a.H, a.lstm, a.ksplit, a.N, a.nseg = B, H, 1, 1, 4 * H, 1
    seg(a, 0, R(B, K), R(4 * H, K), K, K, K)
    bufs = dict(pre=R(B, 4 * H), b=R(4 * H), h=R(B, H), c=R(B, H), ho=R(B, H), co=R(B, H), go=R(B, 4 * H), y=R(B, 2 * H))
    keep.append(bufs)
    ln = torch.randint(1, 100, (B,), dtype=torch.int32, device=dev); keep.append(ln)
    a.pre, a.ldpre, a.b_ih, a.b_hh = ptr(bufs['pre']), 4 * H, ptr(bufs['b']), ptr(bufs['b'])
    a.h_prev, a.c_prev, a.h_out, a.c_out, a.gates_out = ptr(bufs['h']), ptr(bufs['c']), ptr(bufs['ho']), ptr(bufs['co']), ptr(bufs['go'])
    a.lengths, a.t, a.y_out, a.ldy = ptr(ln), 50, ptr(bufs['y']), 2 * H
    return a


def proj(B, Ks, N, ks):
    a = _C.SkinnyArgs(); a.nseg, a.B, a.N, a.ksplit = len(Ks), B, N, ks
    Kt = sum(Ks); w = R(N, Kt); k0 = 0
    for i, K in enumerate(Ks):
        seg(a, i, R(B, K), w, K, K, Kt); a.seg[i].w = w.data_ptr() + 4 * k0; k0 += K
    out = R(ks, B, 84); bias = R(N); keep.extend([out, bias])
    a.out, a.ldo = ptr(out), 84
    if ks > 1:
        a.out_ks = B * 84
    else:
        a.bias = ptr(bias)
    return a


def timeit(a, reps):
    f = lib().mtts_skinny_gemm
    s = stream_ptr()
    for _ in range(20):
        check(f(ctypes.byref(a), s), 'skinny')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            f(ctypes.byref(a), s)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--reps', type=int, default=300); args = ap.parse_args()
    cases = [
        ('chain A cell bwd  B64 H1024 Kq128 parts4 (skinny<1> 64x4)', cell_bwd(64, 1024, 128, 4)),
        ('chain B cell bwd  B64 H1024 no product parts4', cell_bwd(64, 1024, 0, 4)),
        ('h-cols   B64 N1024 K4096 ks4 packed (lo<4> 64x1x4)', product(64, 1024, 4096, 4, True)),
        ('ctx-cols B64 N544  K4096 ks7 packed (lo<4> 34x1x7)', product(64, 544, 4096, 7, True)),
        ('BiLSTM fwd step  B64 H256 K256 row-major + lengths', cell_fwd(64, 256, 256)),
        ('BiLSTM bwd prod  B64 N256 K1024 ks4 row-major', product(64, 256, 1024, 4, False)),
        ('BiLSTM cell bwd  B64 H256 parts4 + lengths', cell_bwd(64, 256, 0, 4, True)),
        ('projection B128 N81 K1024+288 ks1', proj(128, (1024, 288), 81, 1)),
        ('projection B128 N81 K1024+288 ks8 (slabs)', proj(128, (1024, 288), 81, 8)),
        ('projection B1   N81 K1024+288 ks1', proj(1, (1024, 288), 81, 1)),
        ('cell bwd B240 H1024 Kq128 parts4 (lo<4>)', cell_bwd(240, 1024, 128, 4)),
    ]
    for name, a in cases:
        print(f'{name:62s} {timeit(a, args.reps):7.2f} us')


if __name__ == '__main__':
    main()
