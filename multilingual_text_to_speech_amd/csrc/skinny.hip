// Skinny (small-M) fp32 MFMA GEMM launcher; the kernel body lives in skinny_body.h (shared with the fused step launches).
#include <stdlib.h>
#include "skinny_body.h"

template <int MT>
__global__ __launch_bounds__(NT) void skinny_kernel(SkinnyArgs p) {
    __shared__ float red[NW][MT * 16][17];
    skinny_body<MT>(p, red, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Low-register variant (2-deep load pipeline, <= 128 VGPRs): two 512-thread workgroups fit on one CU; used for every launch with
// more than 32 rows (co-residency with the helper streams' GEMM workgroups beats pipeline depth).
template <int MT>
__global__ __launch_bounds__(NT, 4) void skinny_kernel_lo(SkinnyArgs p) {
    __shared__ float red[NW][MT * 16][17];
    skinny_body<MT, 2>(p, red, blockIdx.x, blockIdx.y, blockIdx.z);
}

int skinny_launch(const SkinnyArgs& p, hipStream_t s) {
    MTTS_REQUIRE(p.nseg >= 0 && p.nseg <= 3 && p.B > 0 && (p.nseg > 0 || p.lstm == 2), "skinny: bad nseg/B");
    for (int i = 0; i < p.nseg; ++i) {
        MTTS_REQUIRE((p.seg[i].K & 3) == 0 && (p.seg[i].ldx & 3) == 0 && (p.seg[i].ldw & 3) == 0 && p.seg[i].K > 0,
                     "skinny: segment %d needs K, ldx, ldw multiples of 4 (K=%d ldx=%d ldw=%d)", i, p.seg[i].K,
                     p.seg[i].ldx, p.seg[i].ldw);
        MTTS_REQUIRE(((uintptr_t)p.seg[i].x & 15) == 0 && ((uintptr_t)p.seg[i].w & 15) == 0,
                     "skinny: segment %d pointers must be 16-byte aligned", i);
    }
    MTTS_REQUIRE(p.n_part <= MAX_PART, "skinny: at most %d partial slabs", MAX_PART);
    const int ks = p.ksplit < 1 ? 1 : p.ksplit;
    SkinnyArgs q = p; q.ksplit = ks;
    if (p.nseg == 0) { q.seg[0] = SkSeg{p.gates, p.gates, 0, 0, 0, 0, 0}; q.nseg = 1; }   // pure pointwise: empty K range
    for (int i = q.nseg; i < 3; ++i) q.seg[i] = q.seg[0];      // keep the unused selectors dereferenceable
    if (p.lstm == 2) q.N = p.H;
    if (p.lstm == 1 && !q.h_prev) q.h_prev = q.c_prev;
    const int cbs = p.lstm == 1 ? cdiv(p.H, 4) : cdiv(q.N, 16);
    // LSTM cell backward (K <= the query width): the launch is all epilogue operands (16 loads per (row, unit)); one 16-row tile per
    // workgroup gives four times the workgroups to fetch them (bit-identical: same K chunks per wave, same reduction order).
    if (p.lstm == 2 && p.B > 16 && p.B <= 64 && q.seg[0].K <= 256 && q.nseg == 1 && ks == 1) {
        hipLaunchKernelGGL(skinny_kernel<1>, dim3(cbs, cdiv(p.B, 16), 1), dim3(NT), 0, s, q);
        MTTS_CHECK_LAUNCH("skinny_kernel");
        return 0;
    }
    if (p.B <= 16) hipLaunchKernelGGL(skinny_kernel<1>, dim3(cbs, cdiv(p.B, 16), ks), dim3(NT), 0, s, q);
    else if (p.B <= 32) hipLaunchKernelGGL(skinny_kernel<2>, dim3(cbs, cdiv(p.B, 32), ks), dim3(NT), 0, s, q);
    // more than 32 rows: the <= 128-VGPR variant.  A 4-deep load pipeline (153 VGPRs) is ~10 % faster alone, but cannot share a CU with
    // two GEMM workgroups of the helper streams and then WAITS for them: 92.0-92.3 vs 94.4-94.8 ms per train step (round 2).
    else hipLaunchKernelGGL(skinny_kernel_lo<4>, dim3(cbs, cdiv(p.B, 64), ks), dim3(NT), 0, s, q);
    MTTS_CHECK_LAUNCH("skinny_kernel");
    return 0;
}

MTTS_API int mtts_skinny_gemm(const SkinnyArgs* args, void* stream) { return skinny_launch(*args, (hipStream_t)stream); }
