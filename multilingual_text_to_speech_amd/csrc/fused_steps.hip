// "Fat" step launches: two INDEPENDENT pieces of one decoder step share a single kernel launch so that they overlap on
// the GPU without a second stream (a cross-stream event edge costs ~10 us per step on this platform, a dependent launch ~2.6 us).
//
//   attn_bwd_plus_skinny:  attention backward of step t   ||   dG_att(t+1) x W_hh^T   (input gradient w.r.t. h_att_t, which is
//                          only needed by the cell backward that follows the attention backward)
// Workgroups [0, B*nch) run the attention body, the rest run the skinny-GEMM body; both use 512 threads.
#include <stdlib.h>
#include "attention_bwd_body.h"
#include "skinny_body.h"

// launch bound 4 waves/SIMD (<= 128 VGPRs): an attention workgroup and a GEMM workgroup must fit on one CU together.
// PK: operands of the product in MFMA tile order (skinny_body's packed-only instantiation); the product is a plain one (PLAIN).
template <int PK>
__global__ __launch_bounds__(NT, 4) void attn_bwd_plus_skinny_kernel(AttnBwdArgs a, SkinnyArgs k, int n_attn, int sk_cbs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int id = blockIdx.x;
    if (id < n_attn) {
        attn_bwd_body(a, sm, id / a.nch, id % a.nch);
    } else {
        const int j = id - n_attn;
        float (&red)[NW][64][17] = *reinterpret_cast<float (*)[NW][64][17]>(sm);
        skinny_body<4, 2, PK, 1>(k, red, j % sk_cbs, 0, j / sk_cbs);
    }
}

// Falls back to two separate launches when the fused preconditions do not hold.
int attn_bwd_plus_skinny(const AttnBwdArgs& a, const SkinnyArgs& k, hipStream_t s) {
    const bool fusable = attn_bwd_fast_ok(a) && k.lstm == 0 && k.B > 32 && k.B <= 64 && k.nseg == 1 && k.ksplit >= 1;
    if (!fusable) {
        MTTS_TRY(mtts_attn_step_bwd(&a, s));
        return skinny_launch(k, s);
    }
    SkinnyArgs q = k;
    for (int i = q.nseg; i < 3; ++i) q.seg[i] = q.seg[0];
    const int cbs = cdiv(q.N, 16);
    const int n_attn = a.B * a.nch;
    size_t lds = attn_bwd_fast_lds(a);
    const size_t lds_sk = sizeof(float) * NW * 64 * 17;
    if (lds_sk > lds) lds = lds_sk;
    const dim3 grid(n_attn + cbs * q.ksplit);
    if (q.seg[0].xpack == 2 && q.seg[0].wpack == 2) hipLaunchKernelGGL(attn_bwd_plus_skinny_kernel<3>, grid, dim3(NT), lds, s, a, q, n_attn, cbs);
    else if (q.seg[0].xpack && q.seg[0].wpack) hipLaunchKernelGGL(attn_bwd_plus_skinny_kernel<1>, grid, dim3(NT), lds, s, a, q, n_attn, cbs);
    else if (!q.seg[0].xpack && !q.seg[0].wpack) hipLaunchKernelGGL(attn_bwd_plus_skinny_kernel<2>, grid, dim3(NT), lds, s, a, q, n_attn, cbs);
    else hipLaunchKernelGGL(attn_bwd_plus_skinny_kernel<0>, grid, dim3(NT), lds, s, a, q, n_attn, cbs);
    MTTS_CHECK_LAUNCH("attn_bwd_plus_skinny_kernel");
    return 0;
}
