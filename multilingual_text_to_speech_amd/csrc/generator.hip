// Parameter generator of the 'generated' encoder (K2): convolution kernels emitted from the language embedding.
// Reference: Conv1dGenerated.forward modules/generated.py:34-42 -
//     kernel = Linear(bottleneck -> (O/G)(I/G)k)(hidden)            hidden [G, bott]  (bott = 4 / 8: a bandwidth-bound GEMV)
//     kernel.view(O, I/G, k)  ->  F.conv1d(groups = G)
// Round 1 pushed [G, bott] x [393 216, bott]^T through the 128x128 MFMA GEMM (>= 92 % of every tile idle), viewed the result
// as [O, I/G, k] and repacked it to the implicit-GEMM layout [O, k, I/G] with a second kernel - and the same again, backwards,
// for the gradient.  Here ONE kernel reads the generator weight once and writes the conv's packed layout directly, and ONE
// kernel turns the conv's packed weight gradient into dW_kernel, db_kernel and the hidden-row gradient.
//   flat row of group g, element j = (o' Cg + c) k + t   <->   packed wp[(g Og + o'), t, c]          (Og = O/G, Cg = I/G)
// Thread (o', c) owns the k taps of one input channel: its generator rows are contiguous (k x bott floats), consecutive
// threads are consecutive c, so both the generator read and the packed write are coalesced.
#include "common.h"

constexpr int GP_MAXG = 16, GP_MAXB = 8, GP_MAXK = 8;

struct GenArgs {
    const float* hid;      // [G, bott]
    const float* wk;       // [Og*Cg*k, bott]
    const float* bk;       // [Og*Cg*k] or NULL
    float* wp;             // fwd out: [G*Og, k, Cg]
    const float* dwp;      // bwd in:  [G*Og, k, Cg]
    float* dwk;            // bwd out: [Og*Cg*k, bott]
    float* dbk;            // bwd out: [Og*Cg*k] or NULL
    float* dhid_slab;      // bwd out: [gridDim.x*gridDim.y][G*bott] per-workgroup partial sums
    int G, bott, Og, Cg, k;
};

__global__ __launch_bounds__(256) void gen_params_fwd_kernel(GenArgs p) {
    __shared__ float hs[GP_MAXG * GP_MAXB];
    const int tid = threadIdx.x;
    if (tid < p.G * p.bott) hs[tid] = p.hid[tid];
    __syncthreads();
    const int o = blockIdx.y, c = blockIdx.x * 256 + tid;
    if (c >= p.Cg) return;
    const long j0 = ((long)o * p.Cg + c) * p.k;
    for (int t = 0; t < p.k; ++t) {
        float w[GP_MAXB];
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b) w[b] = b < p.bott ? p.wk[(j0 + t) * p.bott + b] : 0.f;
        const float bias = p.bk ? p.bk[j0 + t] : 0.f;
        for (int g = 0; g < p.G; ++g) {
            float v = bias;
#pragma unroll
            for (int b = 0; b < GP_MAXB; ++b) v += hs[g * p.bott + (b < p.bott ? b : 0)] * w[b];
            p.wp[(((long)g * p.Og + o) * p.k + t) * p.Cg + c] = v;
        }
    }
}

__global__ __launch_bounds__(256) void gen_params_bwd_kernel(GenArgs p) {
    __shared__ float hs[GP_MAXG * GP_MAXB];
    __shared__ float red[4][GP_MAXG * GP_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int GB = p.G * p.bott;
    if (tid < GB) hs[tid] = p.hid[tid];
    __syncthreads();
    const int o = blockIdx.y, c = blockIdx.x * 256 + tid;
    const bool ok = c < p.Cg;
    const long j0 = ((long)o * p.Cg + (ok ? c : 0)) * p.k;
    // dhid[g][b] partial of this thread: visited in (g, b) order below, reduced across the workgroup through shuffles + LDS
    for (int g0 = 0; g0 < p.G; ++g0) {          // outer loop over g keeps the register footprint at bott accumulators
        float dh[GP_MAXB];
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b) dh[b] = 0.f;
        for (int t = 0; t < p.k; ++t) {
            const float d = ok ? p.dwp[(((long)g0 * p.Og + o) * p.k + t) * p.Cg + c] : 0.f;
#pragma unroll
            for (int b = 0; b < GP_MAXB; ++b) dh[b] += (b < p.bott) ? d * p.wk[(j0 + t) * p.bott + b] : 0.f;
        }
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b) {
            const float s = wave_sum(dh[b]);
            if (lane == 0 && b < p.bott) red[wave][g0 * p.bott + b] = s;
        }
    }
    // generator weight / bias gradients: one pass over g per tap
    if (ok) {
        for (int t = 0; t < p.k; ++t) {
            float dw[GP_MAXB], db = 0.f;
#pragma unroll
            for (int b = 0; b < GP_MAXB; ++b) dw[b] = 0.f;
            for (int g = 0; g < p.G; ++g) {
                const float d = p.dwp[(((long)g * p.Og + o) * p.k + t) * p.Cg + c];
                db += d;
#pragma unroll
                for (int b = 0; b < GP_MAXB; ++b) dw[b] += d * hs[g * p.bott + (b < p.bott ? b : 0)];
            }
#pragma unroll
            for (int b = 0; b < GP_MAXB; ++b) if (b < p.bott) p.dwk[(j0 + t) * p.bott + b] = dw[b];
            if (p.dbk) p.dbk[j0 + t] = db;
        }
    }
    __syncthreads();
    if (tid < GB) {
        const long slab = (long)blockIdx.y * gridDim.x + blockIdx.x;
        p.dhid_slab[slab * GB + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
}

static int gen_check(const GenParamsArgs& a) {
    MTTS_REQUIRE(a.G >= 1 && a.G <= GP_MAXG && a.bott >= 1 && a.bott <= GP_MAXB && a.k >= 1 && a.k <= GP_MAXK && a.Og > 0 && a.Cg > 0,
                 "mtts_gen_params: need G <= %d, bottleneck <= %d, kernel size <= %d (G=%d bott=%d k=%d)", GP_MAXG, GP_MAXB, GP_MAXK, a.G,
                 a.bott, a.k);
    MTTS_REQUIRE(a.Og <= 65535, "mtts_gen_params: O/G = %d too large for the grid", a.Og);
    return 0;
}

static GenArgs gen_args(const GenParamsArgs& a) {
    GenArgs p; memset(&p, 0, sizeof(p));
    p.hid = a.hidden; p.wk = a.w_kernel; p.bk = a.b_kernel; p.wp = a.w_packed; p.dwp = a.d_w_packed; p.dwk = a.d_w_kernel; p.dbk = a.d_b_kernel;
    p.dhid_slab = a.d_hidden_slab; p.G = a.G; p.bott = a.bott; p.Og = a.Og; p.Cg = a.Cg; p.k = a.k;
    return p;
}

MTTS_API long mtts_gen_params_slabs(int Og, int Cg) { return (long)Og * ((Cg + 255) / 256); }

MTTS_API int mtts_gen_params_fwd(const GenParamsArgs* args, void* stream) {
    MTTS_TRY(gen_check(*args));
    MTTS_REQUIRE(args->hidden && args->w_kernel && args->w_packed, "mtts_gen_params_fwd: missing buffers");
    hipLaunchKernelGGL(gen_params_fwd_kernel, dim3((args->Cg + 255) / 256, args->Og), dim3(256), 0, (hipStream_t)stream, gen_args(*args));
    MTTS_CHECK_LAUNCH("gen_params_fwd_kernel");
    return 0;
}

MTTS_API int mtts_gen_params_bwd(const GenParamsArgs* args, void* stream) {
    MTTS_TRY(gen_check(*args));
    MTTS_REQUIRE(args->hidden && args->w_kernel && args->d_w_packed && args->d_w_kernel && args->d_hidden_slab, "mtts_gen_params_bwd: missing buffers");
    hipLaunchKernelGGL(gen_params_bwd_kernel, dim3((args->Cg + 255) / 256, args->Og), dim3(256), 0, (hipStream_t)stream, gen_args(*args));
    MTTS_CHECK_LAUNCH("gen_params_bwd_kernel");
    return 0;
}
