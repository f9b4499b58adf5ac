// Location-sensitive attention, one decoder step (forward).
// Reference: modules/attention.py:39-45 (forward), :67-74 (_attent), :76-83 (_normalize), :85-86 (_combine_weights).
//
// Critical path per step: energies e[l] = v . tanh(q + PL[l]) -> masked softmax -> context = w . memory.
// PL = M + bias + loc(cum) depends only on the cumulative alignment, so the workgroups of step t also
// produce PL for step t+1 (31-tap filter bank U = W_loc * W_conv over the LDS-staged cum window).
// Grid (B, nch): every workgroup of a sample recomputes the (cheap) energies/softmax so that the context
// columns and the PL_next rows of that sample can be split over nch CUs without a second launch.
#include "common.h"

constexpr int ATT_THREADS = 512;

__global__ __launch_bounds__(ATT_THREADS) void attn_step_kernel(AttnStepArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, ch = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = ATT_THREADS / 64;
    const int L = p.L, A = p.A, Dm = p.Dm, ksz = p.ksz, pad = (ksz - 1) / 2;
    float* q = sm;                          // [A]
    float* vv = q + A;                      // [A]
    float* w = vv + A;                      // [L]
    float* cumw = w + L;                    // [L + ksz - 1]  cum_out with zero halo
    float* Us = cumw + L + ksz - 1;         // [A * ksz]
    float* part = sm + (((2 * A + 2 * L + ksz - 1 + A * ksz) + 3) & ~3);   // [4 * ATT_THREADS] context partials (16-B aligned)
    const int len = min(p.lengths[b], L);

    for (int a = tid; a < A; a += ATT_THREADS) {
        float s = 0.f;
        for (int k = 0; k < p.kq; ++k) s += p.qpart[(long)k * p.q_ks + (long)b * A + a];
        q[a] = s;
        vv[a] = p.v[a];
        if (ch == 0 && p.q_out) p.q_out[(long)b * A + a] = s;
    }
    if (p.PL_next)
        for (int i = tid; i < A * ksz; i += ATT_THREADS) Us[i] = p.U[i];
    __syncthreads();

    // energies: one wave per position (4 positions in flight per wave), lanes over the attention dimension
    const float* PLb = p.PL + (long)b * L * A;
    for (int l0 = wave; l0 < L; l0 += 4 * nwaves) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + j * nwaves;
            float acc = 0.f;
            if (l < len)
                for (int a = lane; a < A; a += 64) acc += vv[a] * tanhf_(q[a] + PLb[(long)l * A + a]);
            e[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + j * nwaves;
            const float r = wave_sum(e[j]);
            if (lane == 0 && l < L) w[l] = (l < len) ? r : -INFINITY;
        }
    }
    __syncthreads();

    // masked softmax (wave 0) + cumulative alignment window
    if (wave == 0) {
        float mx = -INFINITY;
        for (int l = lane; l < L; l += 64) mx = fmaxf(mx, w[l]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int l = lane; l < L; l += 64) { const float ex = (l < len) ? __expf(w[l] - mx) : 0.f; w[l] = ex; s += ex; }
        s = wave_sum(s);
        const float inv = 1.f / s;
        for (int l = lane; l < L; l += 64) {
            const float wl = w[l] * inv;
            w[l] = wl;
            const float cn = p.cum_in[(long)b * L + l] + wl;
            cumw[pad + l] = cn;
            if (ch == 0) { p.w_out[(long)b * L + l] = wl; p.cum_out[(long)b * L + l] = cn; }
        }
        for (int i = lane; i < pad; i += 64) { cumw[i] = 0.f; cumw[pad + L + i] = 0.f; }
    }
    __syncthreads();

    // context columns [d0, d1) of this chunk: float4 columns x row groups, reduced through LDS
    {
        const int dc = (((Dm + p.nch - 1) / p.nch) + 3) & ~3;
        const int d0 = ch * dc, d1 = min(Dm, d0 + dc);
        const int nc4 = (d1 - d0) >> 2;
        if (nc4 > 0) {
            const int ng = max(1, ATT_THREADS / nc4);
            const int g = tid / nc4, c4 = tid % nc4;
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < ng) {
                const float* mem = p.memory + (long)b * L * Dm + d0 + c4 * 4;
#pragma unroll 4
                for (int l = g; l < len; l += ng) {
                    const float4 m4 = *reinterpret_cast<const float4*>(mem + (long)l * Dm);
                    const float wl = w[l];
                    s4.x += wl * m4.x; s4.y += wl * m4.y; s4.z += wl * m4.z; s4.w += wl * m4.w;
                }
            }
            float4* part4 = reinterpret_cast<float4*>(part);
            part4[tid] = s4;
            __syncthreads();
            if (tid < nc4) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = 0; k < ng; ++k) { const float4 v4 = part4[k * nc4 + tid]; t.x += v4.x; t.y += v4.y; t.z += v4.z; t.w += v4.w; }
                *reinterpret_cast<float4*>(p.ctx_out + (long)b * Dm + d0 + tid * 4) = t;
            }
        }
    }

    // PL for the next step, rows [l0, l1) of this chunk
    if (p.PL_next) {
        const int lc = (L + p.nch - 1) / p.nch;
        const int l0 = ch * lc, l1 = min(L, l0 + lc);
        const float* Mb = p.Mt + (long)b * L * A;
        float* out = p.PL_next + (long)b * L * A;
        for (int i = tid; i < (l1 - l0) * A; i += ATT_THREADS) {
            const int l = l0 + i / A, a = i % A;
            float s = Mb[(long)l * A + a] + p.bias[a];
            const float* u = Us + a * ksz;
            const float* cw = cumw + l;
            for (int j = 0; j < ksz; ++j) s += u[j] * cw[j];
            out[(long)l * A + a] = s;
        }
    }
}

// PL0 = M + bias (cum = 0)
__global__ void attn_pl_init_kernel(const float* __restrict__ Mt, const float* __restrict__ bias, float* __restrict__ PL,
                                    long total, int A) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        PL[i] = Mt[i] + bias[i % A];
}

int attn_pl_init(const float* Mt, const float* bias, float* PL, long total, int A, hipStream_t s) {
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(attn_pl_init_kernel, dim3(blocks), dim3(256), 0, s, Mt, bias, PL, total, A);
    MTTS_CHECK_LAUNCH("attn_pl_init_kernel");
    return 0;
}

int attn_step_launch(const AttnStepArgs& p, hipStream_t s) {
    MTTS_REQUIRE((p.ksz & 1) == 1, "attention kernel size must be odd (got %d)", p.ksz);
    const int dc = (((p.Dm + p.nch - 1) / p.nch) + 3) & ~3;
    MTTS_REQUIRE(dc / 4 <= ATT_THREADS && (p.Dm & 3) == 0, "attn_step: Dm/nch = %d too wide or Dm %% 4 != 0", dc);
    const size_t lds = sizeof(float) * ((((size_t)2 * p.A + 2 * p.L + p.ksz - 1 + (size_t)p.A * p.ksz) + 3 & ~(size_t)3) + 4 * ATT_THREADS);
    MTTS_REQUIRE(lds <= 64 * 1024, "attn_step: LDS request %zu too large", lds);
    hipLaunchKernelGGL(attn_step_kernel, dim3(p.B, p.nch), dim3(ATT_THREADS), lds, s, p);
    MTTS_CHECK_LAUNCH("attn_step_kernel");
    return 0;
}

MTTS_API int mtts_attn_step_fwd(const AttnStepArgs* args, void* stream) {
    return attn_step_launch(*args, (hipStream_t)stream);
}
