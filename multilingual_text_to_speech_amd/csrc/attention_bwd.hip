// Location-sensitive attention, one decoder step, BACKWARD.
// Reference forward: modules/attention.py:39-45,67-86 (autograd derives the rest in the reference).
//
// Grid (B, nch): workgroup (b, ch) owns rows [l0,l1) of sample b.  The softmax-backward scalar
//   S = sum_l w_l dw_l = sum_l w_l (dalign_l + dcum_l) + <dctx, ctx>
// needs no cross-workgroup reduction because sum_l w_l memory_l IS the saved context.
// PL (= M + bias + loc(cum_in)) is recomputed; ds = de * v * (1 - tanh^2) stays in LDS for the three
// contractions that consume it (dU slab, dcum via the transposed filter bank, dq/dbias/dv).
#include "attention_bwd_body.h"

__global__ __launch_bounds__(ATB_THREADS) void attn_bwd_generic_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, ch = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = ATB_THREADS / 64;
    const int L = p.L, A = p.A, Dm = p.Dm, ksz = p.ksz, pad = (ksz - 1) / 2;
    const int lc = (L + p.nch - 1) / p.nch;
    const int l0 = ch * lc, l1 = min(L, l0 + lc), nl = max(0, l1 - l0);
    float* q = sm;                       // [A]
    float* vv = q + A;                   // [A]
    float* w = vv + A;                   // [L]
    float* dex = w + L;                  // [L]   dalign + dcum_out
    float* cumw = dex + L;               // [L + ksz - 1]
    float* Us = cumw + L + ksz - 1;      // [A*ksz]
    float* dctx_s = Us + A * ksz;        // [Dm]
    float* de = dctx_s + Dm;             // [lc]
    float* ds = de + lc;                 // [lc*A]
    float* dcl = ds + lc * A;            // [lc + ksz - 1]
    float* accq = dcl + lc + ksz - 1;    // [A]
    float* accv = accq + A;              // [A]
    float* red = accv + A;               // [8]

    for (int a = tid; a < A; a += ATB_THREADS) { q[a] = p.q[(long)b * A + a]; vv[a] = p.v[a]; accq[a] = 0.f; accv[a] = 0.f; }
    for (int i = tid; i < A * ksz; i += ATB_THREADS) Us[i] = p.U[i];
    for (int l = tid; l < L; l += ATB_THREADS) {
        w[l] = p.w[(long)b * L + l];
        dex[l] = (p.dalign ? p.dalign[(long)b * L + l] : 0.f) + p.dcum_out[(long)b * L + l];
        cumw[pad + l] = p.cum_in[(long)b * L + l];
    }
    for (int i = tid; i < pad; i += ATB_THREADS) { cumw[i] = 0.f; cumw[pad + L + i] = 0.f; }
    for (int i = tid; i < lc + ksz - 1; i += ATB_THREADS) dcl[i] = 0.f;
    float sdot = 0.f;
    for (int d = tid; d < Dm; d += ATB_THREADS) {
        float g = p.dctx[(long)b * Dm + d];
        for (int k = 0; k < p.n_part; ++k) g += p.part[(long)k * p.part_ks + (long)b * p.part_ld + d];
        dctx_s[d] = g;
        if (ch == 0) p.dctx_total[(long)b * Dm + d] = g;
        sdot += g * p.ctx[(long)b * Dm + d];
    }
    __syncthreads();
    for (int l = tid; l < L; l += ATB_THREADS) sdot += w[l] * dex[l];
    const float S = block_sum(sdot, red, tid);

    // dw for the own rows (wave per row), de = w (dw - S)
    for (int r = wave; r < nl; r += nwaves) {
        const int l = l0 + r;
        const float* mem = p.memory + ((long)b * L + l) * Dm;
        float acc = 0.f;
        for (int d = lane; d < Dm; d += 64) acc += dctx_s[d] * mem[d];
        acc = wave_sum(acc);
        if (lane == 0) de[r] = w[l] * (dex[l] + acc - S);
    }
    __syncthreads();

    // ds over own rows x A; dMt accumulation; dq / dv partial sums
    {
        const float* Mb = p.Mt + (long)b * L * A;
        float* dMb = p.dMt + (long)b * L * A;
        for (int i = tid; i < nl * A; i += ATB_THREADS) {
            const int r = i / A, a = i - r * A, l = l0 + r;
            float s = q[a] + Mb[(long)l * A + a] + p.bias[a];
            const float* u = Us + a * ksz;
            const float* cw = cumw + l;
            for (int j = 0; j < ksz; ++j) s += u[j] * cw[j];
            const float th = tanhf_(s);
            const float dsv = de[r] * vv[a] * (1.f - th * th);
            ds[r * A + a] = dsv;
            dMb[(long)l * A + a] += dsv;
            atomicAdd(&accq[a], dsv);
            atomicAdd(&accv[a], de[r] * th);
        }
    }
    __syncthreads();
    {
        const long slab = (long)b * p.nch + ch;
        for (int a = tid; a < A; a += ATB_THREADS) {
            atomicAdd(p.dq + (long)b * A + a, accq[a]);
            p.dbias_slab[slab * A + a] += accq[a];
            p.dv_slab[slab * A + a] += accv[a];
        }
        // dU[a,j] += sum_l ds[l,a] cum_in[l + j - pad]
        float* dUs = p.dU_slab + slab * A * ksz;
        for (int i = tid; i < A * ksz; i += ATB_THREADS) {
            const int a = i / ksz, j = i - a * ksz;
            float s = 0.f;
            for (int r = 0; r < nl; ++r) s += ds[r * A + a] * cumw[l0 + r + j];
            dUs[i] += s;
        }
        // dcum_in[m] += sum_{a,j} ds[m - j + pad, a] U[a,j]  (local window first)
        for (int i = tid; i < nl * ksz; i += ATB_THREADS) {
            const int r = i / ksz, j = i - r * ksz;
            float g = 0.f;
            for (int a = 0; a < A; ++a) g += ds[r * A + a] * Us[a * ksz + j];
            atomicAdd(&dcl[r + j], g);
        }
    }
    __syncthreads();
    for (int i = tid; i < nl + ksz - 1; i += ATB_THREADS) {
        const int m = l0 - pad + i;
        if (m >= 0 && m < L) atomicAdd(p.dcum_in + (long)b * L + m, dcl[i]);
    }
    for (int r = tid; r < nl; r += ATB_THREADS)     // carry: cum_out = cum_in + w
        atomicAdd(p.dcum_in + (long)b * L + l0 + r, p.dcum_out[(long)b * L + l0 + r]);
}

template <int NP>
__global__ __launch_bounds__(ATB_THREADS) void attn_bwd_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_bwd_body<NP>(p, sm, blockIdx.x, blockIdx.y);
}

MTTS_API int mtts_attn_step_bwd(const AttnBwdArgs* args, void* stream) {
    const AttnBwdArgs& p = *args;
    MTTS_REQUIRE((p.ksz & 1) == 1, "attention kernel size must be odd (got %d)", p.ksz);
    const int lc = (p.L + p.nch - 1) / p.nch;
    const size_t lds = sizeof(float) * ((size_t)5 * p.A + 2 * p.L + (p.L + p.ksz - 1) + (size_t)p.A * p.ksz + p.Dm + lc +
                                        (size_t)lc * p.A + (lc + p.ksz - 1) + 16);
    const size_t lds_fast = attn_bwd_fast_lds(p);
    const bool fast = attn_bwd_fast_ok(p);
    MTTS_REQUIRE(fast || lds <= 64 * 1024, "attn_bwd: LDS request %zu too large (raise nch)", lds);
    if (fast) hipLaunchKernelGGL(attn_bwd_kernel<BNP_MAX>, dim3(p.B, p.nch), dim3(ATB_THREADS), lds_fast, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_bwd_generic_kernel, dim3(p.B, p.nch), dim3(ATB_THREADS), lds, (hipStream_t)stream, p);
    MTTS_CHECK_LAUNCH("attn_bwd_kernel");
    return 0;
}
