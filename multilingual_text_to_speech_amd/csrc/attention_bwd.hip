// Location-sensitive attention, one decoder step, BACKWARD.
// Reference forward: modules/attention.py:39-45,67-86 (autograd derives the rest in the reference).
//
// Grid (B, nch): workgroup (b, ch) owns rows [l0,l1) of sample b.  The softmax-backward scalar
//   S = sum_l w_l dw_l = sum_l w_l (dalign_l + dcum_l) + <dctx, ctx>
// needs no cross-workgroup reduction because sum_l w_l memory_l IS the saved context.
// PL (= M + bias + loc(cum_in)) is recomputed; ds = de * v * (1 - tanh^2) stays in LDS for the three
// contractions that consume it (dU slab, dcum via the transposed filter bank, dq/dbias/dv).
#include "common.h"

constexpr int ATB_THREADS = 512;

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < ATB_THREADS / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}

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

// ------------------------------------------------------------------------------------------------------------
// fast kernel: every global operand is requested at entry (see attention.hip for the rationale)
// ------------------------------------------------------------------------------------------------------------
constexpr int BNM_MAX = 8;    // own (row, a) elements per thread
constexpr int BNU_MAX = 8;    // filter-bank elements per thread
constexpr int BNR_MAX = 4;    // own rows per wave
constexpr int BND_MAX = 9;    // memory floats per lane per row (Dm <= 576)
constexpr int BNP_MAX = 8;    // partial slabs
constexpr int BNX_MAX = 2;    // context floats per thread (Dm <= 1024)

__global__ __launch_bounds__(ATB_THREADS) void attn_bwd_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, ch = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = ATB_THREADS / 64;
    const int L = p.L, A = p.A, Dm = p.Dm, ksz = p.ksz, pad = (ksz - 1) / 2;
    const int lc = (L + p.nch - 1) / p.nch;
    const int l0 = ch * lc, l1 = min(L, l0 + lc), nl = max(0, l1 - l0);
    const int nlA = nl * A, AK = A * ksz;
    float* q = sm;                       // [A]
    float* vv = q + A;                   // [A]
    float* bias = vv + A;                // [A]
    float* w = bias + A;                 // [L]
    float* dex = w + L;                  // [L]   dalign + dcum_out
    float* cumw = dex + L;               // [L + ksz - 1]
    float* Us = cumw + L + ksz - 1;      // [A*ksz]
    float* dctx_s = Us + AK;             // [Dm]
    float* de = dctx_s + Dm;             // [lc]
    float* ds = de + lc;                 // [lc*A]
    float* dcl = ds + lc * A;            // [lc + ksz - 1]
    float* accq = dcl + lc + ksz - 1;    // [A]
    float* accv = accq + A;              // [A]
    float* red = accv + A;               // [8]
    const long slab = (long)b * p.nch + ch;

    // ---- burst of independent loads
    const int ac = min(tid, A - 1), lcl = min(tid, L - 1);
    const float q_r = p.q[(long)b * A + ac], v_r = p.v[ac], bias_r = p.bias[ac];
    const float dvs_r = p.dv_slab[slab * A + ac], dbs_r = p.dbias_slab[slab * A + ac];
    const float w_r = p.w[(long)b * L + lcl], cum_r = p.cum_in[(long)b * L + lcl];
    const float dco_r = p.dcum_out[(long)b * L + lcl];
    const float dal_r = p.dalign ? p.dalign[(long)b * L + lcl] : 0.f;
    float dcx[BNX_MAX], cx[BNX_MAX];
#pragma unroll
    for (int j = 0; j < BNX_MAX; ++j) {
        const int d = min(tid + j * ATB_THREADS, Dm - 1);
        float g = p.dctx[(long)b * Dm + d];
        float pp[BNP_MAX];
#pragma unroll
        for (int k = 0; k < BNP_MAX; ++k) pp[k] = (k < p.n_part) ? p.part[(long)k * p.part_ks + (long)b * p.part_ld + d] : 0.f;
#pragma unroll
        for (int k = 0; k < BNP_MAX; ++k) g += pp[k];
        dcx[j] = g;
        cx[j] = p.ctx[(long)b * Dm + d];
    }
    float memr[BNR_MAX][BND_MAX];
#pragma unroll
    for (int j = 0; j < BNR_MAX; ++j) {
        const int l = min(l0 + wave + j * nwaves, L - 1);
        const float* mem = p.memory + ((long)b * L + l) * Dm;
#pragma unroll
        for (int k = 0; k < BND_MAX; ++k) memr[j][k] = mem[min(lane + 64 * k, Dm - 1)];
    }
    float mt[BNM_MAX], dmt[BNM_MAX], us[BNU_MAX], dus[BNU_MAX];
    {
        const float* Mb = p.Mt + ((long)b * L + l0) * A;
        const float* dMb = p.dMt + ((long)b * L + l0) * A;
#pragma unroll
        for (int j = 0; j < BNM_MAX; ++j) { const int i = min(tid + j * ATB_THREADS, max(nlA - 1, 0)); mt[j] = Mb[i]; dmt[j] = dMb[i]; }
        const float* dUs = p.dU_slab + slab * AK;
#pragma unroll
        for (int j = 0; j < BNU_MAX; ++j) { const int i = min(tid + j * ATB_THREADS, AK - 1); us[j] = p.U[i]; dus[j] = dUs[i]; }
    }

    // ---- stage in LDS
    if (tid < A) { q[tid] = q_r; vv[tid] = v_r; bias[tid] = bias_r; accq[tid] = 0.f; accv[tid] = 0.f; }
    if (tid < L) { w[tid] = w_r; dex[tid] = dal_r + dco_r; cumw[pad + tid] = cum_r; }
    if (tid < pad) { cumw[tid] = 0.f; cumw[pad + L + tid] = 0.f; }
    for (int i = tid; i < lc + ksz - 1; i += ATB_THREADS) dcl[i] = 0.f;
#pragma unroll
    for (int j = 0; j < BNU_MAX; ++j) { const int i = tid + j * ATB_THREADS; if (i < AK) Us[i] = us[j]; }
    float sdot = 0.f;
#pragma unroll
    for (int j = 0; j < BNX_MAX; ++j) {
        const int d = tid + j * ATB_THREADS;
        if (d < Dm) {
            dctx_s[d] = dcx[j];
            if (ch == 0) p.dctx_total[(long)b * Dm + d] = dcx[j];
            sdot += dcx[j] * cx[j];
        }
    }
    if (tid < L) sdot += w_r * (dal_r + dco_r);
    __syncthreads();
    const float S = block_sum(sdot, red, tid);

    // ---- dw for the own rows (wave per row, memory rows already in registers), de = w (dw - S)
#pragma unroll
    for (int j = 0; j < BNR_MAX; ++j) {
        const int r = wave + j * nwaves;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < BND_MAX; ++k) { const int d = lane + 64 * k; acc += (d < Dm) ? dctx_s[d] * memr[j][k] : 0.f; }
        acc = wave_sum(acc);
        if (lane == 0 && r < nl) de[r] = w[l0 + r] * (dex[l0 + r] + acc - S);
    }
    __syncthreads();

    // ---- ds over own rows x A; dMt accumulation; dq / dv partial sums
    {
        float* dMb = p.dMt + ((long)b * L + l0) * A;
#pragma unroll
        for (int j = 0; j < BNM_MAX; ++j) {
            const int i = tid + j * ATB_THREADS;
            if (i < nlA) {
                const int r = i / A, a = i - r * A;
                float sacc = q[a] + mt[j] + bias[a];
                const float* u = Us + a * ksz;
                const float* cw = cumw + l0 + r;
                for (int jj = 0; jj < ksz; ++jj) sacc += u[jj] * cw[jj];
                const float th = tanhf_(sacc);
                const float dsv = de[r] * vv[a] * (1.f - th * th);
                ds[i] = dsv;
                dMb[i] = dmt[j] + dsv;
                atomicAdd(&accq[a], dsv);
                atomicAdd(&accv[a], de[r] * th);
            }
        }
    }
    __syncthreads();
    if (tid < A) {
        atomicAdd(p.dq + (long)b * A + tid, accq[tid]);
        p.dbias_slab[slab * A + tid] = dbs_r + accq[tid];
        p.dv_slab[slab * A + tid] = dvs_r + accv[tid];
    }
    {
        float* dUs = p.dU_slab + slab * AK;
#pragma unroll
        for (int j = 0; j < BNU_MAX; ++j) {           // dU[a,jj] += sum_r ds[r,a] cum_in[l0 + r + jj - pad]
            const int i = tid + j * ATB_THREADS;
            if (i < AK) {
                const int a = i / ksz, jj = i - a * ksz;
                float sacc = 0.f;
                for (int r = 0; r < nl; ++r) sacc += ds[r * A + a] * cumw[l0 + r + jj];
                dUs[i] = dus[j] + sacc;
            }
        }
        for (int i = tid; i < nl * ksz; i += ATB_THREADS) {   // dcum window: sum_a ds[r,a] U[a,jj]
            const int r = i / ksz, jj = i - r * ksz;
            float g = 0.f;
            for (int a = 0; a < A; ++a) g += ds[r * A + a] * Us[a * ksz + jj];
            atomicAdd(&dcl[r + jj], g);
        }
    }
    __syncthreads();
    for (int i = tid; i < nl + ksz - 1; i += ATB_THREADS) {
        const int m = l0 - pad + i;
        if (m >= 0 && m < L) atomicAdd(p.dcum_in + (long)b * L + m, dcl[i]);
    }
    if (tid >= l0 && tid < l1) atomicAdd(p.dcum_in + (long)b * L + tid, dco_r);     // carry: cum_out = cum_in + w
}

MTTS_API int mtts_attn_step_bwd(const AttnBwdArgs* args, void* stream) {
    const AttnBwdArgs& p = *args;
    MTTS_REQUIRE((p.ksz & 1) == 1, "attention kernel size must be odd (got %d)", p.ksz);
    const int lc = (p.L + p.nch - 1) / p.nch;
    const size_t lds = sizeof(float) * ((size_t)5 * p.A + 2 * p.L + (p.L + p.ksz - 1) + (size_t)p.A * p.ksz + p.Dm + lc +
                                        (size_t)lc * p.A + (lc + p.ksz - 1) + 16);
    MTTS_REQUIRE(lds <= 64 * 1024, "attn_bwd: LDS request %zu too large (raise nch)", lds);
    const bool fast = p.A <= ATB_THREADS && p.L <= ATB_THREADS && (long)lc * p.A <= (long)BNM_MAX * ATB_THREADS &&
                      (long)p.A * p.ksz <= (long)BNU_MAX * ATB_THREADS && lc <= BNR_MAX * (ATB_THREADS / 64) &&
                      p.Dm <= 64 * BND_MAX && p.Dm <= BNX_MAX * ATB_THREADS && p.n_part <= BNP_MAX &&
                      (p.ksz - 1) / 2 <= ATB_THREADS;
    if (fast) hipLaunchKernelGGL(attn_bwd_kernel, dim3(p.B, p.nch), dim3(ATB_THREADS), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_bwd_generic_kernel, dim3(p.B, p.nch), dim3(ATB_THREADS), lds, (hipStream_t)stream, p);
    MTTS_CHECK_LAUNCH("attn_bwd_kernel");
    return 0;
}
