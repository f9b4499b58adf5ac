// BatchNorm (+activation +dropout +highway gate) over channel-last activations x[R, C], fwd and bwd.
//
// Reference semantics: modules/layers.py:50-86 (ConvBlock), :134-178 (highway variants),
// modules/generated.py:71-96 (generated affine, eps 1e-8).  Statistics run over ALL rows (padding
// included).  Two-pass statistics (mean, then centred second moment) keep the
// eps=1e-8 generated-BN accurate in fp32.
//
// Work split: column c is owned by lane (c % 64) of a workgroup column-block; rows are split over
// grid.y chunks whose partial sums land in a [chunks, C] workspace; every consumer re-reduces the
// <= NCHUNK_MAX partials itself (no atomics -> deterministic, no extra finalize launch).
#include "common.h"

constexpr int NCHUNK_MAX = 128;
// Rows a thread requests before it touches the first value (statistics passes 8, backward passes 4 x (x, dy, keep flag)): written
// row by row every iteration of these column loops was one memory round trip for 4-5 requests per thread - the round-4 trace had
// the backward apply pass at 136 us and the backward reduction at 89 us for 253 / 175 MB (floors 63 / 44 us at the copy rate).
constexpr int BN_U = 8;
constexpr int BN_UB = 4;


__device__ __forceinline__ int bn_chunks(int R) {
    int n = (R + 63) / 64;
    return n < 1 ? 1 : (n > NCHUNK_MAX ? NCHUNK_MAX : n);
}

// pass 1: partial column sums.  grid (ceil(C/64), chunks), block (64, 4)
__global__ void bn_sum_kernel(BnArgs p) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int nch = gridDim.y;
    const int rows_per = (p.R + nch - 1) / nch;
    const int r0 = blockIdx.y * rows_per, r1 = min(p.R, r0 + rows_per);
    float s = 0.f;
    if (c < p.C)
        for (int r = r0 + threadIdx.y; r < r1; r += 4 * BN_U) {      // BN_U rows REQUESTED per round trip, added in row order
            float v[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) v[u] = p.x[(long)min(r + 4 * u, r1 - 1) * p.C + c];
#pragma unroll
            for (int u = 0; u < BN_U; ++u)
                if (r + 4 * u < r1) s += v[u];
        }
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < p.C)
        p.ws[(long)blockIdx.y * p.C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// pass 2: partial centred sums of squares (mean recomputed from pass-1 partials).
__global__ void bn_var_kernel(BnArgs p) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int nch = gridDim.y;
    const int rows_per = (p.R + nch - 1) / nch;
    const int r0 = blockIdx.y * rows_per, r1 = min(p.R, r0 + rows_per);
    float s = 0.f;
    if (c < p.C) {
        float m = 0.f;
        for (int k0 = 0; k0 < nch; k0 += 16) {      // sixteen slabs per memory round trip, added in slab order (see bn_finalize_kernel)
            float pm[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) pm[j] = p.ws[(long)min(k0 + j, nch - 1) * p.C + c];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (k0 + j < nch) m += pm[j];
        }
        m /= (float)p.R;
        for (int r = r0 + threadIdx.y; r < r1; r += 4 * BN_U) {
            float v[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) v[u] = p.x[(long)min(r + 4 * u, r1 - 1) * p.C + c];
#pragma unroll
            for (int u = 0; u < BN_U; ++u)
                if (r + 4 * u < r1) { const float d = v[u] - m; s += d * d; }
        }
    }
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < p.C)
        p.ws[(long)(NCHUNK_MAX + blockIdx.y) * p.C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// finalize: mean / rstd / running stats.  grid ceil(C/256)
__global__ void bn_finalize_kernel(BnArgs p, int nch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    float m = 0.f, v = 0.f;
    // the partial slabs are REQUESTED sixteen at a time and added in slab order (same sums as a plain loop, which the compiler runs
    // as one memory round trip per slab: 33 us per launch at 128 slabs for a kernel of two workgroups - round-4 trace)
    for (int k0 = 0; k0 < nch; k0 += 16) {
        float pm[16], pv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = min(k0 + j, nch - 1);
            pm[j] = p.ws[(long)k * p.C + c]; pv[j] = p.ws[(long)(NCHUNK_MAX + k) * p.C + c];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (k0 + j < nch) { m += pm[j]; v += pv[j]; }
    }
    m /= (float)p.R;
    const float var_b = v / (float)p.R;
    p.save_mean[c] = m;
    p.save_rstd[c] = 1.0f / sqrtf(var_b + p.eps);
    if (p.running_mean) {
        const float var_u = v / (float)(p.R > 1 ? p.R - 1 : 1);
        p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * m;
        p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * var_u;
    }
}

// eval mode: save_mean/rstd from the running statistics
__global__ void bn_eval_stats_kernel(BnArgs p) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    p.save_mean[c] = p.running_mean[c];
    p.save_rstd[c] = 1.0f / sqrtf(p.running_var[c] + p.eps);
}

__device__ __forceinline__ float bn_z(const BnArgs& p, long r, int c) {
    return (p.x[r * p.C + c] - p.save_mean[c]) * p.save_rstd[c] * p.gamma[c] + p.beta[c];
}
__device__ __forceinline__ float bn_drop(const BnArgs& p, long r, int c, float a) {
    return p.mask ? (p.mask[r * p.C + c] ? a * p.mask_scale : 0.f) : a;
}

// apply: y = dropout(act(bn(x)))  [+ highway combine].  grid-stride over output elements.
__global__ void bn_apply_kernel(BnArgs p) {
    const int Cy = p.hw_groups ? p.C / 2 : p.C;
    const long total = (long)p.R * Cy;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    GridRC rc(i0, stride, Cy);
    for (long i = i0; i < total; i += stride, rc.next()) {
        const long r = rc.row; const int cy = rc.col;
        if (!p.hw_groups) {
            p.y[i] = bn_drop(p, r, cy, apply_act(p.act, bn_z(p, r, cy)));
        } else {
            const int Cg = Cy / p.hw_groups; const int g = cy / Cg, cc = cy - g * Cg;
            const int c1 = g * 2 * Cg + cc, c2 = c1 + Cg;
            const float h1 = bn_drop(p, r, c1, apply_act(p.act, bn_z(p, r, c1)));
            const float h2 = bn_drop(p, r, c2, apply_act(p.act, bn_z(p, r, c2)));
            const float pg = sigmoidf_(h1);
            p.y[i] = h2 * pg + p.resid[i] * (1.f - pg);
        }
    }
}

// gradient w.r.t. the BN affine output z for element (r, c):  dz = d(out)/d(z)
__device__ __forceinline__ float bn_dz(const BnArgs& p, long r, int c, float& xhat_out) {
    const float xhat = (p.x[r * p.C + c] - p.save_mean[c]) * p.save_rstd[c];
    xhat_out = xhat;
    const float z = xhat * p.gamma[c] + p.beta[c];
    const float a = apply_act(p.act, z);
    float da;   // gradient arriving at the (dropped) activation output
    if (!p.hw_groups) {
        da = p.dy[r * p.C + c];
    } else {
        const int Cy = p.C / 2; const int Cg = Cy / p.hw_groups;
        const int g = c / (2 * Cg); const int within = c - g * 2 * Cg; const bool is_gate = within < Cg;
        const int cc = is_gate ? within : within - Cg;
        const int cy = g * Cg + cc; const int c1 = g * 2 * Cg + cc, c2 = c1 + Cg;
        const float dout = p.dy[r * Cy + cy];
        const float h1 = bn_drop(p, r, c1, apply_act(p.act, bn_z(p, r, c1)));
        const float pg = sigmoidf_(h1);
        if (is_gate) {
            const float h2 = bn_drop(p, r, c2, apply_act(p.act, bn_z(p, r, c2)));
            da = dout * (h2 - p.resid[r * Cy + cy]) * pg * (1.f - pg);
        } else {
            da = dout * pg;
        }
    }
    if (p.mask) da = p.mask[r * p.C + c] ? da * p.mask_scale : 0.f;
    float dz = da;
    if (p.act == MTTS_ACT_RELU) dz = z > 0.f ? da : 0.f;
    else if (p.act == MTTS_ACT_TANH) dz = da * (1.f - a * a);
    else if (p.act == MTTS_ACT_SIGMOID) dz = da * a * (1.f - a);
    return dz;
}

// Plain (non-highway) blocks: per-column constants in registers, operands of BN_UB rows requested together.
struct BnCol { float mean, rstd, gamma, beta; };
struct BnRows { float x[BN_UB], dy[BN_UB]; int keep[BN_UB]; };
__device__ __forceinline__ void bn_rows_load(const BnArgs& p, int r, int r1, int c, BnRows& w) {
    const uint8_t* mk = p.mask ? p.mask : reinterpret_cast<const uint8_t*>(p.x);      // absent mask: any readable bytes, ignored below
#pragma unroll
    for (int u = 0; u < BN_UB; ++u) {
        const long i = (long)min(r + 4 * u, r1 - 1) * p.C + c;
        w.x[u] = p.x[i]; w.dy[u] = p.dy[i]; w.keep[u] = (int)mk[i];
    }
}
__device__ __forceinline__ float bn_dz_plain(const BnArgs& p, const BnCol& k, float x, float dy, int keep, float& xhat) {
    xhat = (x - k.mean) * k.rstd;
    const float z = xhat * k.gamma + k.beta;
    const float a = apply_act(p.act, z);
    float da = dy;
    if (p.mask) da = keep ? da * p.mask_scale : 0.f;
    float dz = da;
    if (p.act == MTTS_ACT_RELU) dz = z > 0.f ? da : 0.f;
    else if (p.act == MTTS_ACT_TANH) dz = da * (1.f - a * a);
    else if (p.act == MTTS_ACT_SIGMOID) dz = da * a * (1.f - a);
    return dz;
}

// bwd pass 1: partial sums of dz and dz*xhat per channel -> ws[0..nch), ws[NCHUNK_MAX..)
__global__ void bn_bwd_reduce_kernel(BnArgs p) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int nch = gridDim.y;
    const int rows_per = (p.R + nch - 1) / nch;
    const int r0 = blockIdx.y * rows_per, r1 = min(p.R, r0 + rows_per);
    float s1 = 0.f, s2 = 0.f;
    if (c < p.C && !p.hw_groups) {
        const BnCol k = {p.save_mean[c], p.save_rstd[c], p.gamma[c], p.beta[c]};
        for (int r = r0 + threadIdx.y; r < r1; r += 4 * BN_UB) {
            BnRows w; bn_rows_load(p, r, r1, c, w);
#pragma unroll
            for (int u = 0; u < BN_UB; ++u)
                if (r + 4 * u < r1) { float xh; const float dz = bn_dz_plain(p, k, w.x[u], w.dy[u], w.keep[u], xh); s1 += dz; s2 += dz * xh; }
        }
    } else if (c < p.C) {
        for (int r = r0 + threadIdx.y; r < r1; r += 4) {
            float xh; const float dz = bn_dz(p, r, c, xh);
            s1 += dz; s2 += dz * xh;
        }
    }
    red[0][threadIdx.y][threadIdx.x] = s1; red[1][threadIdx.y][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.y == 0 && c < p.C) {
        p.ws[(long)blockIdx.y * p.C + c] = red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x];
        p.ws[(long)(NCHUNK_MAX + blockIdx.y) * p.C + c] = red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x];
    }
}

// bwd pass 2: dx, dgamma/dbeta (row-chunk 0 writes them), dresid.   grid (ceil(C/64), chunks) block (64,4)
__global__ void bn_bwd_apply_kernel(BnArgs p, int nch) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= p.C) return;
    const int nchy = gridDim.y;
    const int rows_per = (p.R + nchy - 1) / nchy;
    const int r0 = blockIdx.y * rows_per, r1 = min(p.R, r0 + rows_per);
    float sdz = 0.f, sdzx = 0.f;
    for (int k0 = 0; k0 < nch; k0 += 16) {          // sixteen slabs per memory round trip, added in slab order (see bn_finalize_kernel)
        float p1[16], p2[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = min(k0 + j, nch - 1);
            p1[j] = p.ws[(long)k * p.C + c]; p2[j] = p.ws[(long)(NCHUNK_MAX + k) * p.C + c];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (k0 + j < nch) { sdz += p1[j]; sdzx += p2[j]; }
    }
    if (blockIdx.y == 0 && threadIdx.y == 0) {
        if (p.dgamma) p.dgamma[c] = sdzx;
        if (p.dbeta) p.dbeta[c] = sdz;
    }
    const float g = p.gamma[c], rs = p.save_rstd[c];
    const float invR = 1.f / (float)p.R;
    if (!p.hw_groups) {
        const BnCol k = {p.save_mean[c], rs, g, p.beta[c]};
        for (int r = r0 + threadIdx.y; r < r1; r += 4 * BN_UB) {
            BnRows w; bn_rows_load(p, r, r1, c, w);
#pragma unroll
            for (int u = 0; u < BN_UB; ++u)
                if (r + 4 * u < r1) {
                    float xh; const float dz = bn_dz_plain(p, k, w.x[u], w.dy[u], w.keep[u], xh);
                    p.dx[(long)(r + 4 * u) * p.C + c] = p.training ? g * rs * (dz - sdz * invR - xh * sdzx * invR) : g * rs * dz;
                }
        }
        return;
    }
    for (int r = r0 + threadIdx.y; r < r1; r += 4) {
        float xh; const float dz = bn_dz(p, r, c, xh);
        float dxv;
        if (p.training) dxv = g * rs * (dz - sdz * invR - xh * sdzx * invR);
        else dxv = g * rs * dz;
        p.dx[(long)r * p.C + c] = dxv;
    }
}

// highway: gradient w.r.t. the block input, dresid = dout * (1 - sigmoid(h1)).
__global__ void hw_dresid_kernel(BnArgs p) {
    const int Cy = p.C / 2;
    const long total = (long)p.R * Cy;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    GridRC rc(i0, stride, Cy);
    for (long i = i0; i < total; i += stride, rc.next()) {
        const long r = rc.row; const int cy = rc.col;
        const int Cg = Cy / p.hw_groups; const int g = cy / Cg, cc = cy - g * Cg;
        const int c1 = g * 2 * Cg + cc;
        const float h1 = bn_drop(p, r, c1, apply_act(p.act, bn_z(p, r, c1)));
        p.dresid[i] = p.dy[i] * (1.f - sigmoidf_(h1));
    }
}

MTTS_API long mtts_bn_workspace_floats(int C) { return 2L * NCHUNK_MAX * C; }

MTTS_API int mtts_bn_act_fwd(const BnArgs* args, void* stream) {
    BnArgs p = *args;
    hipStream_t s = (hipStream_t)stream;
    MTTS_REQUIRE(p.R > 0 && p.C > 0, "mtts_bn_act_fwd: empty input");
    MTTS_REQUIRE(!p.hw_groups || (p.C % (2 * p.hw_groups) == 0 && p.resid), "mtts_bn_act_fwd: bad highway config");
    int nch = (p.R + 63) / 64; nch = nch < 1 ? 1 : (nch > NCHUNK_MAX ? NCHUNK_MAX : nch);
    dim3 grid(cdiv(p.C, 64), nch), block(64, 4);
    if (p.training) {
        hipLaunchKernelGGL(bn_sum_kernel, grid, block, 0, s, p);
        hipLaunchKernelGGL(bn_var_kernel, grid, block, 0, s, p);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(p.C, 256)), dim3(256), 0, s, p, nch);
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(p.C, 256)), dim3(256), 0, s, p);
    }
    const long total = (long)p.R * (p.hw_groups ? p.C / 2 : p.C);
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, s, p);
    MTTS_CHECK_LAUNCH("bn_act_fwd");
    return 0;
}

MTTS_API int mtts_bn_act_bwd(const BnArgs* args, void* stream) {
    BnArgs p = *args;
    hipStream_t s = (hipStream_t)stream;
    MTTS_REQUIRE(p.R > 0 && p.C > 0 && p.dy && p.dx, "mtts_bn_act_bwd: bad arguments");
    int nch = (p.R + 63) / 64; nch = nch < 1 ? 1 : (nch > NCHUNK_MAX ? NCHUNK_MAX : nch);
    dim3 grid(cdiv(p.C, 64), nch), block(64, 4);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, block, 0, s, p);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, block, 0, s, p, nch);
    if (p.hw_groups && p.dresid) {
        const long total = (long)p.R * (p.C / 2);
        int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(hw_dresid_kernel, dim3(blocks), dim3(256), 0, s, p);
    }
    MTTS_CHECK_LAUNCH("bn_act_bwd");
    return 0;
}
