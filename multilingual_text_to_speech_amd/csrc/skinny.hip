// Skinny (small-M) fp32 MFMA GEMM for the autoregressive steps:  Y[B,N] = sum_s X_s[B,K_s] * W_s[N,K_s]^T
// with B = batch rows (<= 64 per row tile), N = thousands of weight rows streamed once per step.
//
// This is the recurrent hot loop of the decoder (reference modules/layers.py:18-47 LSTMCell call sites
// modules/tacotron2.py:185,188; attention query modules/attention.py:68; frame/stop projection
// modules/tacotron2.py:192-193) and of the encoder BiLSTM (modules/encoder.py:41-44).
//
// Decomposition (gfx950): one workgroup owns 16 output columns for all rows of its row tile; its 4 waves
// (one per SIMD) split the K range in 16-wide chunks and reduce through LDS.  Weights go HBM/L2 -> VGPR
// directly (each weight row is consumed by exactly one workgroup: LDS staging would be pure overhead),
// 16 B per lane along K; the MFMA k-slot trick (slot q <-> k = k0 + 4q + s for instruction s) turns one
// float4 per lane into four v_mfma_f32_16x16x4_f32.  Inputs may be given as up to 3 K-segments so the
// concatenations [prenet, context, h] / [h_att, context, h_gen] are never materialised.
//
// Epilogues: raw (optionally K-split partials), bias+activation+dropout, or the fused LSTM cell
// (gate nonlinearities, cell update, dropout / zoneout on h, packed-sequence carry, saved gates).
#include "common.h"



template <int MT>
struct Frag { float4 w; float4 x[MT]; };

template <int MT>
__device__ __forceinline__ void sk_load(const SkinnyArgs& p, int chunk, int row0, int wrow, bool wvalid, int li, int lq,
                                        Frag<MT>& f) {
    f.w = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < MT; ++m) f.x[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (chunk < 0) return;
    // locate (segment, k0) of this chunk
    int s = 0, c = chunk;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < p.nseg) {
            const int nc = (p.seg[i].K + 15) >> 4;
            if (s == i && c >= nc) { c -= nc; s = i + 1; }
        }
    }
    if (s >= p.nseg) return;
    const SkSeg sg = p.seg[s];
    const int k = c * 16 + lq * 4;
    if (k >= sg.K) return;
    if (wvalid) f.w = *reinterpret_cast<const float4*>(sg.w + (long)wrow * sg.ldw + k);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int r = row0 + m * 16 + li;
        if (r < p.B) f.x[m] = *reinterpret_cast<const float4*>(sg.x + (long)r * sg.ldx + k);
    }
}

template <int MT>
__device__ __forceinline__ void sk_mma(const Frag<MT>& f, f32x4 (&acc)[MT]) {
    const float wv[4] = {f.w.x, f.w.y, f.w.z, f.w.w};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float xv[4] = {f.x[m].x, f.x[m].y, f.x[m].z, f.x[m].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], wv[s], acc[m], 0, 0, 0);
    }
}

template <int MT>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs p) {
    __shared__ float red[4][MT * 16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int cb = blockIdx.x;                 // column block
    const int row0 = blockIdx.y * (MT * 16);
    const int ks = blockIdx.z;

    // weight row served by this lane's column li
    int wrow; bool wvalid;
    if (p.lstm) { const int u = cb * 4 + (li & 3); wrow = (li >> 2) * p.H + u; wvalid = u < p.H; }
    else { wrow = cb * 16 + li; wvalid = wrow < p.N; }

    int total = 0;
    for (int i = 0; i < p.nseg; ++i) total += (p.seg[i].K + 15) >> 4;
    const int per = (total + p.ksplit - 1) / p.ksplit;
    const int c_lo = ks * per, c_hi = min(total, c_lo + per);

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // two-stage software pipeline, 2 chunks per stage per wave
    Frag<MT> a0, a1, b0, b1;
    int c = c_lo + wave;
    sk_load<MT>(p, c < c_hi ? c : -1, row0, wrow, wvalid, li, lq, a0);
    sk_load<MT>(p, c + 4 < c_hi ? c + 4 : -1, row0, wrow, wvalid, li, lq, a1);
    for (; c < c_hi; c += 16) {
        sk_load<MT>(p, c + 8 < c_hi ? c + 8 : -1, row0, wrow, wvalid, li, lq, b0);
        sk_load<MT>(p, c + 12 < c_hi ? c + 12 : -1, row0, wrow, wvalid, li, lq, b1);
        sk_mma<MT>(a0, acc);
        sk_mma<MT>(a1, acc);
        sk_load<MT>(p, c + 16 < c_hi ? c + 16 : -1, row0, wrow, wvalid, li, lq, a0);
        sk_load<MT>(p, c + 20 < c_hi ? c + 20 : -1, row0, wrow, wvalid, li, lq, a1);
        sk_mma<MT>(b0, acc);
        sk_mma<MT>(b1, acc);
    }

    // C/D layout 16x16: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + lq * 4 + r][li] = acc[m][r];
    __syncthreads();

    if (!p.lstm) {
        for (int e = threadIdx.x; e < MT * 16 * 16; e += 256) {
            const int rr = e >> 4, cc = e & 15;
            const int row = row0 + rr, col = cb * 16 + cc;
            if (row >= p.B || col >= p.N) continue;
            float v = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc];
            if (p.ksplit > 1) { p.out[(long)ks * p.out_ks + (long)row * p.ldo + col] = v; continue; }
            if (p.bias) v += p.bias[col];
            v = apply_act(p.act, v);
            if (p.mask) v = p.mask[(long)row * p.ldmask + col] ? v * p.mask_scale : 0.f;
            p.out[(long)row * p.ldo + col] = v;
        }
        return;
    }

    // ---- fused LSTM cell: thread -> (row, unit)
    for (int e = threadIdx.x; e < MT * 16 * 4; e += 256) {
        const int rr = e >> 2, uu = e & 3;
        const int row = row0 + rr, u = cb * 4 + uu;
        if (row >= p.B || u >= p.H) continue;
        float g4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cc = g * 4 + uu;
            float v = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc];
            const int gc = g * p.H + u;
            if (p.pre) v += p.pre[(long)row * p.ldpre + gc];
            if (p.b_ih) v += p.b_ih[gc];
            if (p.b_hh) v += p.b_hh[gc];
            g4[g] = v;
        }
        const long hi = (long)row * p.H + u;
        const float cp = p.c_prev[hi];
        const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf_(g4[2]), og = sigmoidf_(g4[3]);
        float cn = fg * cp + ig * gg;
        float hn = og * tanhf_(cn);
        const bool carried = p.lengths && p.t >= p.lengths[row];
        if (p.gates_out) {
            float* go = p.gates_out + (long)row * 4 * p.H + u;
            go[0] = carried ? 0.f : ig; go[p.H] = carried ? 0.f : fg; go[2 * p.H] = carried ? 0.f : gg; go[3 * p.H] = carried ? 0.f : og;
        }
        float ho, co = cn;
        if (carried) { ho = p.h_prev[hi]; co = cp; }
        else if (p.zone == 1) {
            const float hp = p.h_prev[hi];
            ho = (p.hmask && !p.hmask[hi]) ? hp : hn;      // keep flag set -> take the new value
            co = (p.cmask && !p.cmask[hi]) ? cp : cn;
        } else if (p.zone == 2) {
            const float hp = p.h_prev[hi];
            ho = p.zh * hp + (1.f - p.zh) * hn;
            co = p.zc * cp + (1.f - p.zc) * cn;
        } else {
            ho = p.hmask ? (p.hmask[hi] ? hn * p.hscale : 0.f) : hn;
        }
        p.h_out[hi] = ho;
        p.c_out[hi] = co;
        if (p.y_out) p.y_out[(long)row * p.ldy + u] = carried ? 0.f : ho;
    }
}

int skinny_launch(const SkinnyArgs& p, hipStream_t s) {
    for (int i = 0; i < p.nseg; ++i) {
        MTTS_REQUIRE((p.seg[i].K & 3) == 0 && (p.seg[i].ldx & 3) == 0 && (p.seg[i].ldw & 3) == 0,
                     "skinny: segment %d needs K, ldx, ldw multiples of 4 (K=%d ldx=%d ldw=%d)", i, p.seg[i].K,
                     p.seg[i].ldx, p.seg[i].ldw);
        MTTS_REQUIRE(((uintptr_t)p.seg[i].x & 15) == 0 && ((uintptr_t)p.seg[i].w & 15) == 0,
                     "skinny: segment %d pointers must be 16-byte aligned", i);
    }
    MTTS_REQUIRE(p.nseg >= 1 && p.nseg <= 3 && p.B > 0, "skinny: bad nseg/B");
    const int ks = p.ksplit < 1 ? 1 : p.ksplit;
    SkinnyArgs q = p; q.ksplit = ks;
    const int cbs = p.lstm ? cdiv(p.H, 4) : cdiv(p.N, 16);
    if (p.B <= 16) hipLaunchKernelGGL(skinny_kernel<1>, dim3(cbs, cdiv(p.B, 16), ks), dim3(256), 0, s, q);
    else if (p.B <= 32) hipLaunchKernelGGL(skinny_kernel<2>, dim3(cbs, cdiv(p.B, 32), ks), dim3(256), 0, s, q);
    else hipLaunchKernelGGL(skinny_kernel<4>, dim3(cbs, cdiv(p.B, 64), ks), dim3(256), 0, s, q);
    MTTS_CHECK_LAUNCH("skinny_kernel");
    return 0;
}

MTTS_API int mtts_skinny_gemm(const SkinnyArgs* args, void* stream) { return skinny_launch(*args, (hipStream_t)stream); }
