// Skinny (small-M) fp32 MFMA GEMM for the autoregressive steps:  Y[B,N] = sum_s X_s[B,K_s] * W_s[N,K_s]^T
// with B = batch rows (<= 64 per row tile), N = thousands of weight rows streamed once per step.
//
// This is the recurrent hot loop of the decoder (reference modules/layers.py:18-47 LSTMCell call sites
// modules/tacotron2.py:185,188; attention query modules/attention.py:68; frame/stop projection
// modules/tacotron2.py:192-193) and of the encoder BiLSTM (modules/encoder.py:41-44).
//
// Decomposition (gfx950): one workgroup owns 16 output columns for all rows of its row tile; its 4 waves
// (one per SIMD) take the 16-wide K chunks round-robin and reduce through LDS.  Weights go L2/HBM -> VGPR
// directly (each weight row is consumed by exactly one workgroup: LDS staging would be pure overhead),
// 16 B per lane along K; the MFMA k-slot trick (slot q <-> k = k0 + 4q + s for instruction s) turns one
// float4 per lane into four v_mfma_f32_16x16x4_f32.  Inputs may be given as up to 3 K-segments so the
// concatenations [prenet, context, h] / [h_att, context, h_gen] are never materialised.
//
// The chunk index is wave-uniform (readfirstlane), so segment selection is scalar; loads are branch-free
// (clamped addresses, zero-select on the K tail) and software-pipelined four chunks deep so that the
// compiler can use counted vmcnt waits.
//
// Epilogues: raw (optionally K-split partials), bias+activation+dropout, or the fused LSTM cell
// (gate nonlinearities, cell update, dropout / zoneout on h, packed-sequence carry, saved gates).
#include "common.h"

template <int MT>
struct Frag { float4 w; float4 x[MT]; };

// Segment table held in registers (SGPRs): copied field-by-field from the kernel argument so that the
// compiler never needs the argument struct in memory (address-selects on it would force a scratch copy).
struct SegTab {
    const float* x0; const float* x1; const float* x2;
    const float* w0; const float* w1; const float* w2;
    int K0, K1, K2, ldx0, ldx1, ldx2, ldw0, ldw1, ldw2;
    int n0, n1, total;
};

template <int MT>
__device__ __forceinline__ void sk_load(const SegTab t, int c, const int (&rows)[MT], int wrow, int lq, Frag<MT>& f) {
    // chunk index (global over the segments) -> segment parameters; everything here is wave-uniform
    const bool live = c < t.total;
    const int cc = live ? c : 0;
    const bool in0 = cc < t.n0, in1 = cc < t.n0 + t.n1;
    const float* sx = in0 ? t.x0 : (in1 ? t.x1 : t.x2);
    const float* sw = in0 ? t.w0 : (in1 ? t.w1 : t.w2);
    const int sK = in0 ? t.K0 : (in1 ? t.K1 : t.K2);
    const int ldx = in0 ? t.ldx0 : (in1 ? t.ldx1 : t.ldx2);
    const int ldw = in0 ? t.ldw0 : (in1 ? t.ldw1 : t.ldw2);
    const int k0 = (in0 ? cc : (in1 ? cc - t.n0 : cc - t.n0 - t.n1)) * 16;
    const int k = k0 + lq * 4;
    const bool ok = live && (k < sK);
    const int kc = ok ? k : 0;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 wv = *reinterpret_cast<const float4*>(sw + (long)wrow * ldw + kc);
    f.w = ok ? wv : z;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float4 xv = *reinterpret_cast<const float4*>(sx + (long)rows[m] * ldx + kc);
        f.x[m] = ok ? xv : z;
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
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int cb = blockIdx.x;                 // column block
    const int row0 = blockIdx.y * (MT * 16);
    const int ks = blockIdx.z;

    // weight row served by this lane's column li (clamped: out-of-range columns are never stored)
    int wrow;
    if (p.lstm == 1) { const int u = min(cb * 4 + (li & 3), p.H - 1); wrow = (li >> 2) * p.H + u; }
    else wrow = min(cb * 16 + li, p.N - 1);
    int rows[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) rows[m] = min(row0 + m * 16 + li, p.B - 1);

    SegTab t;
    t.x0 = p.seg[0].x; t.x1 = p.seg[1].x; t.x2 = p.seg[2].x;
    t.w0 = p.seg[0].w; t.w1 = p.seg[1].w; t.w2 = p.seg[2].w;
    t.K0 = p.seg[0].K; t.K1 = p.seg[1].K; t.K2 = p.seg[2].K;
    t.ldx0 = p.seg[0].ldx; t.ldx1 = p.seg[1].ldx; t.ldx2 = p.seg[2].ldx;
    t.ldw0 = p.seg[0].ldw; t.ldw1 = p.seg[1].ldw; t.ldw2 = p.seg[2].ldw;
    const int nseg = p.nseg;
    t.n0 = (t.K0 + 15) >> 4;
    t.n1 = nseg > 1 ? (t.K1 + 15) >> 4 : 0;
    const int n2 = nseg > 2 ? (t.K2 + 15) >> 4 : 0;
    t.total = t.n0 + t.n1 + n2;
    const int total = t.total;

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // chunk c is served by wave (c % 4) of K-split ((c / 4) % ksplit); 4-deep software pipeline
    const int step = 4 * p.ksplit;
    int c = ks * 4 + wave;
    Frag<MT> f0, f1, f2, f3;
    sk_load<MT>(t, c, rows, wrow, lq, f0);
    sk_load<MT>(t, c + step, rows, wrow, lq, f1);
    sk_load<MT>(t, c + 2 * step, rows, wrow, lq, f2);
    sk_load<MT>(t, c + 3 * step, rows, wrow, lq, f3);
    for (; c < total; c += 4 * step) {
        sk_mma<MT>(f0, acc);
        sk_load<MT>(t, c + 4 * step, rows, wrow, lq, f0);
        sk_mma<MT>(f1, acc);
        sk_load<MT>(t, c + 5 * step, rows, wrow, lq, f1);
        sk_mma<MT>(f2, acc);
        sk_load<MT>(t, c + 6 * step, rows, wrow, lq, f2);
        sk_mma<MT>(f3, acc);
        sk_load<MT>(t, c + 7 * step, rows, wrow, lq, f3);
    }

    // C/D layout 16x16: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + lq * 4 + r][li] = acc[m][r];
    __syncthreads();

    if (p.lstm == 0) {
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

    if (p.lstm == 2) {
        // ---- LSTM cell backward: the 16 columns of this block are 16 hidden units; thread -> (row, unit)
        for (int e = threadIdx.x; e < MT * 16 * 16; e += 256) {
            const int rr = e >> 4, cc = e & 15;
            const int row = row0 + rr, u = cb * 16 + cc;
            if (row >= p.B || u >= p.H) continue;
            const long hi = (long)row * p.H + u;
            float dh = red[0][rr][cc] + red[1][rr][cc] + red[2][rr][cc] + red[3][rr][cc];
            if (p.dh_a) dh += p.dh_a[(long)row * p.ld_dh_a + u];
            if (p.dh_b) dh += p.dh_b[hi];
            for (int k = 0; k < p.n_part; ++k) dh += p.part[(long)k * p.part_ks + (long)row * p.part_ld + p.part_col0 + u];
            float dc = p.dc_in[hi];
            const float* gp = p.gates + (long)row * 4 * p.H + u;
            const float ig = gp[0], fg = gp[p.H], gg = gp[2 * p.H], og = gp[3 * p.H];
            const float cp = p.c_prev[hi];
            const bool carried = p.lengths && p.t >= p.lengths[row];
            float dh_carry = 0.f, dc_carry = 0.f, dhn = dh, dcn = dc;
            if (carried) { dh_carry = dh; dc_carry = dc; dhn = 0.f; dcn = 0.f; }
            else if (p.zone == 1) {
                if (p.hmask && !p.hmask[hi]) { dh_carry = dh; dhn = 0.f; }
                if (p.cmask && !p.cmask[hi]) { dc_carry = dc; dcn = 0.f; }
            } else if (p.hmask) {
                dhn = p.hmask[hi] ? dh * p.hscale : 0.f;
            }
            const float cn = fg * cp + ig * gg;
            const float th = tanhf_(cn);
            const float d_o = dhn * th;
            const float dct = dcn + dhn * og * (1.f - th * th);
            float* dg = p.dgates_out + (long)row * p.ld_dgates + u;
            dg[0] = dct * gg * ig * (1.f - ig);
            dg[p.H] = dct * cp * fg * (1.f - fg);
            dg[2 * p.H] = dct * ig * (1.f - gg * gg);
            dg[3 * p.H] = d_o * og * (1.f - og);
            p.dc_out[hi] = dct * fg + dc_carry;
            if (p.dh_carry_out) p.dh_carry_out[hi] = dh_carry;
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
    MTTS_REQUIRE(p.nseg >= 0 && p.nseg <= 3 && p.B > 0 && (p.nseg > 0 || p.lstm == 2), "skinny: bad nseg/B");
    for (int i = 0; i < p.nseg; ++i) {
        MTTS_REQUIRE((p.seg[i].K & 3) == 0 && (p.seg[i].ldx & 3) == 0 && (p.seg[i].ldw & 3) == 0 && p.seg[i].K > 0,
                     "skinny: segment %d needs K, ldx, ldw multiples of 4 (K=%d ldx=%d ldw=%d)", i, p.seg[i].K,
                     p.seg[i].ldx, p.seg[i].ldw);
        MTTS_REQUIRE(((uintptr_t)p.seg[i].x & 15) == 0 && ((uintptr_t)p.seg[i].w & 15) == 0,
                     "skinny: segment %d pointers must be 16-byte aligned", i);
    }
    const int ks = p.ksplit < 1 ? 1 : p.ksplit;
    SkinnyArgs q = p; q.ksplit = ks;
    if (p.nseg == 0) { q.seg[0] = SkSeg{p.gates, p.gates, 0, 0, 0}; q.nseg = 1; }   // pure pointwise: empty K range
    for (int i = q.nseg; i < 3; ++i) q.seg[i] = q.seg[0];      // keep the unused selectors dereferenceable
    if (p.lstm == 2) q.N = p.H;
    const int cbs = p.lstm == 1 ? cdiv(p.H, 4) : cdiv(q.N, 16);
    if (p.B <= 16) hipLaunchKernelGGL(skinny_kernel<1>, dim3(cbs, cdiv(p.B, 16), ks), dim3(256), 0, s, q);
    else if (p.B <= 32) hipLaunchKernelGGL(skinny_kernel<2>, dim3(cbs, cdiv(p.B, 32), ks), dim3(256), 0, s, q);
    else hipLaunchKernelGGL(skinny_kernel<4>, dim3(cbs, cdiv(p.B, 64), ks), dim3(256), 0, s, q);
    MTTS_CHECK_LAUNCH("skinny_kernel");
    return 0;
}

MTTS_API int mtts_skinny_gemm(const SkinnyArgs* args, void* stream) { return skinny_launch(*args, (hipStream_t)stream); }
