// Skinny (small-M) fp32 MFMA GEMM for the autoregressive steps:  Y[B,N] = sum_s X_s[B,K_s] * W_s[N,K_s]^T
// with B = batch rows (<= 64 per row tile), N = thousands of weight rows streamed once per step.
//
// This is the recurrent hot loop of the decoder (reference modules/layers.py:18-47 LSTMCell call sites
// modules/tacotron2.py:185,188; attention query modules/attention.py:68; frame/stop projection
// modules/tacotron2.py:192-193) and of the encoder BiLSTM (modules/encoder.py:41-44), forward and backward.
//
// Decomposition (gfx950): one workgroup owns 16 output columns for all rows of its row tile; its 8 waves
// (two per SIMD) take the 16-wide K chunks round-robin and reduce through LDS.  Weights go L2/HBM -> VGPR
// directly (each weight row is consumed by exactly one workgroup: LDS staging would be pure overhead),
// 16 B per lane along K; the MFMA k-slot trick (slot q <-> k = k0 + 4q + s for instruction s) turns one
// float4 per lane into four v_mfma_f32_16x16x4_f32.  Inputs may be given as up to 3 K-segments so the
// concatenations [prenet, context, h] / [h_att, context, h_gen] are never materialised.
//
// These kernels are memory-LATENCY bound (a step is a few microseconds of MFMA work behind ~1-2 us round
// trips), so the structure is about bytes in flight:
//   * the chunk index is wave-uniform (readfirstlane) -> scalar segment selection, branch-free loads
//     (clamped addresses, zero-select on the K tail), 4 chunks per wave in flight, 8 waves per CU;
//   * every operand of the epilogue (precomputed gate addends, biases, previous cell state, masks,
//     saved gates, partial sums of the later step) is requested BEFORE the main loop, so the epilogue
//     adds no further round trip.
//
// Epilogues: raw (optionally K-split partials), bias+activation+dropout, the fused LSTM cell
// (gate nonlinearities, cell update, dropout / zoneout on h, packed-sequence carry, saved gates) and the
// LSTM cell backward.
#pragma once
#include "common.h"

constexpr int NW = 8;                 // waves per workgroup
constexpr int NT = NW * 64;

template <int MT>
struct Frag { float4 w; float4 x[MT]; bool xp, wp; };   // xp/wp: operand already in MFMA tile order (wave-uniform)

// Segment table held in registers (SGPRs): copied field-by-field from the kernel argument so that the
// compiler never needs the argument struct in memory (address-selects on it would force a scratch copy).
struct SegTab {
    const float* x0; const float* x1; const float* x2;
    const float* w0; const float* w1; const float* w2;
    int K0, K1, K2, ldx0, ldx1, ldx2, ldw0, ldw1, ldw2;
    int xp0, xp1, xp2, wp0, wp1, wp2;     // packed-operand flags per segment
    int n0, n1, n2, total;
    int cb, mt0, mt_last;                 // column block / first absolute 16-row tile of this workgroup / last existing tile
};

// One zero tile (1 KiB): the target of every request that must read as zero - chunks past the end of K, the K tail of a row-major
// operand.  The ADDRESS is selected, never the loaded value: a select (or a branch) on the value makes the compiler wait for the
// load right behind it (`s_waitcnt vmcnt(0)` per row-major operand and chunk - one memory round trip each; round 4).
static __device__ __attribute__((aligned(16))) float g_sk_zero[256];

// Loads are QUAD-COALESCED: lane l fetches 16 B of row (l >> 2) at k-quad (l & 3), so four consecutive lanes cover
// 64 contiguous bytes and one wave instruction covers a 16-row x 16-k tile in 16 requests.  (Fetching directly in the
// MFMA operand layout - row = l & 15, k-quad = l >> 4 - puts consecutive lanes on different rows: 64 separate 16-B
// requests per instruction, and the texture-address unit becomes the bottleneck: 18 us -> 12.5 us per LSTM step.)
// The MFMA layout is restored in registers with ds_bpermute when the fragment is consumed.
// PK = 1: every segment has both operands in MFMA tile order (the per-step products of the decoder backward): no row-major address,
// no layout conversion; PK = 2: row-major operands only (BiLSTM steps, query / frame projections); PK = 0: mixed - the launcher
// picks the instantiation.
template <int MT, int PK>
__device__ __forceinline__ void sk_load(const SegTab t, int c, const int (&rows)[MT], int wrow, int kq, int lane, Frag<MT>& f) {
    const bool live = c < t.total;                  // wave-uniform
    const int cc = live ? c : 0;
    // chunk index (global over the segments) -> segment parameters; everything here is wave-uniform
    const bool in0 = cc < t.n0, in1 = cc < t.n0 + t.n1;
    const float* sx = in0 ? t.x0 : (in1 ? t.x1 : t.x2);
    const float* sw = in0 ? t.w0 : (in1 ? t.w1 : t.w2);
    const int nc = in0 ? t.n0 : (in1 ? t.n1 : t.n2);
    const int cs = in0 ? cc : (in1 ? cc - t.n0 : cc - t.n0 - t.n1);      // chunk within its segment
    const float* zp = g_sk_zero + lane * 4;
    if (PK == 1 || PK == 3) {       // 1 KiB contiguous tiles, already in MFMA operand order (PK 3: bf16 pairs of 16-k chunks, see sk_mma)
        f.xp = true; f.wp = true;
        const float* wa = sw + (((long)t.cb * nc + cs) * 64 + lane) * 4;
        f.w = *reinterpret_cast<const float4*>(live ? wa : zp);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* xa = sx + (((long)min(t.mt0 + m, t.mt_last) * nc + cs) * 64 + lane) * 4;
            f.x[m] = *reinterpret_cast<const float4*>(live ? xa : zp);
        }
        return;
    }
    const int sK = in0 ? t.K0 : (in1 ? t.K1 : t.K2);
    const int ldx = in0 ? t.ldx0 : (in1 ? t.ldx1 : t.ldx2);
    const int ldw = in0 ? t.ldw0 : (in1 ? t.ldw1 : t.ldw2);
    if (PK == 2) {       // row-major operands only
        const int k = cs * 16 + kq * 4;
        const bool ok = live && k < sK;             // per lane
        f.xp = false; f.wp = false;
        f.w = *reinterpret_cast<const float4*>(ok ? sw + (long)wrow * ldw + k : zp);
#pragma unroll
        for (int m = 0; m < MT; ++m) f.x[m] = *reinterpret_cast<const float4*>(ok ? sx + (long)rows[m] * ldx + k : zp);
        return;
    }
    const int xp = in0 ? t.xp0 : (in1 ? t.xp1 : t.xp2);
    const int wp = in0 ? t.wp0 : (in1 ? t.wp1 : t.wp2);
    const int k = cs * 16 + kq * 4;
    const bool ok = live && k < sK;                 // per lane; packed tiles are whole (zero padded by the packing kernels)
    f.xp = xp != 0; f.wp = wp != 0;
    {
        const float* wa = wp ? sw + (((long)t.cb * nc + cs) * 64 + lane) * 4 : sw + (long)wrow * ldw + k;
        f.w = *reinterpret_cast<const float4*>((wp ? live : ok) ? wa : zp);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float* xa = xp ? sx + (((long)min(t.mt0 + m, t.mt_last) * nc + cs) * 64 + lane) * 4 : sx + (long)rows[m] * ldx + k;
        f.x[m] = *reinterpret_cast<const float4*>((xp ? live : ok) ? xa : zp);
    }
}

__device__ __forceinline__ float4 to_mfma_layout(const float4 v, int src_lane) {
    return make_float4(__shfl(v.x, src_lane, 64), __shfl(v.y, src_lane, 64), __shfl(v.z, src_lane, 64), __shfl(v.w, src_lane, 64));
}

// PK = 3 (bf16 mode of the per-step backward products, round 5): a tile is a PAIR of 16-k chunks, 4 bf16 each per lane (the fp32 tile's
// lane <-> (row, k quad) map, so the producers change only their store): two v_mfma_f32_16x16x16_bf16 per tile instead of eight
// v_mfma_f32_16x16x4_f32 at 1/16 of their rate, half the operand bytes.  Operands RNE-rounded by their producers
// (mtts_pack_weight_bf16, the cell backward's dg_pack_out with dg_pack_bf16), fp32 accumulation.
typedef __attribute__((ext_vector_type(4))) short sk_s16x4;
__device__ __forceinline__ sk_s16x4 sk_bf16_half(float lo, float hi) {
    const uint2 u = make_uint2(__float_as_uint(lo), __float_as_uint(hi));
    sk_s16x4 r; __builtin_memcpy(&r, &u, 8);
    return r;
}
template <int MT, int PK>
__device__ __forceinline__ void sk_mma(const Frag<MT>& f, f32x4 (&acc)[MT], int src_lane) {
    if (PK == 3) {
        const sk_s16x4 w0 = sk_bf16_half(f.w.x, f.w.y), w1 = sk_bf16_half(f.w.z, f.w.w);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sk_bf16_half(f.x[m].x, f.x[m].y), w0, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sk_bf16_half(f.x[m].z, f.x[m].w), w1, acc[m], 0, 0, 0);
        }
        return;
    }
    const float4 w4 = (PK == 1 || (PK == 0 && f.wp)) ? f.w : to_mfma_layout(f.w, src_lane);
    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float4 x4 = (PK == 1 || (PK == 0 && f.xp)) ? f.x[m] : to_mfma_layout(f.x[m], src_lane);
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], wv[s], acc[m], 0, 0, 0);
    }
}

// ---- epilogue operands, fetched ahead of the main loop ------------------------------------------------------
struct FwdPre {            // LSTM cell forward, one (row, unit) per thread
    float pre[4], bi[4], bh[4], cp, hp;
    int hm, cm, len;
    bool valid;
};
constexpr int MAX_PART = 8;
struct BwdPre {            // LSTM cell backward, one (row, unit)
    float dh_extra, dc, g[4], cp;
    int hm, cm, len;
    bool valid;
};
struct BwdRaw { float a, b, parts[MAX_PART]; };      // the addends of dh_extra as they were loaded (summed by bwd_finish)

// Epilogue operands of the fused LSTM cells.  Absent operands (null pointers, kernel-uniform) are skipped by branches: requesting
// them unconditionally from a fallback address was measured SLOWER (scripts/bench_skinny.py, cell backward at batch 64: 4.5 -> 5.5 us -
// eight more requests per thread, all of a launch on one cache line).
__device__ __forceinline__ void fwd_prefetch(const SkinnyArgs& p, int row, int u, bool valid, FwdPre& f) {
    f.valid = valid;
    const int r = valid ? row : 0, uu = valid ? u : 0;
    const long hi = (long)r * p.H + uu;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int gc = g * p.H + uu;
        f.pre[g] = p.pre ? p.pre[(long)r * p.ldpre + gc] : 0.f;
        f.bi[g] = p.b_ih ? p.b_ih[gc] : 0.f;
        f.bh[g] = p.b_hh ? p.b_hh[gc] : 0.f;
    }
    f.cp = p.c_prev[hi];
    f.hp = p.h_prev ? p.h_prev[hi] : 0.f;
    f.hm = p.hmask ? (int)p.hmask[hi] : 1;
    f.cm = p.cmask ? (int)p.cmask[hi] : 1;
    f.len = p.lengths ? p.lengths[r] : 0x7fffffff;
}

// The addends of dh_extra are only REQUESTED here; bwd_finish sums them (same order as ever: a, b, parts 0..7) after the main
// loop's first fragments have been requested as well.
__device__ __forceinline__ void bwd_prefetch(const SkinnyArgs& p, int row, int u, bool valid, BwdPre& f, BwdRaw& w) {
    f.valid = valid;
    const int r = valid ? row : 0, uu = valid ? u : 0;
    const long hi = (long)r * p.H + uu;
    f.len = p.lengths ? p.lengths[r] : 0x7fffffff;
    // packed-sequence semantics (reference modules/encoder.py:41-44: pad_packed_sequence): the output at a padded position is a
    // constant zero, so the upstream gradient of a carried step is dropped - only the recurrent / carried parts pass through
    w.a = (p.dh_a && p.t < f.len) ? p.dh_a[(long)r * p.ld_dh_a + uu] : 0.f;
    w.b = p.dh_b ? p.dh_b[hi] : 0.f;
#pragma unroll
    for (int k = 0; k < MAX_PART; ++k)
        w.parts[k] = (k < p.n_part) ? p.part[(long)k * p.part_ks + (long)r * p.part_ld + p.part_col0 + uu] : 0.f;
    f.dc = p.dc_in[hi];
    const float* gp = p.gates + (long)r * 4 * p.H + uu;
    f.g[0] = gp[0]; f.g[1] = gp[p.H]; f.g[2] = gp[2 * p.H]; f.g[3] = gp[3 * p.H];
    f.cp = p.c_prev[hi];
    f.hm = p.hmask ? (int)p.hmask[hi] : 1;
    f.cm = p.cmask ? (int)p.cmask[hi] : 1;
}

__device__ __forceinline__ void bwd_finish(const SkinnyArgs& p, BwdPre& f, const BwdRaw& w) {
    float e = w.a;
    if (p.dh_b) e += w.b;
#pragma unroll
    for (int k = 0; k < MAX_PART; ++k) e += w.parts[k];
    f.dh_extra = e;
}

template <int MT>
__device__ __forceinline__ float red_sum(const float (&red)[NW][MT * 16][17], int rr, int cc) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][rr][cc];
    return v;
}

template <int MT>
__device__ __forceinline__ void fwd_cell(const SkinnyArgs& p, const float (&red)[NW][MT * 16][17], int rr, int uu, int row,
                                         int u, const FwdPre& f) {
    if (!f.valid) return;
    const int hm = f.hm, cm = f.cm, len = f.len;
    float g4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        g4[g] = red_sum<MT>(red, rr, g * 4 + uu) + f.pre[g] + f.bi[g] + f.bh[g];
    const long hi = (long)row * p.H + u;
    const float cp = f.cp;
    const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf_(g4[2]), og = sigmoidf_(g4[3]);
    const float cn = fg * cp + ig * gg;
    const float hn = og * tanhf_(cn);
    const bool carried = p.t >= len;
    if (p.gates_out) {
        float* go = p.gates_out + (long)row * 4 * p.H + u;
        go[0] = carried ? 0.f : ig; go[p.H] = carried ? 0.f : fg; go[2 * p.H] = carried ? 0.f : gg; go[3 * p.H] = carried ? 0.f : og;
    }
    float ho, co = cn;
    if (carried) { ho = f.hp; co = cp; }
    else if (p.zone == 1) { ho = hm ? hn : f.hp; co = cm ? cn : cp; }          // keep flag set -> take the new value
    else if (p.zone == 2) { ho = p.zh * f.hp + (1.f - p.zh) * hn; co = p.zc * cp + (1.f - p.zc) * cn; }
    else ho = p.hmask ? (hm ? hn * p.hscale : 0.f) : hn;
    p.h_out[hi] = ho;
    p.c_out[hi] = co;
    if (p.h_pack_out)      // MFMA tile order copy for the next step's X operand: lane 16*q + i, column 4*q + s of chunk u / 16
        p.h_pack_out[((((long)(row >> 4) * (p.H >> 4) + (u >> 4)) * 64) + 4 * (u & 12) + (row & 15)) * 4 + (u & 3)] = ho;
    if (p.y_out) p.y_out[(long)row * p.ldy + u] = carried ? 0.f : ho;
}

// write-through (sc1) store of one float: data that ANOTHER workgroup of the same launch reads (persistent backward, pbwd.hip)
__device__ __forceinline__ void sk_store(float* p, float v, bool sc1) {
    if (sc1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <int MT, int SC1 = 0>
__device__ __forceinline__ void bwd_cell(const SkinnyArgs& p, const float (&red)[NW][MT * 16][17], int rr, int cc, int row,
                                         int u, const BwdPre& f) {
    if (!f.valid) return;
    const int hm = f.hm, cm = f.cm, len = f.len;
    const long hi = (long)row * p.H + u;
    const float dh = red_sum<MT>(red, rr, cc) + f.dh_extra;
    const float dc = f.dc;
    const float ig = f.g[0], fg = f.g[1], gg = f.g[2], og = f.g[3];
    const float cp = f.cp;
    const bool carried = p.t >= len;
    float dh_carry = 0.f, dc_carry = 0.f, dhn = dh, dcn = dc;
    if (carried) { dh_carry = dh; dc_carry = dc; dhn = 0.f; dcn = 0.f; }
    else if (p.zone == 1) {
        if (!hm) { dh_carry = dh; dhn = 0.f; }
        if (!cm) { dc_carry = dc; dcn = 0.f; }
    } else if (p.hmask) {
        dhn = hm ? dh * p.hscale : 0.f;
    }
    const float cn = fg * cp + ig * gg;
    const float th = tanhf_(cn);
    const float d_o = dhn * th;
    const float dct = dcn + dhn * og * (1.f - th * th);
    float* dg = p.dgates_out + (long)row * p.ld_dgates + u;
    const float dgv[4] = {dct * gg * ig * (1.f - ig), dct * cp * fg * (1.f - fg), dct * ig * (1.f - gg * gg), d_o * og * (1.f - og)};
    dg[0] = dgv[0]; dg[p.H] = dgv[1]; dg[2 * p.H] = dgv[2]; dg[3 * p.H] = dgv[3];
    if (p.dg_pack_out && p.dg_pack_bf16) {      // bf16 pair tiles (PK = 3 consumers): [row tile][4H / 32][64 lanes][8 bf16], RNE
        unsigned short* dp = reinterpret_cast<unsigned short*>(p.dg_pack_out);
        const long tile = (long)(row >> 4) * (p.H >> 3);           // 4H / 32 chunk pairs per row tile
        const int lane_e = (4 * (u & 12) + (row & 15)) * 8 + (u & 3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j = g * p.H + u;
            const unsigned b = __float_as_uint(dgv[g]);
            dp[(tile + (j >> 5)) * 512 + lane_e + ((j >> 4) & 1) * 4] = (unsigned short)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
        }
    } else if (p.dg_pack_out) {
        const long tile = (long)(row >> 4) * (p.H >> 2);           // 4H / 16 chunks per row tile
        const int lane_s = (4 * (u & 12) + (row & 15)) * 4 + (u & 3);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            sk_store(p.dg_pack_out + (tile + ((g * p.H + u) >> 4)) * 256 + lane_s, dgv[g], SC1 != 0);
    }
    p.dc_out[hi] = dct * fg + dc_carry;
    if (p.dh_carry_out) p.dh_carry_out[hi] = dh_carry;
}

// Body of the skinny kernel for workgroup (column block cb, row tile, K split ks); `red` is the workgroup's LDS reduction buffer.
// PLAIN = 1: the plain-product epilogue only (p.lstm == 0 is the caller's promise): the fused attention-backward launch
// SC1 = 1: the outputs another workgroup of the SAME launch consumes (raw / K-split partial sums, the cell backward's packed copy of dG)
// are stored write-through (persistent backward, pbwd.hip); 0: plain stores (a kernel boundary publishes them).
template <int MT, int DEPTH = 4, int PK = 0, int PLAIN = 0, int SC1 = 0>
__device__ __forceinline__ void skinny_body(const SkinnyArgs& p, float (&red)[NW][MT * 16][17], const int cb, const int row_tile,
                                            const int ks) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int row0 = row_tile * (MT * 16);

    // ---- epilogue operands first (independent of the GEMM): one memory round trip, overlapped with the main loop
    FwdPre fp; BwdPre bp0, bp1; BwdRaw br0, br1;
    fp.valid = false; bp0.valid = false; bp1.valid = false;
    if (!PLAIN && p.lstm == 1) {
        const int rr = tid >> 2, uu = tid & 3;
        const int row = row0 + rr, u = cb * 4 + uu;
        fwd_prefetch(p, row, u, tid < MT * 64 && row < p.B && u < p.H, fp);
    } else if (!PLAIN && p.lstm == 2) {
        {
            const int rr = tid >> 4, cc = tid & 15;
            const int row = row0 + rr, u = cb * 16 + cc;
            bwd_prefetch(p, row, u, tid < MT * 256 && row < p.B && u < p.H, bp0, br0);
        }
        if (MT * 256 > NT) {
            const int e = tid + NT;
            const int rr = e >> 4, cc = e & 15;
            const int row = row0 + rr, u = cb * 16 + cc;
            bwd_prefetch(p, row, u, e < MT * 256 && row < p.B && u < p.H, bp1, br1);
        }
    }

    // load geometry (quad-coalesced): this lane fetches tile row r4, k-quad kq4; dest lane l takes its MFMA operand
    // from source lane 4*(l & 15) + (l >> 4).  Out-of-range rows/columns are clamped (their results are never stored).
    const int r4 = lane >> 2, kq4 = lane & 3;
    const int src_lane = 4 * (lane & 15) + (lane >> 4);
    int wrow;
    if (!PLAIN && p.lstm == 1) { const int u = min(cb * 4 + (r4 & 3), p.H - 1); wrow = (r4 >> 2) * p.H + u; }
    else wrow = min(cb * 16 + r4, p.N - 1);
    int rows[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) rows[m] = min(row0 + m * 16 + r4, p.B - 1);

    SegTab t;
    t.x0 = p.seg[0].x; t.x1 = p.seg[1].x; t.x2 = p.seg[2].x;
    t.w0 = p.seg[0].w; t.w1 = p.seg[1].w; t.w2 = p.seg[2].w;
    t.K0 = p.seg[0].K; t.K1 = p.seg[1].K; t.K2 = p.seg[2].K;
    t.ldx0 = p.seg[0].ldx; t.ldx1 = p.seg[1].ldx; t.ldx2 = p.seg[2].ldx;
    t.ldw0 = p.seg[0].ldw; t.ldw1 = p.seg[1].ldw; t.ldw2 = p.seg[2].ldw;
    t.xp0 = p.seg[0].xpack; t.xp1 = p.seg[1].xpack; t.xp2 = p.seg[2].xpack;
    t.wp0 = p.seg[0].wpack; t.wp1 = p.seg[1].wpack; t.wp2 = p.seg[2].wpack;
    t.cb = cb; t.mt0 = row_tile * MT; t.mt_last = ((p.B + 15) >> 4) - 1;
    const int nseg = p.nseg;
    constexpr int CSH = PK == 3 ? 5 : 4, CRND = PK == 3 ? 31 : 15;      // k per chunk: 16, or 32 for the bf16 pair tiles
    t.n0 = (t.K0 + CRND) >> CSH;
    t.n1 = nseg > 1 ? (t.K1 + CRND) >> CSH : 0;
    t.n2 = nseg > 2 ? (t.K2 + CRND) >> CSH : 0;
    t.total = t.n0 + t.n1 + t.n2;
    const int total = t.total;

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // chunk c is served by wave (c % NW) of K-split ((c / NW) % ksplit); DEPTH-deep software pipeline per wave (4: default; 2: fewer
    // registers, used where the workgroup shares a CU with another one; 16: everything in flight at once - few-workgroup launches of
    // the free-running loop pay ONE memory round trip).  The first DEPTH requests are unconditional (chunks past the end re-read
    // chunk 0 and are never multiplied) and go out right behind the epilogue operands.
    const int step = NW * p.ksplit;
    int c = ks * NW + wave;
    {
        Frag<MT> f[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) sk_load<MT, PK>(t, c + i * step, rows, wrow, kq4, lane, f[i]);
        if (!PLAIN && p.lstm == 2) {
            bwd_finish(p, bp0, br0);
            if (MT * 256 > NT) bwd_finish(p, bp1, br1);
        }
        for (; c < total; c += DEPTH * step) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                sk_mma<MT, PK>(f[i], acc, src_lane);
                sk_load<MT, PK>(t, c + (DEPTH + i) * step, rows, wrow, kq4, lane, f[i]);
            }
        }
    }

    // C/D layout 16x16: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + lq * 4 + r][li] = acc[m][r];
    __syncthreads();

    if (PLAIN || p.lstm == 0) {
        for (int e = tid; e < MT * 16 * 16; e += NT) {
            const int rr = e >> 4, cc = e & 15;
            const int row = row0 + rr, col = cb * 16 + cc;
            if (row >= p.B || col >= p.N) continue;
            float v = red_sum<MT>(red, rr, cc);
            if (p.ksplit > 1) { sk_store(p.out + (long)ks * p.out_ks + (long)row * p.ldo + col, v, SC1 != 0); continue; }
            if (p.bias) v += p.bias[col];
            v = apply_act(p.act, v);
            if (p.mask) v = p.mask[(long)row * p.ldmask + col] ? v * p.mask_scale : 0.f;
            p.out[(long)row * p.ldo + col] = v;
        }
        return;
    }
    if (p.lstm == 2) {
        {
            const int rr = tid >> 4, cc = tid & 15;
            bwd_cell<MT, SC1>(p, red, rr, cc, row0 + rr, cb * 16 + cc, bp0);
        }
        if (MT * 256 > NT) {
            const int e = tid + NT;
            const int rr = e >> 4, cc = e & 15;
            bwd_cell<MT, SC1>(p, red, rr, cc, row0 + rr, cb * 16 + cc, bp1);
        }
        return;
    }
    {
        const int rr = tid >> 2, uu = tid & 3;
        fwd_cell<MT>(p, red, rr, uu, row0 + rr, cb * 4 + uu, fp);
    }
}

