// Skinny (small-M) fp32 MFMA GEMM launcher; the kernel body lives in skinny_body.h (shared with the fused step launches).
#include <stdlib.h>
#include "skinny_body.h"

// PK (skinny_body.h): 1 = every operand in MFMA tile order (the per-step products of the decoder backward), 2 = row-major only,
// 0 = mixed.
template <int MT, int PK, int DEPTH = 4>
__global__ __launch_bounds__(NT) void skinny_kernel(SkinnyArgs p) {
    __shared__ float red[NW][MT * 16][17];
    skinny_body<MT, DEPTH, PK>(p, red, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Low-register variant (2-deep load pipeline, <= 128 VGPRs): two 512-thread workgroups fit on one CU; used for every launch with
// more than 32 rows (co-residency with the helper streams' GEMM workgroups beats pipeline depth).
template <int MT, int PK>
__global__ __launch_bounds__(NT, 4) void skinny_kernel_lo(SkinnyArgs p) {
    __shared__ float red[NW][MT * 16][17];
    skinny_body<MT, 2, PK>(p, red, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---------------------------------------------------------------------------------------------------------------------------
// One-round-trip kernel for plain products with few output columns in the free-running loop (frame / stop projection: 6 column
// tiles, K = H + Dm; reference modules/tacotron2.py:191-193): workgroup = 16 output columns x 16 rows over the whole K, the 16-wide
// K chunks of all segments numbered through and dealt to the 8 waves round-robin, EVERY chunk of a wave requested before the first
// product.  These launches are bound by the instructions a wave issues (scripts/bench_skinny.py: the generic body with 16 chunks
// in flight took 9.6 us at 2 800 instructions per wave against 5.5 us for the 8-way K split it replaces - which then needs a
// slab-sum launch behind it), so the addressing is kept scalar: per segment ONE per-lane byte offset (row x leading dimension + k
// quad) for x and for w; per chunk a wave-uniform base pointer (segment base + chunk offset) - `global_load v, v_offset, s[base]`.
// Requirements (the launcher checks them, everything else takes the generic kernels): row-major operands, every K a multiple of 16,
// at most PJ_NF chunks per wave (K_total <= 2048), offsets below 2^31.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int PJ_NF = 16;

__global__ __launch_bounds__(NT) void skinny_proj_kernel(SkinnyArgs p) {
    __shared__ float red[NW][16][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int cb = blockIdx.x, row0 = blockIdx.y * 16;
    const int r4 = lane >> 2, kq4 = lane & 3;
    const int src_lane = 4 * (lane & 15) + (lane >> 4);
    const int wrow = min(cb * 16 + r4, p.N - 1), xrow = min(row0 + r4, p.B - 1);
    const int n0 = p.seg[0].K >> 4, n1 = p.nseg > 1 ? p.seg[1].K >> 4 : 0, n2 = p.nseg > 2 ? p.seg[2].K >> 4 : 0;
    const int n01 = n0 + n1, total = n01 + n2;
    // per-lane byte offsets, one per segment and operand
    const unsigned xo0 = (unsigned)(xrow * p.seg[0].ldx + kq4 * 4) * 4u, wo0 = (unsigned)(wrow * p.seg[0].ldw + kq4 * 4) * 4u;
    const unsigned xo1 = (unsigned)(xrow * p.seg[1].ldx + kq4 * 4) * 4u, wo1 = (unsigned)(wrow * p.seg[1].ldw + kq4 * 4) * 4u;
    const unsigned xo2 = (unsigned)(xrow * p.seg[2].ldx + kq4 * 4) * 4u, wo2 = (unsigned)(wrow * p.seg[2].ldw + kq4 * 4) * 4u;
    // epilogue operand with the fragments (thread -> (row tid >> 4, column tid & 15) of the tile, threads 0..255)
    const float bias_v = (p.bias ? p.bias : p.seg[0].w)[p.bias ? min(cb * 16 + (tid & 15), p.N - 1) : 0];
    float4 fx[PJ_NF], fw[PJ_NF];
#pragma unroll
    for (int i = 0; i < PJ_NF; ++i) {
        const int c = min(wave + NW * i, total - 1);            // wave-uniform; chunks past the end re-read the last one (never multiplied)
        const bool in0 = c < n0, in1 = c < n01;
        const int cs = in0 ? c : (in1 ? c - n0 : c - n01);
        const char* xb = reinterpret_cast<const char*>(in0 ? p.seg[0].x : (in1 ? p.seg[1].x : p.seg[2].x)) + cs * 64;
        const char* wb = reinterpret_cast<const char*>(in0 ? p.seg[0].w : (in1 ? p.seg[1].w : p.seg[2].w)) + cs * 64;
        const unsigned xo = in0 ? xo0 : (in1 ? xo1 : xo2), wo = in0 ? wo0 : (in1 ? wo1 : wo2);
        fx[i] = *reinterpret_cast<const float4*>(xb + xo);
        fw[i] = *reinterpret_cast<const float4*>(wb + wo);
    }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PJ_NF; ++i) {
        if (wave + NW * i < total) {                            // wave-uniform
            const float4 x4 = to_mfma_layout(fx[i], src_lane), w4 = to_mfma_layout(fw[i], src_lane);
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s2], wv[s2], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lq * 4 + r][li] = acc[r];
    __syncthreads();
    if (tid < 256) {
        const int rr = tid >> 4, cc = tid & 15;
        const int row = row0 + rr, col = cb * 16 + cc;
        if (row < p.B && col < p.N) {
            float v = red_sum<1>(red, rr, cc);
            if (p.bias) v += bias_v;
            v = apply_act(p.act, v);
            if (p.mask) v = p.mask[(long)row * p.ldmask + col] ? v * p.mask_scale : 0.f;
            p.out[(long)row * p.ldo + col] = v;
        }
    }
}

// Generic one-round-trip variant (16 chunks in flight through the generic body): the shapes skinny_proj_kernel does not take.
template <int MT>
__global__ __launch_bounds__(NT) void skinny_kernel_wide(SkinnyArgs p) {
    __shared__ float red[NW][MT * 16][17];
    skinny_body<MT, 16, 2, 1>(p, red, blockIdx.x, blockIdx.y, blockIdx.z);
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
    MTTS_REQUIRE(!(p.lstm == 2 && p.dg_pack_out) || (p.H & 15) == 0, "skinny: the cell backward's packed copy of dG needs H %% 16 == 0 (H=%d)", p.H);
    const int ks = p.ksplit < 1 ? 1 : p.ksplit;
    SkinnyArgs q = p; q.ksplit = ks;
    if (p.nseg == 0) { q.seg[0] = SkSeg{p.gates, p.gates, 0, 0, 0, 0, 0}; q.nseg = 1; }   // pure pointwise: empty K range
    for (int i = q.nseg; i < 3; ++i) q.seg[i] = q.seg[0];      // keep the unused selectors dereferenceable
    if (p.lstm == 2) q.N = p.H;
    if (p.lstm == 1 && !q.h_prev) q.h_prev = q.c_prev;
    const int cbs = p.lstm == 1 ? cdiv(p.H, 4) : cdiv(q.N, 16);
    // operand forms -> instantiation: 1 = every operand in MFMA tile order (or no product at all), 2 = row-major only, 0 = mixed
    bool all_p = true, none_p = true;
    for (int i = 0; i < p.nseg; ++i) {
        all_p = all_p && p.seg[i].xpack && p.seg[i].wpack;
        none_p = none_p && !p.seg[i].xpack && !p.seg[i].wpack;
    }
    bool all_b = p.nseg > 0;      // every operand in bf16 pair tiles (xpack == wpack == 2): plain products of the bf16 path's backward
    for (int i = 0; i < p.nseg; ++i) all_b = all_b && p.seg[i].xpack == 2 && p.seg[i].wpack == 2;
    for (int i = 0; i < p.nseg; ++i)
        MTTS_REQUIRE(all_b || (p.seg[i].xpack != 2 && p.seg[i].wpack != 2), "skinny: bf16 pair tiles (pack == 2) must be used by every operand of a launch");
    MTTS_REQUIRE(!all_b || p.lstm == 0, "skinny: bf16 pair tiles are for plain products");
    for (int i = 0; all_b && i < p.nseg; ++i) MTTS_REQUIRE((p.seg[i].K & 31) == 0, "skinny: bf16 pair tiles need K %% 32 == 0 (K=%d)", p.seg[i].K);
    const int pk = all_b ? 3 : all_p ? 1 : none_p ? 2 : 0;
#define SK_LAUNCH(KERNEL, MT_, GRID) do { \
        if (pk == 3) hipLaunchKernelGGL((KERNEL<MT_, 3>), GRID, dim3(NT), 0, s, q); \
        else if (pk == 1) hipLaunchKernelGGL((KERNEL<MT_, 1>), GRID, dim3(NT), 0, s, q); \
        else if (pk == 2) hipLaunchKernelGGL((KERNEL<MT_, 2>), GRID, dim3(NT), 0, s, q); \
        else hipLaunchKernelGGL((KERNEL<MT_, 0>), GRID, dim3(NT), 0, s, q); } while (0)
    // LSTM cell backward (K <= the query width): the launch is all epilogue operands (16 loads per (row, unit)); one 16-row tile per
    // workgroup gives four times the workgroups to fetch them (bit-identical: same K chunks per wave, same reduction order).
    if (p.lstm == 2 && p.B > 16 && p.B <= 64 && q.seg[0].K <= 256 && q.nseg == 1 && ks == 1) {
        // at most one K chunk per wave (K <= 128: the query width) -> the depth-1 instantiation: these launches are bound by the
        // instructions a wave issues, and a 4-deep pipeline spends three quarters of them on fragments that do not exist
        const dim3 grid(cbs, cdiv(p.B, 16), 1);
        if (q.seg[0].K > 16 * NW) SK_LAUNCH(skinny_kernel, 1, grid);
        else if (pk == 1) hipLaunchKernelGGL((skinny_kernel<1, 1, 1>), grid, dim3(NT), 0, s, q);
        else if (pk == 2) hipLaunchKernelGGL((skinny_kernel<1, 2, 1>), grid, dim3(NT), 0, s, q);
        else hipLaunchKernelGGL((skinny_kernel<1, 0, 1>), grid, dim3(NT), 0, s, q);
        MTTS_CHECK_LAUNCH("skinny_kernel");
        return 0;
    }
    if (p.lstm == 0 && ks == 1 && pk == 2 && (cbs <= 8 || cbs * cdiv(p.B, 64) <= 32)) {       // few workgroups: one memory round trip
        bool lean = true;
        long ktot = 0;
        for (int i = 0; i < p.nseg; ++i) {
            lean = lean && (p.seg[i].K & 15) == 0 && (long)p.B * p.seg[i].ldx < (1l << 28) && (long)q.N * p.seg[i].ldw < (1l << 28);
            ktot += p.seg[i].K;
        }
        if (lean && ktot <= 16 * NW * PJ_NF) hipLaunchKernelGGL(skinny_proj_kernel, dim3(cbs, cdiv(p.B, 16), 1), dim3(NT), 0, s, q);
        else hipLaunchKernelGGL(skinny_kernel_wide<1>, dim3(cbs, cdiv(p.B, 16), 1), dim3(NT), 0, s, q);
        MTTS_CHECK_LAUNCH("skinny_proj_kernel");
        return 0;
    }
    if (p.B <= 16) SK_LAUNCH(skinny_kernel, 1, dim3(cbs, cdiv(p.B, 16), ks));
    else if (p.B <= 32) SK_LAUNCH(skinny_kernel, 2, dim3(cbs, cdiv(p.B, 32), ks));
    // more than 32 rows: the <= 128-VGPR variant.  A 4-deep load pipeline (153 VGPRs) is ~10 % faster alone, but cannot share a CU with
    // two GEMM workgroups of the helper streams and then WAITS for them: 92.0-92.3 vs 94.4-94.8 ms per train step (round 2).
    else SK_LAUNCH(skinny_kernel_lo, 4, dim3(cbs, cdiv(p.B, 64), ks));
#undef SK_LAUNCH
    MTTS_CHECK_LAUNCH("skinny_kernel");
    return 0;
}

MTTS_API int mtts_skinny_gemm(const SkinnyArgs* args, void* stream) { return skinny_launch(*args, (hipStream_t)stream); }
