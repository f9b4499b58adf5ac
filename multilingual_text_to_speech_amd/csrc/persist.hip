// Persistent (weights-stationary) decoder recurrences for MI355X: ONE launch runs all T steps of a recurrence.
//
// Why: a teacher-forced decoder step is two all-to-all exchanges (LSTM columns <-> attention samples) around ~10 us of
// streaming.  As separate launches every exchange costs a kernel boundary (~2.7 us measured in this decoder) plus the ramp and
// drain of a full-chip grid (~4 us), and every step re-streams 42 MB of recurrent weights from HBM / Infinity Cache.  MI355X has
// 256 CUs x (512 KiB of vector registers + 160 KiB of LDS) = 168 MiB of on-chip storage: the recurrent weights of a decoder LSTM
// (25.7 MB for [W_ih[:, P:] | W_hh], 16.8 MB for the generator's W_hh) stay ON CHIP for the whole decode, split by gate columns
// over the 256 workgroups (one per CU), and the only per-step global traffic is the [B, K] activation exchange (0.25-0.4 MB,
// L2-served) behind a two-level grid barrier (2.3 us measured, scripts/mb/mb_pbar.hip; broadcast reads 105-125 GB/s per CU,
// scripts/mb/mb_bcast.hip).
//
// Kernels
//   pgen7_kernel generator LSTM of the teacher-forced schedule (reference modules/tacotron2.py:187-188 with the input projection
//                hoisted into one GEMM): per step  gates = pre_gen[t] + h_gen[t] W_hh^T -> cell -> h_gen[t+1], as a DATAFLOW pipeline
//                (fp32): multiplier waves with register-stationary weight planes, one service wave per 16-row group, no workgroup
//                barrier in the loop; fp32 exchange.
//   pgen_kernel  the same recurrence with one grid barrier per step and bf16 weights in LDS: the bf16 path (precision 1).
//   pdec_kernel  attention LSTM + location-sensitive attention (reference modules/tacotron2.py:184-186, modules/attention.py:39-86,
//                modules/layers.py:18-47): per step
//                  phase 1 (column role: workgroup c owns LSTM units [4c, 4c+4)):  gates = pre_att[t] + [ctx_t | h_t] W^T -> cell
//                           -> h_{t+1} (row-major + exchange layout)                                        | grid barrier
//                  phase 2 (sample role: workgroup (b, j) owns sample b, attention channels [32j, 32j+32) and context columns
//                           [j Dm/4, (j+1) Dm/4)):  q = W_q[32j.., :] h_{t+1}[b]; partial energies over its 32 channels (location
//                           filter bank on MFMA, exact 3-way bf16 split) -> 4-way exchange of the partial energies (tagged 8-byte
//                           granules) -> masked softmax, cumulative alignment (stays in LDS), context columns -> ctx_{t+1}
//                                                                                                            | grid barrier
//                The query does not wait for the first barrier: it polls the own sample's h row (sentinel pre-filled by the host) -
//                the payload is the flag; the barrier is consumed after the attention step (PsDec.poll_h, MTTS_PDEC_POLL=0: off).
// Arithmetic is the step kernels' (lstm_step.hip): fp32 operands split exactly into three bf16 planes, six
// v_mfma_f32_16x16x32_bf16 terms per product, fp32 accumulation; the cell / attention math is fp32.
//
// Inter-workgroup data follows MI355X_MICROARCH.md (visibility): producers store write-through (sc1) and drain (s_waitcnt vmcnt(0))
// before the barrier arrive, consumers read with sc1 loads (L1 bypass), flags are relaxed agent-scope atomics, every spin is bounded
// and raises a device error word instead of hanging.  Results do not depend on dispatch order or placement; blockIdx % 8 == XCD is
// used for speed only (barrier groups).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "persist_sync.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

constexpr int PS_THREADS = 512;
constexpr int PS_WGS = 256;                 // one workgroup per CU; 4 LSTM units (16 gate columns) each: H = 1024
constexpr int PS_ERR_OFF = 8192, PS_XP_OFF = 16384;     // workspace: [counters: 4 groups x (1 + 8) x 128 B][error word][exchange ...]
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ps_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x27000);
}
__device__ __forceinline__ float4 ps_ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {       // L1-bypassing 16-byte load
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void ps_st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 x) {  // write-through 16-byte store
    u32x4 v; v.x = __float_as_uint(x.x); v.y = __float_as_uint(x.y); v.z = __float_as_uint(x.z); v.w = __float_as_uint(x.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}

// exact 3-way split of two floats into three packed bf16 pairs (truncation; residuals are exact in fp32) - as lstm_step.hip
__device__ __forceinline__ void ps_split_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
    const float rx = x - __uint_as_float(ux & 0xffff0000u), ry = y - __uint_as_float(uy & 0xffff0000u);
    const unsigned vx = __float_as_uint(rx), vy = __float_as_uint(ry);
    const float sx = rx - __uint_as_float(vx & 0xffff0000u), sy = ry - __uint_as_float(vy & 0xffff0000u);
    p1 = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    p2 = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}
union PsFrag { bf16x8 v; unsigned u[4]; };
struct PsFrag3 { PsFrag p[3]; };
__device__ __forceinline__ PsFrag3 ps_split8(const float4& lo, const float4& hi) {      // 8 consecutive k -> 3 bf16 planes
    PsFrag3 f;
    ps_split_pair(lo.x, lo.y, f.p[0].u[0], f.p[1].u[0], f.p[2].u[0]);
    ps_split_pair(lo.z, lo.w, f.p[0].u[1], f.p[1].u[1], f.p[2].u[1]);
    ps_split_pair(hi.x, hi.y, f.p[0].u[2], f.p[1].u[2], f.p[2].u[2]);
    ps_split_pair(hi.z, hi.w, f.p[0].u[3], f.p[1].u[3], f.p[2].u[3]);
    return f;
}
__device__ __forceinline__ unsigned ps_bf16_rne(float x) {      // upper 16 bits of the RNE-rounded value
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// acc += A B^T with fp32 accuracy: six bf16 MFMA terms, small ones first
__device__ __forceinline__ f32x4 ps_mma6(const PsFrag3& a, const PsFrag3& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2].v, b.p[0].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0].v, b.p[2].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1].v, b.p[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1].v, b.p[0].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0].v, b.p[1].v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0].v, b.p[0].v, acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Exchange layout ("XP") of an activation matrix X[rows][K] (K % 32 == 0): the MFMA A-fragment order of v_mfma_f32_16x16x32,
//   float index = ((((rt * nkb + kb) * 2 + hf) * 64) + (q4 * 16 + i16)) * 4 + e     row = 16 rt + i16,  k = 32 kb + 8 q4 + 4 hf + e
// so that a wave reads the fragment (rt, kb) as two fully contiguous 1 KiB loads and a producer that owns 4 consecutive columns
// (k % 4 == 0) of one row writes exactly one 16-byte quantum.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ps_xp_off(int row, int k, int nkb) {        // BYTE offset of the 16-byte quantum holding k .. k+3
    const int rt = row >> 4, i16 = row & 15, kb = k >> 5, j = k & 31, q4 = j >> 3, hf = (j >> 2) & 1;
    return (unsigned)((((((rt * nkb + kb) * 2 + hf) * 64) + (q4 * 16 + i16)) * 4) * 4);
}

// bf16 path (precision 1: operands of every contraction rounded to bf16, RNE): the exchange holds finished bf16 A-fragments ("XB"),
//   byte offset = ((rt * nkb + kb) * 64 + q4 * 16 + i16) * 16 + (k & 4) * 2       for the four bf16 of k .. k+3 (k % 4 == 0)
__device__ __forceinline__ unsigned ps_xb_off(int row, int k, int nkb) {
    const int rt = row >> 4, i16 = row & 15, kb = k >> 5, j = k & 31, q4 = j >> 3;
    return (unsigned)(((rt * nkb + kb) * 64 + q4 * 16 + i16) * 16 + (j & 4) * 2);
}
// publish four consecutive columns k .. k+3 of `row`: fp32 quantum (PREC 0) or four RNE-rounded bf16 (PREC 1), write-through
template <int PREC>
__device__ __forceinline__ void ps_publish4(__amdgpu_buffer_rsrc_t r, int row, int k, int nkb, float4 v) {
    if (PREC) {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        u32x2 w; w.x = ps_bf16_rne(v.x) | (ps_bf16_rne(v.y) << 16); w.y = ps_bf16_rne(v.z) | (ps_bf16_rne(v.w) << 16);
        __builtin_amdgcn_raw_buffer_store_b64(w, r, ps_xb_off(row, k, nkb), 0, 16);
    } else {
        ps_st16_sc1(r, ps_xp_off(row, k, nkb), v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Phase 1: partial gate pre-activations of this workgroup's 16 columns.
//   wave w takes the k-blocks [nkb w / 8, nkb (w+1) / 8) (at most NBW), streams the matching fragments of X (all RT row tiles)
//   from the exchange buffer (sc1 loads, DEPTH blocks in flight), takes its weight fragments from LDS and leaves its
//   [16 RT rows x 16 columns] partial sums in red[w][row][col].
// ---------------------------------------------------------------------------------------------------------------------------
// request the k-blocks [J0, J1) (J1 <= DEPTH) of this wave's share of [kb_lo, kb_hi) (the fragments of all RT row tiles) into xa
template <int NBW, int RT, int DEPTH, int J0 = 0, int J1 = DEPTH>
__device__ __forceinline__ void ps_gates_prefetch(__amdgpu_buffer_rsrc_t xr, int nkb, int kb_lo, int kb_hi, float4 (&xa)[DEPTH][RT][2]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrange = kb_hi - kb_lo;
    const int k0 = kb_lo + ((nrange * wave) >> 3), k1 = kb_lo + ((nrange * (wave + 1)) >> 3);
#pragma unroll
    for (int j = J0; j < J1 && j < NBW; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                xa[j][rt][hf] = ps_ld16_sc1(xr, (k0 + j < k1) ? (unsigned)((((rt * nkb + k0 + j) * 2 + hf) * 1024) + lane * 16) : 0xfffffff0u);
}

// acc += this wave's share of the k-blocks [kb_lo, kb_hi) (the eight waves split the range; at most NBW blocks per wave).
// preloaded: the caller has already run ps_gates_prefetch<.., 0, NPRE> for this range into xa (loads issued earlier, in the shadow of
// other work)
template <int NBW, int RT, int DEPTH, int NPRE = DEPTH>
__device__ __forceinline__ void ps_gates_acc(__amdgpu_buffer_rsrc_t xr, int nkb, int kb_lo, int kb_hi, const float4* __restrict__ wl, f32x4 (&acc)[RT],
                                             float4 (&xa)[DEPTH][RT][2], bool preloaded) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrange = kb_hi - kb_lo;
    const int k0 = kb_lo + ((nrange * wave) >> 3), k1 = kb_lo + ((nrange * (wave + 1)) >> 3);
    // fragment (rt, kb): byte offset ((rt * nkb + kb) * 2 + hf) * 1024 + lane * 16; blocks past k1 read out of range (= 0)
    auto issue = [&](int j, int slot) {
        const int kb = k0 + j;
        const bool ok = kb < k1;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const unsigned off = ok ? (unsigned)((((rt * nkb + kb) * 2 + hf) * 1024) + lane * 16) : 0xfffffff0u;
                xa[slot][rt][hf] = ps_ld16_sc1(xr, off);
            }
    };
    if (!preloaded) ps_gates_prefetch<NBW, RT, DEPTH, 0, NPRE>(xr, nkb, kb_lo, kb_hi, xa);
    if (NPRE < DEPTH) ps_gates_prefetch<NBW, RT, DEPTH, NPRE, DEPTH>(xr, nkb, kb_lo, kb_hi, xa);
    __builtin_amdgcn_sched_barrier(0);          // keep the whole burst ahead of the first use (hipcc would sink loads next to their uses)
    // (MFMA and split VALU of one SIMD do not overlap on gfx950 with 16x16x32 tiles - measured with software-pipelined and with
    //  hand-interleaved streams: their times add, 8.2k + 5.4k cycles per phase at K = 1568, B = 64 - so the loop stays simple.)
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int kb = min(k0 + j, nkb - 1);
        const PsFrag3 wb = ps_split8(wl[(kb * 2 + 0) * 64 + lane], wl[(kb * 2 + 1) * 64 + lane]);
        // two row tiles at a time: their six-term chains alternate, so no MFMA waits for the one issued just before it
#pragma unroll
        for (int r0 = 0; r0 < RT; r0 += 2) {
            PsFrag3 a[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = ps_split8(xa[j % DEPTH][min(r0 + m, RT - 1)][0], xa[j % DEPTH][min(r0 + m, RT - 1)][1]);
#define PS_MM(PA, PB) _Pragma("unroll") for (int m = 0; m < 2; ++m) if (r0 + m < RT) acc[r0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m].p[PA].v, wb.p[PB].v, acc[r0 + m], 0, 0, 0);
            PS_MM(2, 0) PS_MM(0, 2) PS_MM(1, 1) PS_MM(1, 0) PS_MM(0, 1) PS_MM(0, 0)
#undef PS_MM
        }
        if (j + DEPTH < NBW) { issue(j + DEPTH, j % DEPTH); __builtin_amdgcn_sched_barrier(0); }
    }
}

// this wave's partial sums -> red[wave][row][col]   (D layout: column = lane & 15, row = 4 (lane >> 4) + r)
template <int RT>
__device__ __forceinline__ void ps_gates_store(const f32x4 (&acc)[RT], float* __restrict__ red) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* out = red + wave * (64 * 16);
    const int i16 = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(16 * rt + 4 * q4 + r) * 16 + i16] = acc[rt][r];
}

template <int NBW, int RT, int DEPTH>
__device__ __forceinline__ void ps_gates(__amdgpu_buffer_rsrc_t xr, int nkb, const float4* __restrict__ wl, float* __restrict__ red) {
    f32x4 acc[RT];
    float4 xa[DEPTH][RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ps_gates_acc<NBW, RT, DEPTH>(xr, nkb, 0, nkb, wl, acc, xa, false);
    ps_gates_store<RT>(acc, red);
}

template <int NBW, int RT>
__device__ __forceinline__ void ps_gates_bf16(__amdgpu_buffer_rsrc_t xr, int nkb, const uint4* __restrict__ wlb, float* __restrict__ red);

// bf16 form: the exchange holds bf16 fragments, the LDS weight slice is bf16 ([nkb][64 lanes] x 16 B), one MFMA per product
template <int NBW, int RT>
__device__ __forceinline__ void ps_gates_bf16_prefetch(__amdgpu_buffer_rsrc_t xr, int nkb, int kb_lo, int kb_hi, u32x4 (&xa)[NBW][RT]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrange = kb_hi - kb_lo;
    const int k0 = kb_lo + ((nrange * wave) >> 3), k1 = kb_lo + ((nrange * (wave + 1)) >> 3);
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            xa[j][rt] = __builtin_amdgcn_raw_buffer_load_b128(xr, (k0 + j < k1) ? (unsigned)(((rt * nkb + k0 + j) * 1024) + lane * 16) : 0xfffffff0u, 0, 16);
}

template <int NBW, int RT>
__device__ __forceinline__ void ps_gates_bf16_acc(__amdgpu_buffer_rsrc_t xr, int nkb, int kb_lo, int kb_hi, const uint4* __restrict__ wlb, f32x4 (&acc)[RT],
                                                  u32x4 (&xa)[NBW][RT], bool preloaded) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrange = kb_hi - kb_lo;
    const int k0 = kb_lo + ((nrange * wave) >> 3);
    if (!preloaded) ps_gates_bf16_prefetch<NBW, RT>(xr, nkb, kb_lo, kb_hi, xa);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const uint4 w = wlb[min(k0 + j, nkb - 1) * 64 + lane];
        PsFrag wb; wb.u[0] = w.x; wb.u[1] = w.y; wb.u[2] = w.z; wb.u[3] = w.w;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            PsFrag a; a.u[0] = xa[j][rt].x; a.u[1] = xa[j][rt].y; a.u[2] = xa[j][rt].z; a.u[3] = xa[j][rt].w;
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, wb.v, acc[rt], 0, 0, 0);
        }
    }
}

template <int NBW, int RT>
__device__ __forceinline__ void ps_gates_bf16(__amdgpu_buffer_rsrc_t xr, int nkb, const uint4* __restrict__ wlb, float* __restrict__ red) {
    f32x4 acc[RT];
    u32x4 xa[NBW][RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ps_gates_bf16_acc<NBW, RT>(xr, nkb, 0, nkb, wlb, acc, xa, false);
    ps_gates_store<RT>(acc, red);
}

// ---------------------------------------------------------------------------------------------------------------------------
// LSTM cell for (row, unit) = (tid >> 2, tid & 3), tid < 4 B: sums the 8 partials, adds the hoisted projection and the bias,
// applies dropout / zoneout (reference modules/layers.py:26-47) and returns the h that recurs.  c / h of the previous step live
// in the caller's registers.
// ---------------------------------------------------------------------------------------------------------------------------
struct PsCellCfg { int zone; float hscale, zh, zc; };

__device__ __forceinline__ void ps_cell(const float* __restrict__ red, int row, int uu, float4 g4, const float4 pre4, float& c_state, float& h_state,
                                        int hm, int cm, bool has_hmask, const PsCellCfg& cfg, float4& gates_act) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(red + w * (64 * 16) + row * 16 + 4 * uu);
        g4.x += v.x; g4.y += v.y; g4.z += v.z; g4.w += v.w;
    }
    g4.x += pre4.x; g4.y += pre4.y; g4.z += pre4.z; g4.w += pre4.w;
    const float ig = sigmoidf_(g4.x), fg = sigmoidf_(g4.y), gg = tanhf_(g4.z), og = sigmoidf_(g4.w);
    const float cp = c_state, hp = h_state;
    const float cn = fg * cp + ig * gg;
    const float hn = og * tanhf_(cn);
    float ho, co = cn;
    if (cfg.zone == 1) { ho = hm ? hn : hp; co = cm ? cn : cp; }
    else if (cfg.zone == 2) { ho = cfg.zh * hp + (1.f - cfg.zh) * hn; co = cfg.zc * cp + (1.f - cfg.zc) * cn; }
    else ho = has_hmask ? (hm ? hn * cfg.hscale : 0.f) : hn;
    c_state = co; h_state = ho;
    gates_act = make_float4(ig, fg, gg, og);
}

// the four units of a row sit in four consecutive lanes (uu = lane & 3): gather them into lane uu == 0 as one float4
__device__ __forceinline__ float4 ps_quad_gather(float v) {
    float4 r;
    r.x = dpp_f<0x00>(v);      // quad_perm [0,0,0,0]
    r.y = dpp_f<0x55>(v);      // quad_perm [1,1,1,1]
    r.z = dpp_f<0xAA>(v);      // quad_perm [2,2,2,2]
    r.w = dpp_f<0xFF>(v);      // quad_perm [3,3,3,3]
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// pgen: generator LSTM, recurrent part
// ---------------------------------------------------------------------------------------------------------------------------
struct PsGen {
    int B, H, t0, t1;
    const float* w_packed;      // mtts_lstm_pack_weights(fp32) of W_hh: [4H/16 column groups][nkb][2][64][4]
    const float* bias_u;        // [4H] unit-major
    const float* pre;           // [T][B][4H] unit-major hoisted input projection
    float* h; float* c;         // [T+1][B][H]
    float* gates;               // [T][B][4H] gate-major (i | f | g | o) or NULL
    const uint8_t* hmask; const uint8_t* cmask;
    PsCellCfg cell;
    float* xp;                  // [2][64 rows][H] exchange (XP layout)
    PsSync sync;
};

template <int RT, int PREC>
__global__ __launch_bounds__(PS_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void pgen_kernel(PsGen p) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    const int tid = threadIdx.x, c = blockIdx.x;
    const int H = p.H, B = p.B, nkb = H >> 5, N = 4 * H;
    float4* wl = reinterpret_cast<float4*>(psm);                       // fp32: [nkb][2][64] float4 = nkb * 2 KiB; bf16: [nkb][64] x 16 B
    float* red = reinterpret_cast<float*>(psm + (size_t)nkb * (PREC ? 1024 : 2048));    // [8][64][16]
    // ---- stationary weights -> LDS
    {
        const int per = PREC ? 64 : 128;                                // 16-byte quanta per k-block
        const float4* src = reinterpret_cast<const float4*>(p.w_packed) + (size_t)c * nkb * per;
        for (int i = tid; i < nkb * per; i += PS_THREADS) wl[i] = src[i];
    }
    const int row = tid >> 2, uu = tid & 3, u = 4 * c + uu;
    const bool cellthr = tid < 4 * B;
    const int rowc = cellthr ? row : 0;
    const unsigned xp_bytes = (unsigned)(64 * H * (PREC ? 2 : 4));
    auto xregion = [&](int par) { return ps_rsrc(reinterpret_cast<char*>(p.xp) + (size_t)par * xp_bytes, xp_bytes); };
    float c_state = 0.f, h_state = 0.f;
    const float4 bias4 = *reinterpret_cast<const float4*>(p.bias_u + 4 * u);
    if (cellthr) {
        c_state = p.c[((size_t)p.t0 * B + row) * H + u];
        h_state = p.h[((size_t)p.t0 * B + row) * H + u];
    }
    {   // publish h[t0] in the exchange layout
        const float4 h4 = ps_quad_gather(h_state);
        if (cellthr && uu == 0) ps_publish4<PREC>(xregion(p.t0 & 1), row, 4 * c, nkb, h4);
    }
    unsigned epoch = 0;
    if (!ps_barrier(p.sync, ++epoch)) return;
    for (int t = p.t0; t < p.t1; ++t) {
        // operands that do not depend on the exchange: requested first
        const float4 pre4 = *reinterpret_cast<const float4*>(p.pre + ((size_t)t * B + rowc) * N + 4 * u);
        const int hm = (p.hmask && cellthr) ? (int)p.hmask[((size_t)t * B + row) * H + u] : 1;
        const int cm = (p.cmask && cellthr) ? (int)p.cmask[((size_t)t * B + row) * H + u] : 1;
        if (PREC) ps_gates_bf16<4, RT>(xregion(t & 1), nkb, reinterpret_cast<const uint4*>(wl), red);
        else ps_gates<4, RT, 4>(xregion(t & 1), nkb, wl, red);
        __syncthreads();
        float4 ga;
        if (cellthr) ps_cell(red, row, uu, bias4, pre4, c_state, h_state, hm, cm, p.hmask != nullptr, p.cell, ga);
        const float4 h4 = ps_quad_gather(h_state);
        if (cellthr) {
            const size_t o = ((size_t)(t + 1) * B + row) * H + u;
            p.c[o] = c_state;
            if (uu == 0) {
                *reinterpret_cast<float4*>(p.h + o) = h4;
                ps_publish4<PREC>(xregion((t + 1) & 1), row, 4 * c, nkb, h4);
            }
            if (p.gates) {
                float* go = p.gates + ((size_t)t * B + row) * N + u;
                go[0] = ga.x; go[H] = ga.y; go[2 * H] = ga.z; go[3 * H] = ga.w;
            }
        }
        if (!ps_barrier(p.sync, ++epoch)) return;       // also orders the reads of red before the next step's writes
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// helpers of the pipelined (dataflow) generator kernel below
// ---------------------------------------------------------------------------------------------------------------------------
struct PsBar2 { unsigned* cnt; unsigned* err; };      // cnt[(g * 8 + x) * 32]: counter x of row group g


// ---------------------------------------------------------------------------------------------------------------------------
// pgen7: DATAFLOW form of the generator recurrence.  Workgroup = 8 MULTIPLIER waves + one SERVICE wave per 16-row group (12 waves):
//   multiplier wave w  owns the k-blocks [nkb w / 8, nkb (w + 1) / 8) for good: its weight planes (bf16 x 3) live in its REGISTERS for
//                      the whole decode (no LDS or global weight traffic per step at all); per group-step it splits and multiplies the
//                      prefetched fp32 fragments, refills each fragment register with the next group-step's data the moment its MFMAs
//                      are issued, leaves its partial sums in LDS and bumps the group's `done` counter (LDS atomic);
//   service wave g     waits for done[g] (LDS), runs the LSTM cell of its 16 rows x 4 units, publishes h (fp32 XP quanta, write-through),
//                      drains, arrives at the group's grid counters, polls them and posts `seen[g]` (LDS) when the publish has landed
//                      from every workgroup - the multipliers only ever look at LDS.
// No s_barrier after start-up: every wait is a data-flow condition, so the grid-barrier latency of one row group is covered by the
// other groups' arithmetic.  (scripts/mb/pgen4_variant.inc: the same kernel with a bf16-plane exchange, for bit-equality checks.)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int PS4_THREADS = 768;

template <int NG>
__global__ __launch_bounds__(PS4_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void pgen7_kernel(PsGen p) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    constexpr int NBW = 4;                                    // k-blocks per multiplier wave (nkb = 32)
    const int tid = threadIdx.x, lane = tid & 63, c = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, B = p.B, nkb = H >> 5, N = 4 * H;
    float* red = reinterpret_cast<float*>(psm);                                        // [NG][8][16][16]
    volatile unsigned* done = reinterpret_cast<volatile unsigned*>(red + NG * 8 * 256);   // [4]  multiplier waves finished, per group (monotonic)
    volatile unsigned* seen = done + 4;                                                // [4]  publish number that has landed, per group
    volatile unsigned* lerr = done + 8;
    const PsBar2 bar{p.sync.cnt, p.sync.err};
    if (tid < 12) done[tid] = 0;
    __syncthreads();
    const unsigned per_pub = PS_WGS / 8;
    const int n_steps = p.t1 - p.t0, n_gs = n_steps * NG;
    const unsigned xg_bytes = (unsigned)(nkb * 2048);      // fp32 A-fragment image of 16 rows (XP layout, rt = 0)
    auto xregion = [&](int g, int par) { return ps_rsrc(reinterpret_cast<char*>(p.xp) + (size_t)(g * 2 + par) * xg_bytes, xg_bytes); };

    if (wave >= 8) {
        // =========================== service wave of row group g ===========================
        const int g = wave - 8;
        if (g >= NG) return;
        const int rl = lane >> 2, uu = lane & 3, u = 4 * c + uu, row = 16 * g + rl;
        const bool valid = row < B;
        const int rowc = valid ? row : 0;
        const float4 bias4 = *reinterpret_cast<const float4*>(p.bias_u + 4 * u);
        float c_state = valid ? p.c[((size_t)p.t0 * B + row) * H + u] : 0.f;
        float h_state = valid ? p.h[((size_t)p.t0 * B + row) * H + u] : 0.f;
        // Two-level arrive (as ps_barrier): the counters that take 32 atomics per publish are polled by nobody; the polled word
        // (top counter of the group) takes 8.  sub[g][x] = cnt[(g * 9 + 1 + x) * 32], top[g] = cnt[g * 9 * 32].
        unsigned* top = bar.cnt + (g * 9) * 32;
        unsigned* sub = bar.cnt + (g * 9 + 1 + (blockIdx.x & 7)) * 32;
        auto arrive = [&](unsigned pub) {
            if (lane == 0) {
                const unsigned prev = __hip_atomic_fetch_add(sub, 1u, PS_RLX, PS_AGENT);
                if (prev + 1 == pub * per_pub) __hip_atomic_fetch_add(top, 1u, PS_RLX, PS_AGENT);
            }
        };
        auto land = [&](unsigned pub) -> bool {      // wait until publish `pub` of this group has arrived everywhere, then tell the multipliers
            unsigned spins = 0;
            const unsigned limit = ps_spin_limit(pub);
            while (__hip_atomic_load(top, PS_RLX, PS_AGENT) < pub * 8u) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0 && (spins > limit || __hip_atomic_load(bar.err, PS_RLX, PS_AGENT) != 0)) {
                    if (lane == 0) { __hip_atomic_store(bar.err, 2u, PS_RLX, PS_AGENT); *lerr = 1; }
                    return false;
                }
            }
            if (lane == 0) seen[g] = pub;
            return true;
        };
        {
            const float4 h4 = ps_quad_gather(h_state);
            if (valid && uu == 0) ps_st16_sc1(xregion(g, p.t0 & 1), ps_xp_off(rl, 4 * c, nkb), h4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            arrive(1u);
            if (!land(1u)) return;
        }
        for (int s = 0; s < n_steps; ++s) {
            const int t = p.t0 + s;
            const float4 pre4 = *reinterpret_cast<const float4*>(p.pre + ((size_t)t * B + rowc) * N + 4 * u);
            const size_t mo = ((size_t)t * B + rowc) * H + u;
            const unsigned hm = p.hmask ? (unsigned)p.hmask[mo] : 1u, cm = p.cmask ? (unsigned)p.cmask[mo] : 1u;
            {   // the eight partial sums of this group-step are in LDS
                unsigned spins = 0;
                while (done[g] < 8u * (unsigned)(s + 1)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (*lerr != 0 || ++spins > (PS_SPIN_MAX << 2)) { if (lane == 0) { *lerr = 1; __hip_atomic_store(bar.err, 2u, PS_RLX, PS_AGENT); } return; }
                }
            }
            const float* redg = red + g * (8 * 256);
            float4 g4 = bias4;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float4 v = *reinterpret_cast<const float4*>(redg + w * 256 + rl * 16 + 4 * uu);
                g4.x += v.x; g4.y += v.y; g4.z += v.z; g4.w += v.w;
            }
            g4.x += pre4.x; g4.y += pre4.y; g4.z += pre4.z; g4.w += pre4.w;
            const float ig = sigmoidf_(g4.x), fg = sigmoidf_(g4.y), gg = tanhf_(g4.z), og = sigmoidf_(g4.w);
            const float cp = c_state, hp = h_state;
            const float cn = fg * cp + ig * gg;
            const float hn = og * tanhf_(cn);
            float ho, co = cn;
            if (p.cell.zone == 1) { ho = hm ? hn : hp; co = cm ? cn : cp; }
            else if (p.cell.zone == 2) { ho = p.cell.zh * hp + (1.f - p.cell.zh) * hn; co = p.cell.zc * cp + (1.f - p.cell.zc) * cn; }
            else ho = p.hmask ? (hm ? hn * p.cell.hscale : 0.f) : hn;
            c_state = co; h_state = ho;
            const float4 h4 = ps_quad_gather(h_state);
            if (valid && uu == 0) ps_st16_sc1(xregion(g, (t + 1) & 1), ps_xp_off(rl, 4 * c, nkb), h4);      // ONE 16-byte fp32 quantum (the consumers split)
            if (s + 1 < n_steps) {      // exchange first: drain the write-through stores, arrive; the saved state follows
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                arrive((unsigned)(s + 2));
            }
            if (valid) {
                const size_t o = ((size_t)(t + 1) * B + row) * H + u;
                if (uu == 0) *reinterpret_cast<float4*>(p.h + o) = h4;
                p.c[o] = c_state;
                if (p.gates) {
                    float* go = p.gates + ((size_t)t * B + row) * N + u;
                    go[0] = ig; go[H] = fg; go[2 * H] = gg; go[3 * H] = og;
                }
            }
            if (s + 1 < n_steps && !land((unsigned)(s + 2))) return;
        }
        return;
    }

    // =========================== multiplier wave ===========================
    const int k0 = (nkb * wave) >> 3, k1 = (nkb * (wave + 1)) >> 3;
    PsFrag wreg[NBW][3];
    {
        const float4* src = reinterpret_cast<const float4*>(p.w_packed) + (size_t)c * nkb * 128;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int kb = min(k0 + j, nkb - 1);
            const PsFrag3 f = ps_split8(src[(kb * 2 + 0) * 64 + lane], src[(kb * 2 + 1) * 64 + lane]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wreg[j][pl] = f.p[pl];
        }
    }
    auto wait_seen = [&](int g, unsigned pub) -> bool {
        unsigned spins = 0;
        const unsigned limit = pub <= 1u ? 0xfffffff0u : (PS_SPIN_MAX << 2);      // publish 1: the service wave's own (patient) wait decides
        while (seen[g] < pub) {
            __builtin_amdgcn_s_sleep(1);
            if (*lerr != 0 || ++spins > limit) { *lerr = 1; return false; }
        }
        return true;
    };
    float4 xb[NBW][2];                                        // fp32 fragments of the next group-step (two 16-byte halves per k-block)
    auto issue_blk = [&](int j, __amdgpu_buffer_rsrc_t r) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) xb[j][hf] = ps_ld16_sc1(r, (unsigned)((((k0 + j) * 2 + hf) * 1024) + lane * 16));
    };
    auto issue_all = [&](__amdgpu_buffer_rsrc_t r) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) issue_blk(j, r);
    };
    if (n_gs > 0) {
        if (!wait_seen(0, 1u)) return;
        issue_all(xregion(0, p.t0 & 1));
    }
    const int i16 = lane & 15, q4 = lane >> 4;
    for (int i = 0; i < n_gs; ++i) {
        const int g = i % NG, in = i + 1, gn = in % NG, tn = p.t0 + in / NG;
        const bool has_next = in < n_gs;
        const unsigned pubn = (unsigned)(in / NG + 1);
        const bool early = has_next && seen[gn] >= pubn;
        const __amdgpu_buffer_rsrc_t nxr = xregion(gn, tn & 1);
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const PsFrag3 f3 = ps_split8(xb[j][0], xb[j][1]);         // exact 3-way split here: 4 B per element over the L2 instead of 6
            const PsFrag (&a)[3] = f3.p;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, wreg[j][0].v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, wreg[j][2].v, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, wreg[j][1].v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, wreg[j][0].v, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, wreg[j][1].v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, wreg[j][0].v, acc1, 0, 0, 0);
            if (early) issue_blk(j, nxr);
            __builtin_amdgcn_sched_barrier(0);
        }
        float* rw = red + (g * 8 + wave) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[(4 * q4 + r) * 16 + i16] = acc0[r] + acc1[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned*>(done) + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (has_next && !early) {
            if (!wait_seen(gn, pubn)) return;
            issue_all(nxr);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// pdec: attention LSTM + location-sensitive attention of the teacher-forced schedule, all steps in one launch (B <= 64).
// Per step two phases with one grid barrier after each (see the file header):
//   phase 1  every workgroup in its COLUMN role (LSTM units [4c, 4c+4)): gates over K = Dm + H from the exchange buffer [ctx_t | h_t]
//            (weights fp32 in LDS, exact 3-way bf16 split products), cell, h_{t+1} -> row-major (write-through) + exchange layout
//   phase 2  workgroups (b, j), b < B, in their SAMPLE role: query slice q[32j .. 32j+32) = W_q h_{t+1}[b] (W_q slice streamed from L2
//            into registers behind the barrier), location features on MFMA, partial energies over the 32 channels, 4-way exchange of the
//            partial energies through tagged granules, softmax (every workgroup of the sample computes the same weights in the same
//            order), cumulative alignment kept in LDS, context columns [j Dm/4, (j+1) Dm/4) -> row-major + exchange layout
// ---------------------------------------------------------------------------------------------------------------------------
struct PsDec {
    int B, L, H, A, Dm, ksz, t0, t1;
    const float* w_packed;      // mtts_lstm_pack_weights(fp32) of [W_ih[:, P:] | W_hh]
    const float* bias_u;        // [4H] unit-major
    const float* pre;           // [T][B][4H] unit-major hoisted prenet projection
    float* h; float* c;         // [T+1][B][H]
    float* gates;               // [T][B][4H] or NULL
    const uint8_t* hmask; const uint8_t* cmask;
    PsCellCfg cell;
    const float* w_query;       // [A][H]
    const float* memory;        // [B][L][Dm]
    const float* Mt;            // [B][L][A]
    const float* U;             // [A][ksz]
    const float* att_bias; const float* v;      // [A]
    const int* lengths;         // [B]
    float* ctx;                 // [T+1][B][Dm]
    float* cum;                 // [T+1][B][L]
    float* align;               // [T][B][L]
    float* q_all;               // [T][B][A] or NULL
    float* xp;                  // [2][64 rows][Dm + H] exchange (XP layout)
    unsigned long long* eg;     // [64][4][128] partial-energy granules {tag, value}
    PsSync sync;
    int poll_h;                 // 1: the sample role polls its h row (pre-filled with PS_SENTINEL by the host) instead of waiting for the barrier
    int early_h;                // 1: the h-part fragments of the next step's gates are requested inside the attention step (after the energy exchange,
                                //    once the h barrier is seen complete) instead of behind the context barrier's arrive
};

constexpr unsigned PS_SENTINEL = 0xffffffffu;      // a NaN pattern no arithmetic produces
constexpr int PD_LMAX = 128;         // encoder positions of one position tile (8 waves x 16 rows)
constexpr int PD_LT_MAX = 3;         // position tiles a sample workgroup can take: inputs of up to 384 characters (reference data: max 304)
constexpr int PD_NCM = 9;            // memory float4 per thread and position tile in the context phase

// LT = position tiles of 128 encoder positions.  LT == 1 (L <= 128) is the fast instantiation: the sample's Mt slice sits in LDS and
// the h-part fragments of the next gates are requested inside the attention step.  LT > 1 (round 5: the reference attends over any L,
// modules/attention.py:39-45; real CSS10 batches exceed 128 characters): LDS has no room for LT x 16 KiB of Mt beside the weight slice,
// so every lane re-requests the 8 x LT Mt values it needs (L2-resident: 40 KiB per workgroup at L = 320) together with the query
// weights, the energies / softmax / context stages loop over the tiles, the memory rows of the context are LT x 9 float4 per thread,
// and the early h-part requests are off (their registers hold the memory rows).
template <int RT, int PREC, int LT = 1>
__global__ __launch_bounds__(PS_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void pdec_kernel(PsDec p) {
    constexpr int LM = PD_LMAX * LT;      // positions the LDS / granule rows are sized for
    extern __shared__ __attribute__((aligned(16))) char psm[];
    const int tid = threadIdx.x, lane = tid & 63, c = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, B = p.B, L = p.L, A = p.A, Dm = p.Dm, K = Dm + H, nkb = K >> 5, N = 4 * H;
    const int pad = (p.ksz - 1) >> 1;
    // ---- LDS carve
    float4* wl = reinterpret_cast<float4*>(psm);                                       // [nkb][2][64] float4
    float* red = reinterpret_cast<float*>(psm + (size_t)nkb * (PREC ? 1024 : 2048));      // [8][64][16] (phase 1) / scratch (phase 2)
    float* Mt_s = red + 8 * 64 * 16;                                                    // [PD_LMAX][32] (LT == 1 only)
    float* cumw = Mt_s + (LT == 1 ? PD_LMAX * 32 : 0);                                  // [LM + 64] cumulative alignment with zero halo
    uint4* Upl = reinterpret_cast<uint4*>(cumw + LM + 64);                              // [2 col tiles][3 planes][64 lanes]
    float* vb = reinterpret_cast<float*>(Upl + 2 * 3 * 64);                             // v[32], bias[32]
    volatile unsigned* eflag = reinterpret_cast<volatile unsigned*>(vb + 64);           // [4] workgroup-uniform decision of thread 0 (early h-part loads)
    // phase-2 scratch inside `red`
    float* hs = red;                                   // [H] query operand: the sample's h row
    float* qs = hs + 1024;                             // [32]
    float* es = qs + 32;                               // [4][LM] partial energies of the four slices
    float* wsm = es + 4 * LM;                          // [LM] alignment weights
    float* ctxp = wsm + LM;                            // [ng][Dq] context partial sums

    // ---- roles
    const int row = tid >> 2, uu = tid & 3, u = 4 * c + uu;          // column role: cell thread (row, unit), tid < 4 B
    const bool cellthr = tid < 4 * B;
    const int rowc = cellthr ? row : 0;
    const int sb = c >> 2, sj = c & 3;                                 // sample role: sample sb, slice sj
    const bool has_sample = sb < B;
    const int sbc = has_sample ? sb : 0;
    const int Dq = Dm >> 2, d0 = sj * Dq, nc4 = Dq >> 2, ng = PS_THREADS / nc4;
    const int len = min(p.lengths[sbc], L);
    const unsigned xp_bytes = (unsigned)(64 * K * (PREC ? 2 : 4));
    auto xregion = [&](int par) { return ps_rsrc(reinterpret_cast<char*>(p.xp) + (size_t)par * xp_bytes, xp_bytes); };

    // ---- stationary data -> LDS
    {
        const int per = PREC ? 64 : 128;                                // 16-byte quanta per k-block (bf16-packed / fp32-packed weights)
        const float4* src = reinterpret_cast<const float4*>(p.w_packed) + (size_t)c * nkb * per;
        for (int i = tid; i < nkb * per; i += PS_THREADS) wl[i] = src[i];
        if (LT == 1) for (int i = tid; i < L * 32; i += PS_THREADS) { const int l = i >> 5, a = i & 31; Mt_s[i] = p.Mt[((size_t)sbc * L + l) * A + 32 * sj + a]; }
        for (int i = tid; i < LM + 64; i += PS_THREADS) {
            const int l = i - pad;
            cumw[i] = (has_sample && l >= 0 && l < L) ? p.cum[((size_t)p.t0 * B + sb) * L + l] : 0.f;
        }
        if (tid < 2 * 64) {      // location filter bank slice as MFMA B fragments: lane (i16 = channel, q4) holds taps 8 q4 .. 8 q4 + 7
            const int ct = tid >> 6, l2 = tid & 63, a = 32 * sj + 16 * ct + (l2 & 15), q4 = l2 >> 4;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int k = 8 * q4 + e; f[e] = k < p.ksz ? p.U[(size_t)a * p.ksz + k] : 0.f; }
            const PsFrag3 fr = ps_split8(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) Upl[(ct * 3 + pl) * 64 + l2] = make_uint4(fr.p[pl].u[0], fr.p[pl].u[1], fr.p[pl].u[2], fr.p[pl].u[3]);
        }
        if (tid < 32) { vb[tid] = p.v[32 * sj + tid]; vb[32 + tid] = p.att_bias[32 * sj + tid]; }
    }
    float c_state = 0.f, h_state = 0.f;
    const float4 bias4 = *reinterpret_cast<const float4*>(p.bias_u + 4 * u);
    if (cellthr) {
        c_state = p.c[((size_t)p.t0 * B + row) * H + u];
        h_state = p.h[((size_t)p.t0 * B + row) * H + u];
    }
    {   // exchange buffer of step t0: [ctx[t0] | h[t0]]
        const float4 h4 = ps_quad_gather(h_state);
        if (cellthr && uu == 0) ps_publish4<PREC>(xregion(p.t0 & 1), row, Dm + 4 * c, nkb, h4);
        if (has_sample && tid < nc4) {
            const float4 c4v = *reinterpret_cast<const float4*>(p.ctx + ((size_t)p.t0 * B + sb) * Dm + d0 + 4 * tid);
            ps_publish4<PREC>(xregion(p.t0 & 1), sb, d0 + 4 * tid, nkb, c4v);
        }
    }
    unsigned epoch = 0;
    if (!ps_barrier(p.sync, ++epoch)) return;
    // The gate GEMM of a step is split by operand: the h-part (k-blocks [Dm/32, nkb): h_t is known one barrier earlier than ctx_t)
    // runs in the shadow of the barrier that publishes ctx_t; only the ctx-part sits between that barrier and the cell.
    const int kb_ctx = Dm >> 5;
    f32x4 acc[RT];
    float4 xf[3][RT][2];         // fragment registers of the gate products (fp32 exchange: 3 k-blocks in flight; bf16: all 4 of a wave's share)
    u32x4 xb[4][RT];
    // ctx-part (k-blocks [0, kb_ctx), at most 3 per wave) / h-part (k-blocks [kb_ctx, nkb), 4 per wave) of the gates of step tt
    auto gates_ctx = [&](int tt) {
        if (PREC) { u32x4 x3[3][RT]; ps_gates_bf16_acc<3, RT>(xregion(tt & 1), nkb, 0, kb_ctx, reinterpret_cast<const uint4*>(wl), acc, x3, false); }
        else ps_gates_acc<3, RT, 3>(xregion(tt & 1), nkb, 0, kb_ctx, wl, acc, xf, false);
    };
    constexpr int NPRE = RT >= 4 ? 2 : 3;      // k-blocks requested early (four row tiles: two, the register file has no room for a third)
    auto gates_h_prefetch = [&](int tt) {
        if (PREC) ps_gates_bf16_prefetch<4, RT>(xregion(tt & 1), nkb, kb_ctx, nkb, xb);
        else ps_gates_prefetch<4, RT, 3, 0, NPRE>(xregion(tt & 1), nkb, kb_ctx, nkb, xf);
    };
    auto gates_h = [&](int tt, bool preloaded) {
        if (PREC) ps_gates_bf16_acc<4, RT>(xregion(tt & 1), nkb, kb_ctx, nkb, reinterpret_cast<const uint4*>(wl), acc, xb, preloaded);
        else ps_gates_acc<4, RT, 3, NPRE>(xregion(tt & 1), nkb, kb_ctx, nkb, wl, acc, xf, preloaded);
    };
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gates_h(p.t0, false);

    for (int t = p.t0; t < p.t1; ++t) {
        // ================= phase 1: attention LSTM (column role) =================
        const float4 pre4 = *reinterpret_cast<const float4*>(p.pre + ((size_t)t * B + rowc) * N + 4 * u);
        const size_t mo = ((size_t)t * B + rowc) * H + u;
        const unsigned hm = p.hmask ? (unsigned)p.hmask[mo] : 1u, cm = p.cmask ? (unsigned)p.cmask[mo] : 1u;
        gates_ctx(t);
        ps_gates_store<RT>(acc, red);
        __syncthreads();
        float4 ga;
        if (cellthr) ps_cell(red, row, uu, bias4, pre4, c_state, h_state, (int)hm, (int)cm, p.hmask != nullptr, p.cell, ga);
        {
            const float4 h4 = ps_quad_gather(h_state);
            if (cellthr && uu == 0) {
                ps_publish4<PREC>(xregion((t + 1) & 1), row, Dm + 4 * c, nkb, h4);
                // row-major copy (write-through): the sample role reads rows of it after the barrier
                ps_st16_sc1(ps_rsrc(p.h + (size_t)(t + 1) * B * H, (unsigned)(B * H * 4)), (unsigned)((row * H + u) * 4), h4);
            }
        }
        ps_bar_arrive(p.sync, ++epoch);
        constexpr int NMEM = PD_NCM * (LT < 2 ? LT : 2);      // memory rows per thread requested ahead; LT == 3: the third tile's rows are requested inside the context stage
        float4 wq4[16]; float4 mem4[NMEM];
        float mtr[LT > 1 ? LT : 1][2][4];      // LT > 1: the lane's Mt values of the energy stage (position tile, channel tile, row)
        // operands of phase 2 that do not depend on h_{t+1}.  Query weights: requested behind the arrive, landing while the barrier
        // completes (streamed from L2 every step: neither LDS nor the register file has 128 KiB to spare beside phase 1):
        // lane (a = tid >> 4, l16 = tid & 15) takes k = 64 i + 4 l16 .. + 4, i.e. 256 contiguous bytes per channel and load
        {
            const float* wq = p.w_query + (size_t)(32 * sj + (tid >> 4)) * H + 4 * (tid & 15);
#pragma unroll
            for (int i = 0; i < 16; ++i) wq4[i] = *reinterpret_cast<const float4*>(wq + 64 * i);
            if (LT > 1) {
                // one descriptor over the sample's [L][A] slice, ONE per-lane offset, the position tile in the scalar offset, (row, channel
                // tile) as immediates: 8 LT requests without 8 LT loop-invariant 64-bit addresses held in registers across the step loop.
                // The position tile sits in the PER-LANE offset (round 6): the hardware's range check covers the vector offset, so positions
                // >= L read as zero without touching memory whatever a target does with the scalar offset (their energies are never used).
                const __amdgpu_buffer_rsrc_t mt_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Mt + (size_t)sbc * L * A), 0, L * A * 4, 0x00020000);
                const unsigned mt_v = (unsigned)(((4 * (lane >> 4)) * A + 32 * sj + (lane & 15)) * 4);
#pragma unroll
                for (int it = 0; it < LT; ++it) {
                    const unsigned mt_s = (unsigned)(16 * (wave + 8 * it) * A * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            mtr[it][ct][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mt_r, mt_v + mt_s + (unsigned)((r * A + 16 * ct) * 4), 0, 0));
                }
            }
        }
        // memory columns of the context: requested after the query (they are needed three stages later; keeping them out of the query's
        // register budget leaves room for the early h-part fragments)
        auto pd_prefetch_mem = [&]() {
            const int c4 = tid % nc4, lg = tid / nc4;
            if (LT == 1) {
                const float* mem = p.memory + (size_t)sbc * L * Dm + d0 + 4 * c4;
#pragma unroll
                for (int i = 0; i < NMEM; ++i) {
                    const int l = min(lg + i * ng, L - 1);
                    mem4[i] = *reinterpret_cast<const float4*>(mem + (size_t)l * Dm);
                }
            } else {      // descriptor over the sample's [L][Dm] rows; the row group is part of the per-lane offset, which the hardware range-checks: rows >= L read as zero
                const __amdgpu_buffer_rsrc_t mem_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.memory + (size_t)sbc * L * Dm), 0, L * Dm * 4, 0x00020000);
                const unsigned mem_v = (unsigned)((lg * Dm + d0 + 4 * c4) * 4), mem_step = (unsigned)(ng * Dm * 4);
#pragma unroll
                for (int i = 0; i < NMEM; ++i) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(mem_r, mem_v + (unsigned)i * mem_step, 0, 0);
                    mem4[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            }
        };
        if (cellthr) {
            const size_t o = ((size_t)(t + 1) * B + row) * H + u;
            p.c[o] = c_state;
            if (p.gates) {
                float* go = p.gates + ((size_t)t * B + row) * N + u;
                go[0] = ga.x; go[H] = ga.y; go[2 * H] = ga.z; go[3 * H] = ga.w;
            }
        }
        // h_{t+1} has to be complete for the h-part of the next gates (every row) and for the query (the own sample's row only).
        // poll_h: the query does not wait for the grid barrier - the h rows of steps (t0, t1] were pre-filled with a sentinel by the
        // host, every 16-byte quantum of a row is written by ONE store of its owner, and the four waves that fetch the row simply
        // re-read it until no sentinel is left (the payload is the flag: one store -> load hop instead of drain -> two-level arrive ->
        // poll -> fetch).  The barrier itself is then waited for after the attention step, where it has long completed.
        if (!p.poll_h && !ps_bar_wait(p.sync, epoch)) return;
        bool early = false;          // h-part fragments of step t + 1 already requested (workgroup-uniform)

        // ================= phase 2: attention (sample role) =================
        if (has_sample) {
            // ---- h_{t+1}[sb] -> LDS (16 chunks of 64 with 4 floats of padding: conflict-free reads below)
            {
                const __amdgpu_buffer_rsrc_t hr = ps_rsrc(p.h + ((size_t)(t + 1) * B + sb) * H, (unsigned)(H * 4));
                if (tid < (H >> 2)) {
                    float4 v = ps_ld16_sc1(hr, tid * 16);
                    if (p.poll_h) {
                        unsigned spins = 0;
                        while (__builtin_amdgcn_ballot_w64(__float_as_uint(v.x) == PS_SENTINEL || __float_as_uint(v.y) == PS_SENTINEL ||
                                                           __float_as_uint(v.z) == PS_SENTINEL || __float_as_uint(v.w) == PS_SENTINEL) != 0ull) {
                            __builtin_amdgcn_s_sleep(1);
                            v = ps_ld16_sc1(hr, tid * 16);
                            if (++spins > (PS_SPIN_MAX >> 3)) { __hip_atomic_store(p.sync.err, 2u, PS_RLX, PS_AGENT); break; }      // every retry is a ~1 us round trip: ~0.5 s
                        }
                    }
                    *reinterpret_cast<float4*>(hs + 4 * tid) = v;
                }
            }
            __syncthreads();
            // ---- query slice: thread (a = tid >> 4, chunk = tid & 15)
            {
                const float* hc = hs + 4 * (tid & 15);
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float4 h4 = *reinterpret_cast<const float4*>(hc + 64 * i);
                    acc += wq4[i].x * h4.x + wq4[i].y * h4.y + wq4[i].z * h4.z + wq4[i].w * h4.w;
                }
                acc = row16_sum(acc);
                if ((tid & 15) == 0) {
                    qs[tid >> 4] = acc;
                    if (p.q_all) p.q_all[((size_t)t * B + sb) * A + 32 * sj + (tid >> 4)] = acc;
                }
            }
            if (LT > 1) __builtin_amdgcn_sched_barrier(0);      // keep the memory rows' registers behind the query weights' (no hoisting above the query)
            pd_prefetch_mem();
            __syncthreads();
            // ---- location features (MFMA, exact split) + partial energies: wave w <-> positions [16 w, 16 w + 16)
#pragma unroll
            for (int it = 0; it < LT; ++it) {
                const int i16 = lane & 15, q4 = lane >> 4, l0 = 16 * (wave + 8 * it);
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = cumw[l0 + i16 + 8 * q4 + e];
                const PsFrag3 af = ps_split8(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]));
                float e4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    PsFrag3 bf;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const uint4 w = Upl[(ct * 3 + pl) * 64 + lane];
                        bf.p[pl].u[0] = w.x; bf.p[pl].u[1] = w.y; bf.p[pl].u[2] = w.z; bf.p[pl].u[3] = w.w;
                    }
                    const f32x4 loc = ps_mma6(af, bf, (f32x4){0.f, 0.f, 0.f, 0.f});
                    const int a = 16 * ct + i16;
                    const float qa = qs[a] + vb[32 + a], va = vb[a];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int l = min(l0 + 4 * q4 + r, PD_LMAX - 1);
                        e4[r] += va * tanhf_(qa + (LT == 1 ? Mt_s[l * 32 + a] : mtr[it][ct][r]) + loc[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = row16_sum(e4[r]);
                    const int l = l0 + 4 * q4 + r;
                    if (i16 == 0 && l < L) {
                        es[sj * LM + l] = e;
                        __hip_atomic_store(p.eg + ((size_t)(sb * 4 + sj)) * LM + l, ((unsigned long long)epoch << 32) | __float_as_uint(e), PS_RLX, PS_AGENT);
                    }
                }
                if (LT > 1) __builtin_amdgcn_sched_barrier(0);      // one position tile at a time (register budget)
            }
            // ---- the other three slices' partial energies (tagged granules: the data is the flag)
            for (int idx = tid; idx < 3 * LM; idx += PS_THREADS) {
                const int o = idx / LM, l = idx - o * LM, j2 = o + (o >= sj ? 1 : 0);
                if (l < L) {
                    unsigned long long* gp = p.eg + ((size_t)(sb * 4 + j2)) * LM + l;
                    unsigned long long x = __hip_atomic_load(gp, PS_RLX, PS_AGENT);
                    unsigned spins = 0;
                    while ((unsigned)(x >> 32) != epoch) {
                        __builtin_amdgcn_s_sleep(1);
                        x = __hip_atomic_load(gp, PS_RLX, PS_AGENT);
                        if (++spins > PS_SPIN_MAX) { __hip_atomic_store(p.sync.err, 2u, PS_RLX, PS_AGENT); break; }
                    }
                    es[j2 * LM + l] = __uint_as_float((unsigned)x);
                }
            }
            // Early h-part loads: by now the grid barrier of the h hand-off has normally completed (it trails the polled row by about
            // one energy stage); thread 0 looks ONCE, without waiting.  If it has, every wave requests its first three h-part k-blocks
            // of the next step's gates here - they travel while the softmax and the context run - and the barrier's wait is skipped.
            if (tid == 0) eflag[0] = (LT == 1 && p.early_h && t + 1 < p.t1 && (!p.poll_h || __hip_atomic_load(p.sync.cnt, PS_RLX, PS_AGENT) >= epoch * 8u)) ? 1u : 0u;
            __syncthreads();
            early = LT == 1 && eflag[0] != 0u;
            if (LT == 1) { if (early) gates_h_prefetch(t + 1); }
            // ---- masked softmax over the positions, cumulative alignment.  EVERY wave computes it (the others would idle at the barrier
            //      below), wave 0 stores: with the DPP reductions of common.h inside an `if (wave == 0)` region this compiler rejects the
            //      kernel ("Illegal instruction detected: Operand has incorrect register class", ROCm 7.2), and the shuffle butterflies
            //      they replace were ~1.4 k of this stage's 3.3 k cycles
            {
                float e0[2 * LT], mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < 2 * LT; ++k) {
                    const int l = lane + 64 * k;
                    e0[k] = (l < L) ? ((es[l] + es[LM + l]) + (es[2 * LM + l] + es[3 * LM + l])) : 0.f;
                    if (l < len) mx = fmaxf(mx, e0[k]);
                }
                mx = wave_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < 2 * LT; ++k) { const int l = lane + 64 * k; e0[k] = (l < len) ? __expf(e0[k] - mx) : 0.f; sum += e0[k]; }
                sum = wave_sum(sum);
                const float inv = 1.f / sum;
#pragma unroll
                for (int k = 0; k < 2 * LT; ++k) {
                    const int l = lane + 64 * k;
                    if (l < L && wave == 0) {
                        const float wl_ = e0[k] * inv, cn = cumw[pad + l] + wl_;
                        wsm[l] = wl_; cumw[pad + l] = cn;       // their copies in HBM are written by the last wave after the context stage
                    }
                }
            }
            __syncthreads();
            // ---- context columns of this slice
            {
                const int c4 = tid % nc4, lg = tid / nc4;
                float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < NMEM; ++i) {
                    const int l = lg + i * ng;
                    const float wl_ = (lg < ng && l < L) ? wsm[l] : 0.f;
                    s4.x += wl_ * mem4[i].x; s4.y += wl_ * mem4[i].y; s4.z += wl_ * mem4[i].z; s4.w += wl_ * mem4[i].w;
                }
                if (LT > 2) {      // rows of the third position tile (inputs above 256 characters): one more round trip, in index order
                    __builtin_amdgcn_sched_barrier(0);      // the requests below reuse the registers of the rows consumed above
                    const __amdgpu_buffer_rsrc_t mem_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.memory + (size_t)sbc * L * Dm), 0, L * Dm * 4, 0x00020000);
                    const unsigned mem_v = (unsigned)((lg * Dm + d0 + 4 * c4) * 4), mem_step = (unsigned)(ng * Dm * 4);
                    float4 m3[PD_NCM];
#pragma unroll
                    for (int i = 0; i < PD_NCM; ++i) {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(mem_r, mem_v + (unsigned)(NMEM + i) * mem_step, 0, 0);
                        m3[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                    }
#pragma unroll
                    for (int i = 0; i < PD_NCM; ++i) {
                        const int l = lg + (NMEM + i) * ng;
                        const float wl_ = (lg < ng && l < L) ? wsm[l] : 0.f;
                        s4.x += wl_ * m3[i].x; s4.y += wl_ * m3[i].y; s4.z += wl_ * m3[i].z; s4.w += wl_ * m3[i].w;
                    }
                }
                if (lg < ng) *reinterpret_cast<float4*>(ctxp + (lg * nc4 + c4) * 4) = s4;
            }
            __syncthreads();
            // alignment / cumulative alignment of this step -> HBM (saved for the backward): off wave 0's softmax chain, where the two
            // stores queued behind the early h-part loads
            if (sj == 0 && wave == PS_THREADS / 64 - 1) {
#pragma unroll
                for (int k = 0; k < 2 * LT; ++k) {
                    const int l = lane + 64 * k;
                    if (l < L) { p.align[((size_t)t * B + sb) * L + l] = wsm[l]; p.cum[((size_t)(t + 1) * B + sb) * L + l] = cumw[pad + l]; }
                }
            }
            if (tid < nc4) {
                float4 tt = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = 0; k < ng; ++k) { const float4 v4 = *reinterpret_cast<const float4*>(ctxp + (k * nc4 + tid) * 4); tt.x += v4.x; tt.y += v4.y; tt.z += v4.z; tt.w += v4.w; }
                *reinterpret_cast<float4*>(p.ctx + ((size_t)(t + 1) * B + sb) * Dm + d0 + 4 * tid) = tt;
                ps_publish4<PREC>(xregion((t + 1) & 1), sb, d0 + 4 * tid, nkb, tt);
            }
        }
        if (p.poll_h && !early && !ps_bar_wait(p.sync, epoch)) return;       // the h barrier of this step (complete long ago; `early`: seen complete)
        ps_bar_arrive(p.sync, ++epoch);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t + 1 < p.t1) gates_h(t + 1, early);      // h_{t+1} landed one barrier ago
        if (!ps_bar_wait(p.sync, epoch)) return;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
#include <mutex>
#include <map>

// Per-device state of the persistent launches.  A persistent kernel needs ALL its 256 workgroups resident at once (they spin on each
// other); that is (a) checked per device and per kernel instance with the occupancy query (CU count, LDS and register budget of THIS
// device - a CU-masked or partitioned GPU fails it and the per-step launch schedule runs instead), and (b) protected inside one process
// by serialising persistent launches across streams: a launch on stream s first waits for the previous persistent launch of the device
// if that one went to ANOTHER stream (two half-resident grids would starve each other until the bounded spins give up).  Other
// processes on the same GPU cannot be seen from here: share a GPU between processes with MTTS_PERSIST=0.
namespace {
struct PsDevice {
    std::mutex mu;
    int cus = -1;                                   // multiProcessorCount, -1 = not queried yet
    std::map<std::pair<const void*, size_t>, int> ready;   // (kernel, dynamic LDS bytes) -> 1 ok / 0 does not fit (occupancy checked for THAT request)
    std::map<const void*, size_t> lds_attr;         // kernel -> largest MaxDynamicSharedMemorySize set so far (the attribute is per function, not per launch)
    std::mutex launch_mu;                           // held across serialize -> launch -> record of one persistent launch (two host threads, two streams)
    hipEvent_t last_ev = nullptr; hipStream_t last_stream = nullptr; bool have_last = false;
    std::map<const void*, std::pair<unsigned*, unsigned long>> err_of;   // workspace -> (error word its kernels report to, sequence number of the last launch)
    unsigned long err_seq = 0;
};
PsDevice g_ps_dev[64];
PsDevice& ps_dev() { int d = 0; (void)hipGetDevice(&d); return g_ps_dev[d & 63]; }

bool ps_device_ok() {
    PsDevice& D = ps_dev();
    std::lock_guard<std::mutex> lk(D.mu);
    if (D.cus < 0) {
        int dev = 0; hipDeviceProp_t prop;
        D.cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
    }
    return D.cus >= PS_WGS;
}

// dynamic-LDS attribute + co-residency of PS_WGS workgroups of `fn` on the current device, cached per (device, kernel, LDS request):
// one kernel instance serves several LDS sizes (pdec_lds depends on Dm and on the position tiles), so the attribute is raised whenever a
// larger request than any before arrives and the occupancy verdict is kept per request.
bool ps_kernel_ready(const void* fn, int threads, size_t lds) {
    PsDevice& D = ps_dev();
    std::lock_guard<std::mutex> lk(D.mu);
    const auto key = std::make_pair(fn, lds);
    auto it = D.ready.find(key);
    if (it != D.ready.end()) return it->second != 0;
    int ok = 0, per_cu = 0;
    bool attr_ok = D.cus >= PS_WGS;
    auto la = D.lds_attr.find(fn);
    if (attr_ok && (la == D.lds_attr.end() || la->second < lds)) {
        attr_ok = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        if (attr_ok) D.lds_attr[fn] = lds;
    }
    if (attr_ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) == hipSuccess && (long)per_cu * D.cus >= PS_WGS)
        ok = 1;
    (void)hipGetLastError();
    D.ready[key] = ok;
    return ok != 0;
}

// order this persistent launch behind the device's previous one when that went to another stream; call ps_launched() after the launch
int ps_serialize(hipStream_t s) {
    PsDevice& D = ps_dev();
    std::lock_guard<std::mutex> lk(D.mu);
    if (D.have_last && D.last_stream != s) MTTS_CHECK_HIP(hipStreamWaitEvent(s, D.last_ev, 0));
    return 0;
}
int ps_launched(hipStream_t s) {
    PsDevice& D = ps_dev();
    std::lock_guard<std::mutex> lk(D.mu);
    if (!D.last_ev) MTTS_CHECK_HIP(hipEventCreateWithFlags(&D.last_ev, hipEventDisableTiming));
    MTTS_CHECK_HIP(hipEventRecord(D.last_ev, s));
    D.last_stream = s; D.have_last = true;
    return 0;
}
unsigned* ps_err_word(const DecoderArgs& a) {
    unsigned* e = a.persist_err ? (unsigned*)a.persist_err : (unsigned*)((char*)a.persist_ws + PS_ERR_OFF);
    PsDevice& D = ps_dev();
    std::lock_guard<std::mutex> lk(D.mu);
    D.err_of[a.persist_ws] = std::make_pair(e, ++D.err_seq);
    while (D.err_of.size() > 64) {                   // callers allocate a workspace per decode: forget the LEAST recently launched one only
        auto oldest = D.err_of.begin();
        for (auto i = D.err_of.begin(); i != D.err_of.end(); ++i) if (i->second.second < oldest->second.second) oldest = i;
        D.err_of.erase(oldest);
    }
    return e;
}
}  // namespace

// MTTS_PERSIST=0 switches the persistent recurrences off (the per-step launch schedule then runs everywhere)
bool persist_enabled() {
    static const bool on = [] { const char* e = getenv("MTTS_PERSIST"); return !(e && e[0] == '0'); }();
    return on && ps_device_ok();
}

// workspace layout (DecoderArgs.persist_ws): [counters 8 KiB][error word ...16 KiB][generator exchange][attention exchange][granules]
static long ps_ws_gen_off() { return PS_XP_OFF; }
static long ps_ws_att_off(int H) { return ps_ws_gen_off() + 8L * (H / 32) * 3072; }
static long ps_ws_eg_off(int H, int Dm) { return ps_ws_att_off(H) + 2L * 64 * (Dm + H) * 4; }

// bytes of the exchange / synchronisation workspace a decoder call hands to the persistent kernels (DecoderArgs.persist_ws)
MTTS_API long mtts_decoder_persist_ws_bytes(int B, int L, int H, int Dm, int A) {
    (void)B; (void)A; (void)L;
    return ps_ws_eg_off(H, Dm) + 64L * 4 * (PD_LMAX * PD_LT_MAX) * 8 + 1024;
}

bool g_pdec_poll_off = false;                 // harness switch: barrier-only hand-off of h (bit-equality check of the two forms)
bool g_pdec_early_off = false;                // harness switch: h-part fragments requested behind the context barrier's arrive only

// ---- generator LSTM: kernel instance for a shape (fp32: the dataflow kernel pgen7; bf16: one barrier per step, bf16 weights in LDS)
namespace {
struct PsInst { const void* fn; int threads; size_t lds; };
PsInst pgen_instance(int B, int H, int precision) {
    const int RT = (B + 15) / 16;
    PsInst k{nullptr, 0, 0};
    if (precision == 0) {
        k.threads = PS4_THREADS; k.lds = (size_t)RT * 8 * 256 * 4 + 64;
        k.fn = RT == 1 ? (const void*)pgen7_kernel<1> : RT == 2 ? (const void*)pgen7_kernel<2> : RT == 3 ? (const void*)pgen7_kernel<3> : (const void*)pgen7_kernel<4>;
    } else {
        k.threads = PS_THREADS; k.lds = (size_t)(H / 32) * 1024 + 8 * 64 * 16 * 4;
        k.fn = RT == 1 ? (const void*)pgen_kernel<1, 1> : RT == 2 ? (const void*)pgen_kernel<2, 1> : RT == 3 ? (const void*)pgen_kernel<3, 1> : (const void*)pgen_kernel<4, 1>;
    }
    return k;
}
size_t pdec_lds(int H, int Dm, int precision, int LT) {
    size_t lds = (size_t)((Dm + H) / 32) * (precision ? 1024 : 2048) + 8 * 64 * 16 * 4 + (LT == 1 ? PD_LMAX * 32 * 4 : 0) + (PD_LMAX * LT + 64) * 4 + 2 * 3 * 64 * 16 + 64 * 4 + 16;
    return lds;
}
typedef void (*PdecFn)(PsDec);
// kernel instance by (row tiles, precision, position tiles)
PdecFn pdec_fn(int RT, int precision, int LT) {
#define PDEC_ROW(PR, LTT) { pdec_kernel<1, PR, LTT>, pdec_kernel<2, PR, LTT>, pdec_kernel<3, PR, LTT>, pdec_kernel<4, PR, LTT> }
    static const PdecFn table[2][PD_LT_MAX][4] = {{PDEC_ROW(0, 1), PDEC_ROW(0, 2), PDEC_ROW(0, 3)}, {PDEC_ROW(1, 1), PDEC_ROW(1, 2), PDEC_ROW(1, 3)}};
#undef PDEC_ROW
    return table[precision ? 1 : 0][LT - 1][RT - 1];
}
PsInst pdec_instance(int B, int L, int H, int Dm, int precision) {
    const int RT = (B + 15) / 16, LT = (L + PD_LMAX - 1) / PD_LMAX;
    return PsInst{(const void*)pdec_fn(RT, precision, LT), PS_THREADS, pdec_lds(H, Dm, precision, LT)};
}
}  // namespace

bool pgen_supported(const DecoderArgs& a) {
    if (!(persist_enabled() && a.fast && (a.precision == 0 || a.precision == 1) && a.H == 4 * PS_WGS && a.B >= 1 && a.B <= 64 && a.persist_ws &&
          a.gen_w2p && a.gen_bias_u && a.pre_gen && a.persist_ws_bytes >= mtts_decoder_persist_ws_bytes(a.B, a.L, a.H, a.Dm, a.A)))
        return false;
    const PsInst k = pgen_instance(a.B, a.H, a.precision);
    return ps_kernel_ready(k.fn, k.threads, k.lds);
}

// generator LSTM steps [t0, t1) in one launch (h_gen[t0] / c_gen[t0] are the initial state)
int pgen_launch(const DecoderArgs& a, int t0, int t1, hipStream_t s) {
    MTTS_REQUIRE(pgen_supported(a), "pgen_launch: unsupported shape");
    if (t1 <= t0) return 0;
    PsGen p; memset(&p, 0, sizeof(p));
    p.B = a.B; p.H = a.H; p.t0 = t0; p.t1 = t1;
    p.w_packed = (const float*)a.gen_w2p; p.bias_u = a.gen_bias_u; p.pre = a.pre_gen;
    p.h = a.h_gen; p.c = a.c_gen; p.gates = a.gates_gen;
    if (a.zone) {
        if (a.training) { p.cell.zone = 1; p.hmask = a.gen_hmask; p.cmask = a.gen_cmask; }
        else { p.cell.zone = 2; p.cell.zh = a.p_hidden; p.cell.zc = a.p_cell; }
    } else if (a.training && a.gen_hmask && a.p_hidden > 0.f) {
        p.hmask = a.gen_hmask; p.cell.hscale = 1.f / (1.f - a.p_hidden);
    }
    char* ws = (char*)a.persist_ws;
    p.sync.cnt = (unsigned*)ws; p.sync.err = ps_err_word(a);
    p.xp = (float*)(ws + ps_ws_gen_off());
    std::lock_guard<std::mutex> launch_lk(ps_dev().launch_mu);      // serialize -> launch -> record is one critical section
    MTTS_TRY(ps_serialize(s));
    MTTS_CHECK_HIP(hipMemsetAsync(ws, 0, PS_ERR_OFF, s));           // counters (the error word is sticky until the host reads it)
    const PsInst k = pgen_instance(a.B, a.H, a.precision);
    const int RT = (a.B + 15) / 16;
    if (a.precision == 0) {
#define PGEN7_GO(G) hipLaunchKernelGGL((pgen7_kernel<G>), dim3(PS_WGS), dim3(PS4_THREADS), k.lds, s, p);
        if (RT == 1) PGEN7_GO(1) else if (RT == 2) PGEN7_GO(2) else if (RT == 3) PGEN7_GO(3) else PGEN7_GO(4)
#undef PGEN7_GO
        MTTS_CHECK_LAUNCH("pgen7_kernel");
    } else {
#define PGEN_GO(R) hipLaunchKernelGGL((pgen_kernel<R, 1>), dim3(PS_WGS), dim3(PS_THREADS), k.lds, s, p);
        if (RT == 1) PGEN_GO(1) else if (RT == 2) PGEN_GO(2) else if (RT == 3) PGEN_GO(3) else PGEN_GO(4)
#undef PGEN_GO
        MTTS_CHECK_LAUNCH("pgen_kernel");
    }
    return ps_launched(s);
}

bool pdec_supported(const DecoderArgs& a) {
    if (!(persist_enabled() && a.fast && (a.precision == 0 || a.precision == 1) && a.H == 4 * PS_WGS && a.A == 128 && a.B >= 1 && a.B <= 64 && a.L >= 1 && a.L <= PD_LMAX * PD_LT_MAX &&
          a.persist_ws && a.att_w2p && a.att_bias_u && a.att_w_pre_u && a.pre_att && (a.Dm & 31) == 0 && a.Dm / 32 <= 24 &&
          (a.ksz & 1) == 1 && a.ksz <= 32 && a.persist_ws_bytes >= mtts_decoder_persist_ws_bytes(a.B, a.L, a.H, a.Dm, a.A)))
        return false;
    static const bool off = [] { const char* e = getenv("MTTS_PDEC"); return e && e[0] == '0'; }();
    if (off) return false;
    const int nc4 = a.Dm / 16, ng = PS_THREADS / nc4;
    const int LT = (a.L + PD_LMAX - 1) / PD_LMAX;
    static const int lt_max = [] { const char* e = getenv("MTTS_PDEC_LT"); return e ? atoi(e) : PD_LT_MAX; }();      // MTTS_PDEC_LT=1: long inputs take the per-step schedule (A/B)
    if (LT > lt_max) return false;
    if (!(nc4 >= 1 && ng >= 1 && (a.L + ng - 1) / ng <= PD_NCM * LT)) return false;
    const PsInst k = pdec_instance(a.B, a.L, a.H, a.Dm, a.precision);
    return ps_kernel_ready(k.fn, k.threads, k.lds);
}

// attention LSTM + attention, steps [t0, t1) in one launch; needs U, Mt, pre_att and the packed weights of mtts_decoder_fwd's set-up
int pdec_launch(const DecoderArgs& a, int t0, int t1, hipStream_t s) {
    MTTS_REQUIRE(pdec_supported(a), "pdec_launch: unsupported shape");
    if (t1 <= t0) return 0;
    PsDec p; memset(&p, 0, sizeof(p));
    p.B = a.B; p.L = a.L; p.H = a.H; p.A = a.A; p.Dm = a.Dm; p.ksz = a.ksz; p.t0 = t0; p.t1 = t1;
    p.w_packed = (const float*)a.att_w2p; p.bias_u = a.att_bias_u; p.pre = a.pre_att;
    p.h = a.h_att; p.c = a.c_att; p.gates = a.gates_att;
    if (a.zone) {
        if (a.training) { p.cell.zone = 1; p.hmask = a.att_hmask; p.cmask = a.att_cmask; }
        else { p.cell.zone = 2; p.cell.zh = a.p_hidden; p.cell.zc = a.p_cell; }
    } else if (a.training && a.att_hmask && a.p_hidden > 0.f) {
        p.hmask = a.att_hmask; p.cell.hscale = 1.f / (1.f - a.p_hidden);
    }
    p.w_query = a.w_query; p.memory = a.memory; p.Mt = a.Mt; p.U = a.U; p.att_bias = a.att_bias; p.v = a.w_energy; p.lengths = a.lengths;
    p.ctx = a.ctx; p.cum = a.cum; p.align = a.align; p.q_all = a.q_all;
    char* ws = (char*)a.persist_ws;
    p.sync.cnt = (unsigned*)ws; p.sync.err = ps_err_word(a);
    p.xp = (float*)(ws + ps_ws_att_off(a.H));
    p.eg = (unsigned long long*)(ws + ps_ws_eg_off(a.H, a.Dm));
    std::lock_guard<std::mutex> launch_lk(ps_dev().launch_mu);
    MTTS_TRY(ps_serialize(s));
    MTTS_CHECK_HIP(hipMemsetAsync(ws, 0, PS_ERR_OFF, s));
    const int LT = (a.L + PD_LMAX - 1) / PD_LMAX;
    MTTS_CHECK_HIP(hipMemsetAsync(p.eg, 0, (size_t)64 * 4 * (PD_LMAX * LT) * 8, s));
    static const bool poll_on = [] { const char* e = getenv("MTTS_PDEC_POLL"); return !(e && e[0] == '0'); }();
    p.poll_h = (poll_on && !g_pdec_poll_off) ? 1 : 0;
    static const bool early_on = [] { const char* e = getenv("MTTS_PDEC_EARLY"); return !(e && e[0] == '0'); }();
    p.early_h = (early_on && !g_pdec_early_off) ? 1 : 0;
    if (p.poll_h)       // h rows of the steps this launch produces: sentinel until their owner's store lands
        MTTS_CHECK_HIP(hipMemsetAsync(a.h_att + (size_t)(t0 + 1) * a.B * a.H, 0xff, (size_t)(t1 - t0) * a.B * a.H * sizeof(float), s));
    const size_t lds = pdec_lds(a.H, a.Dm, a.precision, LT);
    const int RT = (a.B + 15) / 16;
    hipLaunchKernelGGL(pdec_fn(RT, a.precision, LT), dim3(PS_WGS), dim3(PS_THREADS), lds, s, p);
    MTTS_CHECK_LAUNCH("pdec_kernel");
    return ps_launched(s);
}

// device error word of the persistent launches on this workspace (0 = ok, 2 = a hand-off timed out); synchronises the stream.
// Reads the word the kernels actually report to: DecoderArgs.persist_err of the last launch on this workspace, or the word inside
// the workspace when no launch named another.
MTTS_API int mtts_decoder_persist_status(const void* persist_ws, void* stream) {
    const unsigned* src = (const unsigned*)((const char*)persist_ws + PS_ERR_OFF);
    {
        PsDevice& D = ps_dev();
        std::lock_guard<std::mutex> lk(D.mu);
        auto it = D.err_of.find(persist_ws);
        if (it != D.err_of.end()) src = it->second.first;
    }
    unsigned v = 0;
    if (hipMemcpyAsync(&v, src, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return (int)v;
}

// ---- host helpers shared with the persistent decoder BACKWARD (pbwd.hip) -----------------------------------------------------------
bool ps_ready_ext(const void* fn, int threads, size_t lds) { return ps_kernel_ready(fn, threads, lds); }
// One persistent launch on `s` with the forward's workspace (a.persist_ws: counters zeroed here, error word = the decode's): ordered
// behind the device's previous persistent launch when that went to another stream; `go` issues the kernel.
int ps_run_launch(const DecoderArgs& a, hipStream_t s, void (*go)(void* ctx, unsigned* cnt, unsigned* err, hipStream_t s), void* ctx) {
    char* ws = (char*)a.persist_ws;
    unsigned* err = ps_err_word(a);
    std::lock_guard<std::mutex> launch_lk(ps_dev().launch_mu);
    MTTS_TRY(ps_serialize(s));
    MTTS_CHECK_HIP(hipMemsetAsync(ws, 0, PS_ERR_OFF, s));
    go(ctx, (unsigned*)ws, err, s);
    MTTS_CHECK_LAUNCH("persistent launch");
    return ps_launched(s);
}

