// Recurrent LSTM step of the decoder as TWO launches that never re-read the batch operand:
//
//   G  lstm_gates_kernel   partial gate pre-activations  P[ks][B][4H] = X[:, kslice] W[:, kslice]^T
//                          grid = (4H / 128 column tiles) x (KS K-slices); 256 workgroups at H = 1024, KS = 8
//   C  lstm_cell_q_kernel  sum of the KS partials + hoisted addend + biases -> LSTM cell (dropout / zoneout on h)
//                          -> h, c, saved gates, and the query projection partials  q_part[ut] = h[:, tile ut] W_q[:, tile ut]^T
//
// Replaces, for the attention LSTM of the teacher-forced ("fast") schedule, one skinny_kernel launch (16 gate columns x all
// rows x the full K per workgroup: every one of 256 workgroups re-read the whole [B, Dm+H] operand from L2, 1.57x the
// algorithmic bytes on the fabric) plus the separate query-projection launch.  Reference: DropoutLSTMCell / ZoneoutLSTMCell
// modules/layers.py:18-47 called at modules/tacotron2.py:185, query projection modules/attention.py:68.
//
// G is W-stationary and K-split: a workgroup streams its [128 columns x K/KS] weight slice HBM -> VGPR exactly once (packed
// per call into wave-tile order by mtts_lstm_pack_weights: 1 KiB contiguous per wave instruction), stages the matching
// [64 rows x K/KS] slice of X through LDS once for all eight waves, and for batches above 64 rows loops over row tiles with
// the weights still in registers.  Workgroups that share a K-slice have the same blockIdx % 8, i.e. sit on one XCD and
// share that slice of X in its L2.
// Arithmetic: fp32 operands are split exactly into three bf16 planes (x = x1 + x2 + x3, see gemm.hip) and every product is
// evaluated as six v_mfma_f32_16x16x32_bf16 terms with fp32 accumulation (precision 0), or operands are rounded to one
// bf16 plane (precision 1: the bf16 path, weights stored as bf16).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int LS_THREADS = 512;
constexpr int LS_COLS = 128;         // gate columns per workgroup = 8 waves x 16
constexpr int LS_MAXKB = 10;         // 32-wide k-blocks per K-slice (weights of a slice stay in registers): K <= 8 x 320 with 8 slices
constexpr int LS_PLANE_B = 64 * 64;  // one k-block of one plane in LDS: 64 rows x 32 bf16
constexpr int LS_MIN_KS = 8;

// K-slices for nkb 32-wide blocks when a slice may hold at most nb_max blocks (0 = LS_MAXKB): at least 8 slices.
// nb_max = 4 selects the two-workgroups-per-CU instantiation (training: the step kernels then share CUs with the helper streams'
// GEMM workgroups instead of waiting for them; 95.7 vs 97.5 ms per train step) at the price of more partial slabs.
static inline int ls_ksplit(int nkb, int nb_max = 0) {
    const int nb = nb_max >= 1 && nb_max <= LS_MAXKB ? nb_max : LS_MAXKB;
    const int need = (nkb + nb - 1) / nb;
    return need > LS_MIN_KS ? need : LS_MIN_KS;
}

__device__ __forceinline__ unsigned bf16_rne(float x) {      // upper 16 bits of the RNE-rounded value
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// exact 3-way split of two floats into three packed bf16 pairs (truncation; residuals are exact in fp32)
__device__ __forceinline__ void ls_split_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
    const float rx = x - __uint_as_float(ux & 0xffff0000u), ry = y - __uint_as_float(uy & 0xffff0000u);
    const unsigned vx = __float_as_uint(rx), vy = __float_as_uint(ry);
    const float sx = rx - __uint_as_float(vx & 0xffff0000u), sy = ry - __uint_as_float(vy & 0xffff0000u);
    p1 = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    p2 = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}

union Frag8 { bf16x8 v; unsigned u[4]; };

// ---------------------------------------------------------------------------------------------------------------------------
// weight packing (once per decoder call)
// fp32 layout:  [col tile j][wave w][k-block kb][half h][lane][4]   lane = 16 q + i holds W[col(j, w, i)][32 kb + 8 q + 4 h + e]
// bf16 layout:  [col tile j][wave w][k-block kb][lane][8]           lane = 16 q + i holds W[col(j, w, i)][32 kb + 8 q + e]
// LSTM column order (unit-major): col(j, w, i) = gate (i & 3) of unit 32 j + 4 w + (i >> 2)  ->  source row (i & 3) H + unit
// ---------------------------------------------------------------------------------------------------------------------------
struct LsPack {
    const float* w0; const float* w1; const float* w2;
    int K0, K1, K2, ld0, ld1, ld2;
    int H, nkb, precision;
    void* dst;
};

__device__ __forceinline__ float ls_pack_src(const LsPack& p, int row, int k) {
    if (k < p.K0) return p.w0[(long)row * p.ld0 + k];
    k -= p.K0;
    if (k < p.K1) return p.w1[(long)row * p.ld1 + k];
    k -= p.K1;
    return p.w2[(long)row * p.ld2 + k];
}

__global__ void lstm_pack_kernel(LsPack p) {
    const long rows = (long)4 * p.H;
    const long total = rows * p.nkb * 32;
    const long n = p.precision ? total / 8 : total / 4;          // one lane-quantum (16 B; planes: one per plane) per thread
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        int lane, h = 0; long rest;
        if (p.precision) { lane = (int)(t & 63); rest = t >> 6; }
        else { lane = (int)(t & 63); h = (int)((t >> 6) & 1); rest = t >> 7; }
        const int kb = (int)(rest % p.nkb); const long jw = rest / p.nkb;
        const int w = (int)(jw & 7), j = (int)(jw >> 3);
        const int i = lane & 15, q = lane >> 4;
        const int row = (i & 3) * p.H + 32 * j + 4 * w + (i >> 2);
        const int k = 32 * kb + 8 * q + 4 * h;
        if (p.precision == 2) {      // exact 3-way split, planes: [col group][k-block][plane][lane] x 16 B
            unsigned o[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ls_split_pair(ls_pack_src(p, row, k + 2 * e), ls_pack_src(p, row, k + 2 * e + 1), o[0][e], o[1][e], o[2][e]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                reinterpret_cast<uint4*>(p.dst)[((jw * p.nkb + kb) * 3 + pl) * 64 + lane] = make_uint4(o[pl][0], o[pl][1], o[pl][2], o[pl][3]);
        } else if (p.precision) {
            unsigned o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = bf16_rne(ls_pack_src(p, row, k + 2 * e)) | (bf16_rne(ls_pack_src(p, row, k + 2 * e + 1)) << 16);
            reinterpret_cast<uint4*>(p.dst)[t] = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            reinterpret_cast<float4*>(p.dst)[t] = make_float4(ls_pack_src(p, row, k), ls_pack_src(p, row, k + 1), ls_pack_src(p, row, k + 2),
                                                               ls_pack_src(p, row, k + 3));
        }
    }
}

// dst[(4 u + g) * K + k] = src[(g H + u) * ld + k]   (rows of a [4H, K] LSTM matrix into unit-major order; K = 1: a bias vector,
// optionally the sum of two)
__global__ void lstm_rows_unit_major_kernel(const float* __restrict__ src, const float* __restrict__ src2, int ld, int H, int K,
                                            float* __restrict__ dst) {
    const long total = (long)4 * H * K;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long r = t / K; const int k = (int)(t - r * K);
        const int u = (int)(r >> 2), g = (int)(r & 3);
        const long s = ((long)g * H + u) * ld + k;
        dst[t] = src[s] + (src2 ? src2[s] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// G: partial gate GEMM
// ---------------------------------------------------------------------------------------------------------------------------
struct LsGates {
    const float* x0; const float* x1; const float* x2;
    int K0, K1, K2, ld0, ld1, ld2;
    const void* wp;
    int nkb, KS, B, N;
    int nbmax;            // k-blocks of the longest slice: LDS holds NPL x nbmax blocks
    float* part;          // [KS][B][N]
};

template <int PREC, int NB>      // NB: k-blocks per K-slice held in registers (the smallest instantiation that fits is launched)
__device__ __forceinline__ void lstm_gates_body(const LsGates& p, const int bid, char* sm) {
    constexpr int NPL = PREC ? 1 : 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = bid % p.KS, j = bid / p.KS;
    const int kb0 = (int)((long)s * p.nkb / p.KS), kb1 = (int)((long)(s + 1) * p.nkb / p.KS);
    const int nb = kb1 - kb0;                           // <= NB (host guarantees); blocks kb >= nb are skipped by wave-uniform guards
    const int i16 = lane & 15, q4 = lane >> 4;

    const int srow = tid >> 3, sk4 = tid & 7;           // staging: 8 threads cover the 32 k of one row
    const int n_row_tiles = (p.B + 63) >> 6;
    // X slice of a row tile: one float4 per thread and k-block (row-major fp32 in global memory; mostly L2 hits).
    // NOTE: every loop over k-blocks is fully unrolled with `if (kb < nb)` guards (no break): register arrays stay registers.
    float4 xv[NB];
#define LS_LOAD_X(ROW0)                                                                                           \
    {                                                                                                             \
        const int rowc = min((ROW0) + srow, p.B - 1);                                                             \
        _Pragma("unroll") for (int kb = 0; kb < NB; ++kb) if (kb < nb) {                                          \
            int kg = 32 * (kb0 + kb);                                                                             \
            const float* xs; int ld;                                                                              \
            if (kg < p.K0) { xs = p.x0; ld = p.ld0; }                                                             \
            else if (kg < p.K0 + p.K1) { xs = p.x1; ld = p.ld1; kg -= p.K0; }                                     \
            else { xs = p.x2; ld = p.ld2; kg -= p.K0 + p.K1; }                                                    \
            xv[kb] = *reinterpret_cast<const float4*>(xs + (long)rowc * ld + kg + 4 * sk4);                       \
        }                                                                                                         \
    }
    LS_LOAD_X(0)                                         // requested BEFORE the weights: it gates the first MFMA

    // ---- this wave's weight slice -> registers (one or two 16-byte loads per lane and k-block, 1 KiB contiguous per
    //      instruction); requested in k order and consumed in k order, so block kb's MFMAs start while later blocks still stream
    float4 wr[NB][PREC ? 1 : 2];
    {
        const float4* src = reinterpret_cast<const float4*>(p.wp) + ((long)(j * 8 + wave) * p.nkb + kb0) * (PREC ? 64 : 128) + lane;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) if (kb < nb) {
            wr[kb][0] = src[(long)kb * (PREC ? 64 : 128)];
            if (!PREC) wr[kb][1] = src[(long)kb * 128 + 64];
        }
    }

    for (int rt = 0; rt < n_row_tiles; ++rt) {
        const int row0 = rt * 64;
        // ---- X slice of this row tile -> bf16 plane(s) in LDS
        {
            if (rt > 0) { LS_LOAD_X(row0) __syncthreads(); }     // the previous row tile's fragments have been consumed
            char* dst = sm + srow * 64 + (((sk4 >> 1) ^ ((srow >> 2) & 3)) * 16) + (sk4 & 1) * 8;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) if (kb < nb) {
                if (PREC) {
                    const unsigned a = bf16_rne(xv[kb].x) | (bf16_rne(xv[kb].y) << 16), b = bf16_rne(xv[kb].z) | (bf16_rne(xv[kb].w) << 16);
                    *reinterpret_cast<uint2*>(dst + kb * LS_PLANE_B) = make_uint2(a, b);
                } else {
                    unsigned a1, a2, a3, b1, b2, b3;
                    ls_split_pair(xv[kb].x, xv[kb].y, a1, a2, a3);
                    ls_split_pair(xv[kb].z, xv[kb].w, b1, b2, b3);
                    *reinterpret_cast<uint2*>(dst + (0 * NB + kb) * LS_PLANE_B) = make_uint2(a1, b1);
                    *reinterpret_cast<uint2*>(dst + (1 * NB + kb) * LS_PLANE_B) = make_uint2(a2, b2);
                    *reinterpret_cast<uint2*>(dst + (2 * NB + kb) * LS_PLANE_B) = make_uint2(a3, b3);
                }
            }
        }
        __syncthreads();

        // ---- 16 columns x up to 64 rows per wave
        const int mt_n = min(4, (p.B - row0 + 15) >> 4);
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int a_off = i16 * 64 + ((q4 ^ ((i16 >> 2) & 3)) * 16);
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) if (kb < nb) {
            Frag8 wb[NPL];
            if (PREC) {
                wb[0].u[0] = __float_as_uint(wr[kb][0].x); wb[0].u[1] = __float_as_uint(wr[kb][0].y);
                wb[0].u[2] = __float_as_uint(wr[kb][0].z); wb[0].u[3] = __float_as_uint(wr[kb][0].w);
            } else {
                ls_split_pair(wr[kb][0].x, wr[kb][0].y, wb[0].u[0], wb[1].u[0], wb[2].u[0]);
                ls_split_pair(wr[kb][0].z, wr[kb][0].w, wb[0].u[1], wb[1].u[1], wb[2].u[1]);
                ls_split_pair(wr[kb][1].x, wr[kb][1].y, wb[0].u[2], wb[1].u[2], wb[2].u[2]);
                ls_split_pair(wr[kb][1].z, wr[kb][1].w, wb[0].u[3], wb[1].u[3], wb[2].u[3]);
            }
            // two row tiles at a time: their accumulators alternate, so no MFMA waits for the one issued just before it
#pragma unroll
            for (int mp = 0; mp < 4; mp += 2) if (mp < mt_n) {
                bf16x8 a[2][NPL];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        a[m][pl] = *reinterpret_cast<const bf16x8*>(sm + (pl * NB + kb) * LS_PLANE_B + min(mp + m, mt_n - 1) * 1024 + a_off);
#define LS_MM(PA, PB)                                                                                            \
    _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                               \
        acc[mp + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][PA], wb[PB].v, acc[mp + m], 0, 0, 0);
                if (PREC) { LS_MM(0, 0) }
                else { LS_MM(2, 0) LS_MM(0, 2) LS_MM(1, 1) LS_MM(1, 0) LS_MM(0, 1) LS_MM(0, 0) }      // small terms first
#undef LS_MM
            }
        }
        // ---- partial slab: D layout col = lane & 15, row = 4 (lane >> 4) + r
        const int col = j * LS_COLS + wave * 16 + i16;
        if (col < p.N) {
            float* out = p.part + ((long)s * p.B) * p.N + col;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 16 * m + 4 * q4 + r;
                    if (row < p.B) out[(long)row * p.N] = acc[m][r];
                }
        }
    }
#undef LS_LOAD_X
}

// NB = 4 is compiled for two workgroups per CU (<= 128 VGPRs, 48 KiB LDS): with short K-slices (KS >= nkb / 4) two of these
// kernels - or one of them and an attention / GEMM workgroup of another stream - share a CU and hide each other's latencies.
template <int PREC, int NB>
__global__ __launch_bounds__(LS_THREADS, (NB <= 4 ? 4 : 2)) void lstm_gates_kernel(LsGates p) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    lstm_gates_body<PREC, NB>(p, blockIdx.x, sm);
}

// ---------------------------------------------------------------------------------------------------------------------------
// C: partial sum + LSTM cell + query partials.  Workgroup = 16 units x 16 rows, thread = one (row, unit).
// ---------------------------------------------------------------------------------------------------------------------------
struct LsCell {
    const float* part; int KS, B, H;
    const float* pre; int ldpre;          // unit-major [B, 4H] or NULL
    const float* bias_u;                  // unit-major [4H] (b_ih + b_hh) or NULL
    const float* h_prev; const float* c_prev;
    float* h_out; float* c_out; float* gates_out;
    const uint8_t* hmask; const uint8_t* cmask;
    float hscale; int zone; float zh, zc;
    const float* wq; int A; float* qpart;  // [H/16][B][A] or NULL
};

constexpr int LC_MAXCT = 4;      // query column tiles per wave: A <= 4 waves x 4 x 16 = 256

__device__ __forceinline__ void lstm_cell_q_body(const LsCell& p, float (&hs)[16][17]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ut = blockIdx.x, rt = blockIdx.y;
    const int r = tid >> 4, uu = tid & 15;
    const int row = 16 * rt + r, u = 16 * ut + uu;
    const bool valid = row < p.B;
    const int rowc = valid ? row : p.B - 1;
    const int N = 4 * p.H;
    const int i16 = lane & 15, q4 = lane >> 4;

    // query-projection operand of this wave (independent of everything else: requested first)
    float4 wq4[LC_MAXCT];
    const int nct = p.wq ? p.A >> 4 : 0;
#pragma unroll
    for (int c = 0; c < LC_MAXCT; ++c) {
        const int ct = wave + 4 * c;
        wq4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ct < nct) wq4[c] = *reinterpret_cast<const float4*>(p.wq + (long)(16 * ct + i16) * p.H + 16 * ut + 4 * q4);
    }

    const long hi = (long)rowc * p.H + u;
    float4 g4 = p.bias_u ? *reinterpret_cast<const float4*>(p.bias_u + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pre4 = p.pre ? *reinterpret_cast<const float4*>(p.pre + (long)rowc * p.ldpre + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float cp = p.c_prev[hi];
    const float hp = p.h_prev ? p.h_prev[hi] : 0.f;
    const int hm = p.hmask ? (int)p.hmask[hi] : 1, cm = p.cmask ? (int)p.cmask[hi] : 1;
    const float* ps = p.part + (long)rowc * N + 4 * u;
    const long slab = (long)p.B * N;
    float4 pv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) pv[k] = *reinterpret_cast<const float4*>(ps + (long)min(k, p.KS - 1) * slab);
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < p.KS) { g4.x += pv[k].x; g4.y += pv[k].y; g4.z += pv[k].z; g4.w += pv[k].w; }
    for (int k = 16; k < p.KS; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(ps + (long)k * slab);
        g4.x += v.x; g4.y += v.y; g4.z += v.z; g4.w += v.w;
    }
    g4.x += pre4.x; g4.y += pre4.y; g4.z += pre4.z; g4.w += pre4.w;

    const float ig = sigmoidf_(g4.x), fg = sigmoidf_(g4.y), gg = tanhf_(g4.z), og = sigmoidf_(g4.w);
    const float cn = fg * cp + ig * gg;
    const float hn = og * tanhf_(cn);
    float ho, co = cn;
    if (p.zone == 1) { ho = hm ? hn : hp; co = cm ? cn : cp; }
    else if (p.zone == 2) { ho = p.zh * hp + (1.f - p.zh) * hn; co = p.zc * cp + (1.f - p.zc) * cn; }
    else ho = p.hmask ? (hm ? hn * p.hscale : 0.f) : hn;
    if (valid) {
        p.h_out[hi] = ho;
        p.c_out[hi] = co;
        if (p.gates_out) {
            float* go = p.gates_out + (long)row * N + u;
            go[0] = ig; go[p.H] = fg; go[2 * p.H] = gg; go[3 * p.H] = og;
        }
    }
    if (!p.qpart) return;
    hs[r][uu] = valid ? ho : 0.f;
    __syncthreads();
    // q_part[ut][rows of this workgroup][A] = h_tile [16 x 16] W_q[:, 16 ut .. +16]^T on v_mfma_f32_16x16x4_f32 (exact fp32)
    const float av[4] = {hs[i16][4 * q4 + 0], hs[i16][4 * q4 + 1], hs[i16][4 * q4 + 2], hs[i16][4 * q4 + 3]};
#pragma unroll
    for (int c = 0; c < LC_MAXCT; ++c) {
        const int ct = wave + 4 * c;
        if (ct >= nct) break;
        const float bv[4] = {wq4[c].x, wq4[c].y, wq4[c].z, wq4[c].w};
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bv[s2], acc, 0, 0, 0);
        float* out = p.qpart + ((long)ut * p.B) * p.A + 16 * ct + i16;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int orow = 16 * rt + 4 * q4 + rr;
            if (orow < p.B) out[(long)orow * p.A] = acc[rr];
        }
    }
}

__global__ __launch_bounds__(256) void lstm_cell_q_kernel(LsCell p) {
    __shared__ float hs[16][17];
    lstm_cell_q_body(p, hs);
}

// ---------------------------------------------------------------------------------------------------------------------------
// V: the whole LSTM step of a batch of ONE or TWO rows in ONE launch (single-utterance synthesis, the reference's synthesize.py mode:
// Decoder.inference, modules/tacotron2.py:229-242 with layers.py:18-47 and attention.py:68).  With one or two rows the step is a
// matrix-VECTOR product: the K-split pair G + C spends it on 64-row MFMA tiles and on the exact 3-way split of every weight fragment
// (14.6 + 4.7 us per LSTM at batch 1 for a 26 MB weight stream, profiles/r04_inference_small_batches.txt).  Here a workgroup owns 16
// LSTM units = four 16-column groups of the packed weight (the fp32 layout of lstm_pack_kernel, read as it is); wave w = (column group
// w & 3, k-block parity w >> 2) streams its half of the group's k-blocks as float4 pairs (lane 16 q + i: column i, k = 32 kb + 8 q .. + 7)
// against the rows' activations (the 16 lanes of a q read the same 32 bytes: one request) on plain fp32 FMA; the four k quads of a
// column are summed across the lanes, the two k-block parities through LDS in a fixed order; then the cell of lstm_cell_q_body for
// (row, unit) and the query partials q_part[unit group][B][A] (the same slabs as the other paths: the attention kernel sums H / 16).
// Inference only (no saved gates, no training dropout): the launcher takes it when gates_out is NULL.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int LV_MAXB = 2;       // rows in registers; with 4 / 8 rows the requests in flight per wave no longer cover the latency (batch 5: 110 vs 77 us per frame)
struct LsGemv {
    const float* x0; const float* x1; const float* x2;
    int K0, K1, K2, ld0, ld1, ld2;
    const float* wp; int nkb;
    LsCell c;             // part / KS unused
};

template <int NB, int UNR>      // NB rows in registers (B <= NB), UNR k-blocks requested together
__global__ __launch_bounds__(LS_THREADS) void lstm_gemv_kernel(LsGemv p) {
    __shared__ float gs[8][16][NB];          // [wave][column of the group][row]
    __shared__ float hs[NB][16];
    __shared__ float qs[4][NB][128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ut = blockIdx.x;
    const int grp = wave & 3, kh = wave >> 2;
    const int i = lane & 15, q = lane >> 4;
    const LsCell& c = p.c;
    const int B = c.B, H = c.H, N = 4 * H;

    // ---- cell / query operands first (threads (row b = tid >> 4, unit uu = tid & 15) and (channel a, unit quad ug))
    const int cb = tid >> 4, cu = tid & 15;
    const bool cell_thread = cb < B;
    const int cbr = min(cb, B - 1), u = 16 * ut + cu;
    const long hi = (long)cbr * H + u;
    const float4 bias4 = *reinterpret_cast<const float4*>((c.bias_u ? c.bias_u : c.c_prev) + (c.bias_u ? 4 * u : 0));
    const float4 pre4 = *reinterpret_cast<const float4*>((c.pre ? c.pre : c.c_prev) + (c.pre ? (long)cbr * c.ldpre + 4 * u : 0));
    const float cp = c.c_prev[hi];
    const float hp = (c.h_prev ? c.h_prev : c.c_prev)[hi];
    const int A = c.qpart ? c.A : 0;
    const int qa = tid & 127, qg = tid >> 7;                       // A <= 128: channel qa, units 4 qg .. + 3 of the group
    const float4 wq4 = *reinterpret_cast<const float4*>((A ? c.wq : c.c_prev) + (A ? (long)min(qa, A - 1) * H + 16 * ut + 4 * qg : 0));

    // ---- gate products of this wave: column group grp, k-blocks kb = kh, kh + 2, ...
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    const long jw = (long)(ut >> 1) * 8 + 4 * (ut & 1) + grp;
    const float4* wb = reinterpret_cast<const float4*>(p.wp) + jw * p.nkb * 128 + lane;      // + (kb * 2 + h) * 64
    const int nk0 = p.K0 >> 5, nk1 = p.K1 >> 5;
    for (int kb0 = kh; kb0 < p.nkb; kb0 += 2 * UNR) {
        float4 w4[UNR][2], x4[UNR][NB][2];
#pragma unroll
        for (int un = 0; un < UNR; ++un) {
            const int kb = min(kb0 + 2 * un, p.nkb - 1);              // wave-uniform; blocks past the end re-read the last one (not added)
            const bool in0 = kb < nk0, in1 = kb < nk0 + nk1;
            const float* xs = in0 ? p.x0 : (in1 ? p.x1 : p.x2);
            const int ld = in0 ? p.ld0 : (in1 ? p.ld1 : p.ld2);
            const int kk = 32 * (in0 ? kb : (in1 ? kb - nk0 : kb - nk0 - nk1)) + 8 * q;
            w4[un][0] = wb[(long)kb * 128]; w4[un][1] = wb[(long)kb * 128 + 64];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float* xr = xs + (long)min(b, B - 1) * ld + kk;
                x4[un][b][0] = *reinterpret_cast<const float4*>(xr); x4[un][b][1] = *reinterpret_cast<const float4*>(xr + 4);
            }
        }
#pragma unroll
        for (int un = 0; un < UNR; ++un) {
            if (kb0 + 2 * un < p.nkb) {                              // wave-uniform
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float a = acc[b];
                    a += w4[un][0].x * x4[un][b][0].x; a += w4[un][0].y * x4[un][b][0].y; a += w4[un][0].z * x4[un][b][0].z; a += w4[un][0].w * x4[un][b][0].w;
                    a += w4[un][1].x * x4[un][b][1].x; a += w4[un][1].y * x4[un][b][1].y; a += w4[un][1].z * x4[un][b][1].z; a += w4[un][1].w * x4[un][b][1].w;
                    acc[b] = a;
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {              // the four k quads of a column: lanes i, i + 16, i + 32, i + 48
        acc[b] += __shfl_xor(acc[b], 16, 64);
        acc[b] += __shfl_xor(acc[b], 32, 64);
        if (q == 0) gs[wave][i][b] = acc[b];
    }
    __syncthreads();

    // ---- cell (lstm_cell_q_body's arithmetic): unit cu of the group = column group cu >> 2, columns 4 (cu & 3) + gate
    if (cell_thread) {
        const int g = cu >> 2, c0 = 4 * (cu & 3);
        float gi = gs[g][c0 + 0][cb] + gs[g + 4][c0 + 0][cb], gf = gs[g][c0 + 1][cb] + gs[g + 4][c0 + 1][cb];
        float gg_ = gs[g][c0 + 2][cb] + gs[g + 4][c0 + 2][cb], go = gs[g][c0 + 3][cb] + gs[g + 4][c0 + 3][cb];
        if (c.bias_u) { gi += bias4.x; gf += bias4.y; gg_ += bias4.z; go += bias4.w; }
        if (c.pre) { gi += pre4.x; gf += pre4.y; gg_ += pre4.z; go += pre4.w; }
        const float ig = sigmoidf_(gi), fg = sigmoidf_(gf), gg = tanhf_(gg_), og = sigmoidf_(go);
        const float cn = fg * cp + ig * gg;
        const float hn = og * tanhf_(cn);
        const float hpv = c.h_prev ? hp : 0.f;
        float ho = hn, co = cn;
        if (c.zone == 2) { ho = c.zh * hpv + (1.f - c.zh) * hn; co = c.zc * cp + (1.f - c.zc) * cn; }
        c.h_out[hi] = ho;
        c.c_out[hi] = co;
        hs[cb][cu] = ho;
    }
    if (!A) return;
    __syncthreads();
    // ---- query partials of the group: q_part[ut][b][a] = sum over its 16 units of h[b][u] W_q[a][u]; four unit quads through LDS
    if (qa < A) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (b < B) qs[qg][b][qa] = hs[b][4 * qg] * wq4.x + hs[b][4 * qg + 1] * wq4.y + hs[b][4 * qg + 2] * wq4.z + hs[b][4 * qg + 3] * wq4.w;
    }
    __syncthreads();
    for (int e = tid; e < B * A; e += LS_THREADS) {
        const int b = e / A, a = e - b * A;
        c.qpart[((long)ut * B + b) * A + a] = ((qs[0][b][a] + qs[1][b][a]) + qs[2][b][a]) + qs[3][b][a];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// F: the whole LSTM step of a LARGE batch (B > 64) in ONE launch - gate GEMM over the full K, cell, query partials.
// With more than 64 rows there is enough work per launch without splitting K, and the K-split form pays for it: at batch 240 the
// partial slabs of the two decoder LSTMs are 150 MB of write + read traffic per step (456 MB measured per decoder step against 122 MB
// algorithmic, profiles/r04_*) and the cell needs a second launch.  Here a workgroup owns 16 LSTM units (64 unit-major gate columns =
// four 16-column groups of the packed weight) x 32 NRT rows; wave w = (column group w & 3, row half w >> 2) walks the whole K:
//   * weight fragments straight from the packed copy (L2: the RG row groups of a unit group are placed on ONE XCD, so every weight
//     block crosses the fabric once per step), activation fragments straight from the row-major [B, K] operands (the four waves of a
//     row half read the same lines: L1 hits), DEPTH k-blocks in flight;
//   * products: precision 0 on v_mfma_f32_16x16x4_f32 - exact fp32 products, no operand split (at this size the 3-way bf16 split is
//     VALU-bound, the fp32 matrix rate is not the limit of the step: 2.6 GFLOP = 16 us for the attention LSTM at batch 240);
//     precision 1: activations rounded to bf16 (RNE) on the fly, bf16-packed weights, one v_mfma_f32_16x16x32_bf16 per fragment pair;
//   * epilogue: the four gates of a unit sit in four neighbouring columns of the wave's own tile -> cell of lstm_cell_q_body through a
//     wave-private LDS patch (no block barrier), h tile -> LDS -> query partials q_part[unit group][B][A] on fp32 MFMA (same slabs
//     as the K-split path: the attention kernel sums H / 16 of them).
// ---------------------------------------------------------------------------------------------------------------------------
struct LsFused {
    const float* x0; const float* x1; const float* x2;
    int K0, K1, K2, ld0, ld1, ld2;
    const void* wp; int nkb;
    LsCell c;             // part / KS unused
};

// Cell / query operands of a wave's (row, unit) pairs and its query-weight fragments.  Every load is unconditional - an absent operand
// reads a valid stand-in address and is replaced when it is used: a load under a condition makes the compiler drain the whole operand
// stream in front of the first product.
constexpr int LF_NQ = 2;                                    // query channel tiles per wave: A <= 8 waves x 2 x 16 = 256
template <int NRT>
struct LfCellOps { float4 pre4[NRT]; float cp[NRT], hp[NRT]; unsigned hm[NRT], cm[NRT]; float4 wq4[LF_NQ]; int nct; };

template <int NRT>
__device__ __forceinline__ void lf_cell_prefetch(const LsCell& c, LfCellOps<NRT>& o, int row0, int ug, int ct, int wave, int lane) {
    const int B = c.B, H = c.H, i16 = lane & 15, q4 = lane >> 4;
    const float* pre_p = c.pre ? c.pre : c.c_prev;  const int pre_ld = c.pre ? c.ldpre : 0;
    const float* hp_p = c.h_prev ? c.h_prev : c.c_prev;
    const uint8_t* hm_p = c.hmask ? c.hmask : reinterpret_cast<const uint8_t*>(c.c_prev);
    const uint8_t* cm_p = c.cmask ? c.cmask : reinterpret_cast<const uint8_t*>(c.c_prev);
#pragma unroll
    for (int n = 0; n < NRT; ++n) {
        const int pi = lane + 64 * n, rl = pi >> 2, uu = pi & 3;
        const int row = min(row0 + rl, B - 1), u = 16 * ug + 4 * ct + uu;
        const long hi = (long)row * H + u;
        o.pre4[n] = *reinterpret_cast<const float4*>(pre_p + (c.pre ? (long)row * pre_ld + 4 * u : 0));
        o.cp[n] = c.c_prev[hi];
        o.hp[n] = hp_p[hi];
        o.hm[n] = hm_p[hi];
        o.cm[n] = cm_p[hi];
    }
    o.nct = c.qpart ? c.A >> 4 : 0;
#pragma unroll
    for (int k = 0; k < LF_NQ; ++k) {
        const int cta = min(wave + 8 * k, max(o.nct - 1, 0));
        o.wq4[k] = *reinterpret_cast<const float4*>((o.nct ? c.wq + (long)(16 * cta + i16) * H + 16 * ug + 4 * q4 : c.c_prev));
    }
}

// Gate pre-activations of the wave's own [16 NRT rows x 16 gate columns] tile (accumulator layout) -> wave-private LDS patch -> cell per
// (row, unit) -> h tile in LDS -> query partials q_part[ug][rows of the workgroup][A] on exact fp32 MFMA (wave <-> channel tiles).
template <int NRT>
__device__ __forceinline__ void lf_cell_epilogue(const LsCell& c, const LfCellOps<NRT>& o, float (*red)[16 * NRT][16], float (*hs)[17],
                                                 const f32x4 (&g)[NRT], int row0, int rg, int ug, int ct, int rh, int wave, int lane) {
    constexpr int RPW = 32 * NRT;
    const int B = c.B, H = c.H, N = 4 * H, i16 = lane & 15, q4 = lane >> 4;
    float (*rw)[16] = red[wave];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[16 * rt + 4 * q4 + r][i16] = g[rt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int n = 0; n < NRT; ++n) {
        const int pi = lane + 64 * n, rl = pi >> 2, uu = pi & 3;
        const int row = row0 + rl, u = 16 * ug + 4 * ct + uu;
        const bool valid = row < B;
        float4 g4 = *reinterpret_cast<const float4*>(&rw[rl][4 * uu]);
        if (c.bias_u) { const float4 b4 = *reinterpret_cast<const float4*>(c.bias_u + 4 * u); g4.x += b4.x; g4.y += b4.y; g4.z += b4.z; g4.w += b4.w; }
        if (c.pre) { g4.x += o.pre4[n].x; g4.y += o.pre4[n].y; g4.z += o.pre4[n].z; g4.w += o.pre4[n].w; }
        const float ig = sigmoidf_(g4.x), fg = sigmoidf_(g4.y), gg = tanhf_(g4.z), og = sigmoidf_(g4.w);
        const float cn = fg * o.cp[n] + ig * gg;
        const float hn = og * tanhf_(cn);
        const float hpv = c.h_prev ? o.hp[n] : 0.f;
        const bool hk = !c.hmask || o.hm[n] != 0u, ck = !c.cmask || o.cm[n] != 0u;
        float ho, co = cn;
        if (c.zone == 1) { ho = hk ? hn : hpv; co = ck ? cn : o.cp[n]; }
        else if (c.zone == 2) { ho = c.zh * hpv + (1.f - c.zh) * hn; co = c.zc * o.cp[n] + (1.f - c.zc) * cn; }
        else ho = c.hmask ? (hk ? hn * c.hscale : 0.f) : hn;
        if (valid) {
            const long hi = (long)row * H + u;
            c.h_out[hi] = ho;
            c.c_out[hi] = co;
            if (c.gates_out) {
                float* go = c.gates_out + (long)row * N + u;
                go[0] = ig; go[H] = fg; go[2 * H] = gg; go[3 * H] = og;
            }
        }
        hs[rh * (16 * NRT) + rl][4 * ct + uu] = valid ? ho : 0.f;
    }
    if (!c.qpart) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LF_NQ; ++k) {
        const int cta = wave + 8 * k;
        if (cta >= o.nct) break;
        const float bv[4] = {o.wq4[k].x, o.wq4[k].y, o.wq4[k].z, o.wq4[k].w};
#pragma unroll
        for (int rtq = 0; rtq < RPW / 16; ++rtq) {
            const float av[4] = {hs[16 * rtq + i16][4 * q4 + 0], hs[16 * rtq + i16][4 * q4 + 1], hs[16 * rtq + i16][4 * q4 + 2], hs[16 * rtq + i16][4 * q4 + 3]};
            f32x4 qa = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) qa = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bv[s2], qa, 0, 0, 0);
            float* out = c.qpart + ((long)ug * B) * c.A + 16 * cta + i16;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int orow = rg * RPW + 16 * rtq + 4 * q4 + rr;
                if (orow < B) out[(long)orow * c.A] = qa[rr];
            }
        }
    }
}

template <int PREC, int NRT, int DEPTH>
__global__ __launch_bounds__(LS_THREADS, 4) void lstm_fused_kernel(LsFused p) {
    static_assert(DEPTH == 4 || DEPTH == 2, "the LDS double buffer follows the parity of the slot index");
    constexpr int RPW = 32 * NRT;                           // rows per workgroup
    // PREC 0: fp32 operands, exact products on v_mfma_f32_16x16x4_f32.  PREC 1: bf16 operands (the bf16 path).  PREC 2: fp32 operands
    // as three exact bf16 planes - the weights pre-split ONCE per decoder call by mtts_lstm_pack_weights (precision 2), the activations
    // split by the staging threads on their way into LDS - and six v_mfma_f32_16x16x32_bf16 terms per fragment pair (the products of
    // csrc/gemm.hip and of the K-split step kernel: fp32-accurate): half the matrix-pipe time of the fp32 MFMA form.
    constexpr int NPLX = PREC == 2 ? 3 : 1;                 // activation planes in LDS
    constexpr int NWF = PREC == 2 ? 3 : (PREC ? 1 : 2);     // 16-byte weight quanta per lane and k-block
    constexpr int XLD = PREC ? 20 : 36;                     // LDS row of a staged k-block in 4-byte words: 32 bf16 / 32 floats + pad
    __shared__ __attribute__((aligned(16))) float red[8][16 * NRT][16];
    __shared__ __attribute__((aligned(16))) float xs[2][NPLX][RPW][XLD];
    __shared__ float hs[RPW][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q4 = lane >> 4;
    const int B = p.c.B, H = p.c.H, N = 4 * H, nug = H >> 4, RG = (int)gridDim.x / nug;
    int ug, rg;
    if ((nug & 7) == 0) { const int id = blockIdx.x, slot = id >> 3; ug = (id & 7) * (nug >> 3) + slot / RG; rg = slot % RG; }      // row groups of a unit group share an XCD
    else { ug = (int)blockIdx.x / RG; rg = (int)blockIdx.x % RG; }
    const int ct = wave & 3, rh = wave >> 2;
    const int row0 = rg * RPW + rh * (16 * NRT);
    const int cgrp = ug * 4 + ct;

    // ---- operand streams.  Activations: the workgroup's [RPW rows x 32 k] block is fetched ONCE, 8 lanes per row (whole 128-byte
    // lines; a wave reading MFMA fragments straight from the row-major operand gathers 16 rows per instruction, four waves repeat
    // every line, and the h operand's 4 KiB row stride lands the 16 rows on one L2 channel: 44 us per launch for the loads alone at
    // batch 240, scripts/mb/mb_lstm_fused.hip) and staged through a double-buffered LDS block; weights: fragments from the packed copy.
    const int srow = (tid >> 3) & (RPW - 1), sch = tid & 7; // staging role: row of the block, 16-byte chunk (32-row workgroups: both
                                                            // halves of the workgroup fetch and store the same values - no condition)
    const int grow = min(rg * RPW + srow, B - 1);
    float4 xg[DEPTH];
    float4 wa[DEPTH][NWF];
    auto issue = [&](int kb, int slot) {
        int kg = 32 * kb;
        const float* xsrc; int ld;
        if (kg < p.K0) { xsrc = p.x0; ld = p.ld0; }
        else if (kg < p.K0 + p.K1) { xsrc = p.x1; ld = p.ld1; kg -= p.K0; }
        else { xsrc = p.x2; ld = p.ld2; kg -= p.K0 + p.K1; }
        xg[slot] = *reinterpret_cast<const float4*>(xsrc + (long)grow * ld + kg + 4 * sch);
        const float4* ws = reinterpret_cast<const float4*>(p.wp) + ((long)cgrp * p.nkb + kb) * (64 * NWF) + lane;
#pragma unroll
        for (int f = 0; f < NWF; ++f) wa[slot][f] = ws[64 * f];
    };
    auto stage = [&](int slot, int buf) {                   // this thread's 16 bytes of the block in `slot` -> LDS
        const float4 v = xg[slot];
        if (PREC == 2) {
            unsigned a1, a2, a3, b1, b2, b3;
            ls_split_pair(v.x, v.y, a1, a2, a3);
            ls_split_pair(v.z, v.w, b1, b2, b3);
            *reinterpret_cast<uint2*>(&xs[buf][0][srow][2 * sch]) = make_uint2(a1, b1);
            *reinterpret_cast<uint2*>(&xs[buf][NPLX > 1 ? 1 : 0][srow][2 * sch]) = make_uint2(a2, b2);
            *reinterpret_cast<uint2*>(&xs[buf][NPLX > 2 ? 2 : 0][srow][2 * sch]) = make_uint2(a3, b3);
        } else if (PREC) *reinterpret_cast<uint2*>(&xs[buf][0][srow][2 * sch]) = make_uint2(bf16_rne(v.x) | (bf16_rne(v.y) << 16), bf16_rne(v.z) | (bf16_rne(v.w) << 16));
        else *reinterpret_cast<float4*>(&xs[buf][0][srow][4 * sch]) = make_float4(v.x, v.y, v.z, v.w);
    };
    // Every load of the stream is UNCONDITIONAL (blocks past the end re-read the last block and are never multiplied): with loads or
    // their uses under a condition the compiler's wait-count bookkeeping gives up and waits for vmcnt(0) before every block.
    const int last = p.nkb - 1;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(min(d, last), d);
    __builtin_amdgcn_sched_barrier(0);

    // ---- cell / query operands (independent of the products: requested behind the first blocks, landed long before the epilogue)
    const LsCell& c = p.c;
    LfCellOps<NRT> co;
    lf_cell_prefetch<NRT>(c, co, row0, ug, ct, wave, lane);
    __builtin_amdgcn_sched_barrier(0);

    // ---- products over the whole K
    f32x4 acc[NRT][2];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) { acc[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    auto use = [&](int d, int buf) {                        // block in weight slot d x the staged activation block in xs[buf]
        if (PREC == 2) {
            Frag8 wb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const float4 w4 = wa[d][pl < NWF ? pl : 0];
                wb[pl].u[0] = __float_as_uint(w4.x); wb[pl].u[1] = __float_as_uint(w4.y); wb[pl].u[2] = __float_as_uint(w4.z); wb[pl].u[3] = __float_as_uint(w4.w);
            }
            Frag8 a[NRT][3];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const uint4 av = *reinterpret_cast<const uint4*>(&xs[buf][pl < NPLX ? pl : 0][rh * (16 * NRT) + 16 * rt + i16][4 * q4]);
                    a[rt][pl].u[0] = av.x; a[rt][pl].u[1] = av.y; a[rt][pl].u[2] = av.z; a[rt][pl].u[3] = av.w;
                }
            // six terms, small ones first (the order of gemm.hip / lstm_gates_body); the two accumulators of a row tile alternate
#define LF_MM(PA, PB, S) _Pragma("unroll") for (int rt = 0; rt < NRT; ++rt) acc[rt][S] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][PA].v, wb[PB].v, acc[rt][S], 0, 0, 0);
            LF_MM(2, 0, 0) LF_MM(0, 2, 1) LF_MM(1, 1, 0) LF_MM(1, 0, 1) LF_MM(0, 1, 0) LF_MM(0, 0, 1)
#undef LF_MM
        } else if (PREC) {
            Frag8 wb;
            wb.u[0] = __float_as_uint(wa[d][0].x); wb.u[1] = __float_as_uint(wa[d][0].y);
            wb.u[2] = __float_as_uint(wa[d][0].z); wb.u[3] = __float_as_uint(wa[d][0].w);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const uint4 av = *reinterpret_cast<const uint4*>(&xs[buf][0][rh * (16 * NRT) + 16 * rt + i16][4 * q4]);      // 8 bf16: k = 8 q4 .. + 7
                Frag8 a; a.u[0] = av.x; a.u[1] = av.y; a.u[2] = av.z; a.u[3] = av.w;
                acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, wb.v, acc[rt][0], 0, 0, 0);
            }
        } else {
            // k of (hf, e) = 32 kb + 8 q4 + 4 hf + e on both operands; two accumulators per row tile alternate
            float4 xv[NRT][2];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) xv[rt][hf] = *reinterpret_cast<const float4*>(&xs[buf][0][rh * (16 * NRT) + 16 * rt + i16][8 * q4 + 4 * hf]);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float wv[4] = {wa[d][hf < NWF ? hf : 0].x, wa[d][hf < NWF ? hf : 0].y, wa[d][hf < NWF ? hf : 0].z, wa[d][hf < NWF ? hf : 0].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) {
                        const float xe = e == 0 ? xv[rt][hf].x : e == 1 ? xv[rt][hf].y : e == 2 ? xv[rt][hf].z : xv[rt][hf].w;
                        acc[rt][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xe, wv[e], acc[rt][e & 1], 0, 0, 0);
                    }
            }
        }
    };
    // block kb lives in global slot kb % DEPTH and in LDS buffer kb & 1 (DEPTH even: both are static in the unrolled body).
    // Per block ONE barrier: it publishes the next block's LDS copy and retires this block's reads before that buffer is rewritten.
    stage(0, 0);
    __syncthreads();
    const int nfull = (p.nkb / DEPTH) * DEPTH;
    for (int kb0 = 0; kb0 < nfull; kb0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            stage((d + 1) % DEPTH, (d + 1) & 1);            // block kb0 + d + 1 (requested DEPTH - 1 blocks ago)
            use(d, d & 1);
            issue(min(kb0 + d + DEPTH, last), d);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < p.nkb - nfull) {                            // slots 0 .. hold the blocks nfull .. (uniform condition)
            if (d + 1 < DEPTH) stage(d + 1, (d + 1) & 1);
            use(d, d & 1);
            __syncthreads();
        }

    // ---- cell of the wave's own tile, h tile, query partials
    f32x4 gsum[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) gsum[rt][r] = acc[rt][0][r] + acc[rt][1][r];
    lf_cell_epilogue<NRT>(c, co, red, hs, gsum, row0, rg, ug, ct, rh, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------
// F2: the fused step with bf16-plane products (PREC 1: the bf16 path; PREC 2: fp32 as three exact planes, six terms) re-tiled around the
// matrix pipe.  Round-4 knock-outs of F (scripts/mb/mb_lstm_fused.hip, batch 240, K = 1312): 31 us per launch against 6 us of MFMA -
// every 32-k block paid its own barrier with only 2 x 12 MFMAs per SIMD behind it, four waves read every activation fragment from LDS
// (96 KB of fragment reads per 64 k and workgroup: the LDS port, not the matrix pipe, set the pace), the two waves of a column group
// fetched the SAME weight fragments, and the fragment reads of a block started only after its staging stores.  Here
//   * one barrier per 128 k: wave w = (column-group pair w & 1, k quarter w >> 1) multiplies ITS 32-k quarter of the block for ALL
//     16-row tiles of the workgroup and TWO column groups (2 NRT tiles x 2 groups x 6 terms = 48 MFMAs per wave and barrier): every
//     weight fragment is fetched by exactly one wave, every activation fragment is read from LDS by two;
//   * the activation fragments of row tile r + 1 are read while the terms of tile r issue (register double buffer), the next block is
//     staged (split + LDS stores) behind the first tiles' terms, global loads run two blocks (256 k) ahead;
//   * the four k quarters meet once, after the loop, through LDS; then wave w = (column group w & 3, row half w >> 2) runs the cell /
//     query epilogue of F on its own tile.
// One workgroup per CU (dynamic LDS up to 102 KiB, <= 256 VGPRs): the two LSTM chains of a decoder step take turns on the chip - side
// by side they competed for the same LDS port and matrix pipe, their sum is what a step costs either way.
// ---------------------------------------------------------------------------------------------------------------------------
template <int PREC, int NRT> struct Lf2Geom {
    static constexpr int RPW = 32 * NRT, NPLX = PREC == 2 ? 3 : 1, XLD = 68;       // LDS row of a staged 128-k block in 4-byte words: 128 bf16 + pad
    static constexpr int XS_BYTES = 2 * NPLX * RPW * XLD * 4;                      // double-buffered activation planes
    static constexpr int PART_BYTES = 4 * 4 * RPW * 16 * 4;                        // [k quarter][column group][row][16] partial tiles
    static constexpr int RED_BYTES = 8 * 16 * NRT * 16 * 4, HS_BYTES = RPW * 17 * 4;
    static constexpr int SM_BYTES = XS_BYTES > PART_BYTES ? (XS_BYTES > RED_BYTES + HS_BYTES ? XS_BYTES : RED_BYTES + HS_BYTES)
                                                          : (PART_BYTES > RED_BYTES + HS_BYTES ? PART_BYTES : RED_BYTES + HS_BYTES);
};

template <int PREC, int NRT>
__global__ __launch_bounds__(LS_THREADS, 2) void lstm_fused2_kernel(LsFused p) {
    static_assert(PREC == 1 || PREC == 2, "plane products only (fp32 MFMA: lstm_fused_kernel<0, ..>)");
    using G = Lf2Geom<PREC, NRT>;
    // 128-k blocks in flight (global -> registers): weights DW, activations DX.  The activations are consumed one block EARLIER than the
    // weights (block it + 1 is staged during iteration it); a third slot pays only for the 64-row fp32 form (27.8 vs 28.6 us at batch
    // 240; 32-row workgroups and the bf16 form are 0.5-1.8 us FASTER with two: scripts/mb/mb_lstm_fused.hip).
    constexpr int DW = 2, DX = (PREC == 2 && NRT == 2) ? 3 : 2, UNR = DX == 3 ? 6 : 2;
    static_assert(UNR % DW == 0 && UNR % DX == 0 && UNR % 2 == 0, "ring slots and the LDS buffer are static in the unrolled body");
    constexpr int RPW = G::RPW, RT = 2 * NRT;               // rows per workgroup; 16-row tiles (every wave multiplies all of them)
    constexpr int NPLX = G::NPLX, XLD = G::XLD;
    constexpr int NXG = 2 * NRT;                            // 16-byte quanta a thread stages per block: RPW x 128 floats / 512 threads
    // LDS: the activation planes during the loop; afterwards the partial tiles of the k quarters, then the epilogue's patches / h tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lf2_sm[];
    unsigned (*xs)[NPLX][RPW][XLD] = reinterpret_cast<unsigned (*)[NPLX][RPW][XLD]>(lf2_sm);
    float (*part)[4][RPW][16] = reinterpret_cast<float (*)[4][RPW][16]>(lf2_sm);
    float (*red)[16 * NRT][16] = reinterpret_cast<float (*)[16 * NRT][16]>(lf2_sm);
    float (*hs)[17] = reinterpret_cast<float (*)[17]>(lf2_sm + G::RED_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q4 = lane >> 4;
    const int B = p.c.B, H = p.c.H, nug = H >> 4, RG = (int)gridDim.x / nug;
    int ug, rg;
    if ((nug & 7) == 0) { const int id = blockIdx.x, slot = id >> 3; ug = (id & 7) * (nug >> 3) + slot / RG; rg = slot % RG; }      // row groups of a unit group share an XCD
    else { ug = (int)blockIdx.x / RG; rg = (int)blockIdx.x % RG; }
    const int ctp = wave & 1, kq = wave >> 1;               // loop role: column groups 2 ctp, 2 ctp + 1; k quarter
    const int ct = wave & 3, rh = wave >> 2;                // epilogue role: column group, row half
    const int row0 = rg * RPW + rh * (16 * NRT);
    const int last = p.nkb - 1, nit = (p.nkb + 3) >> 2;     // 32-k blocks of the packed weight; 128-k iterations

    // staging role: row of the block, 16-byte chunk of a 32-k quarter; 64-row workgroups: all four quarters, 32-row ones: 2 (tid >> 8) + j
    const int srow = (tid >> 3) & (RPW - 1), sch = tid & 7;
    const int grow = min(rg * RPW + srow, B - 1);
    const int sq0 = NRT == 2 ? 0 : 2 * (tid >> 8);
    float4 xg[DX][NXG];
    float4 wa[DW][2][NPLX];
    // every load unconditional (blocks past the end re-read the last one)
    auto issue_x = [&](int it, int slot) {
#pragma unroll
        for (int j = 0; j < NXG; ++j) {
            int kg = 32 * min(4 * it + sq0 + j, last);
            const float* xsrc; int ld;
            if (kg < p.K0) { xsrc = p.x0; ld = p.ld0; }
            else if (kg < p.K0 + p.K1) { xsrc = p.x1; ld = p.ld1; kg -= p.K0; }
            else { xsrc = p.x2; ld = p.ld2; kg -= p.K0 + p.K1; }
            xg[slot][j] = *reinterpret_cast<const float4*>(xsrc + (long)grow * ld + kg + 4 * sch);
        }
    };
    auto issue_w = [&](int it, int slot) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const float4* ws = reinterpret_cast<const float4*>(p.wp) + ((long)(ug * 4 + 2 * ctp + cc) * p.nkb + min(4 * it + kq, last)) * (64 * NPLX) + lane;
#pragma unroll
            for (int f = 0; f < NPLX; ++f) wa[slot][cc][f] = ws[64 * f];
        }
    };
    // block `it` (register slot `slot`) -> LDS buffer `buf`; a 32-k quarter past the end of K is staged as zeros (its weight fragment
    // is a finite re-read)
    auto stage = [&](int it, int slot, int buf) {
#pragma unroll
        for (int j = 0; j < NXG; ++j) {
            const int quarter = sq0 + j;
            const bool ok = 4 * it + quarter <= last;
            float4 v = xg[slot][j];
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            unsigned* dst = &xs[buf][0][srow][16 * quarter + 2 * sch];
            if (PREC == 2) {
                unsigned a1, a2, a3, b1, b2, b3;
                ls_split_pair(v.x, v.y, a1, a2, a3);
                ls_split_pair(v.z, v.w, b1, b2, b3);
                *reinterpret_cast<uint2*>(dst) = make_uint2(a1, b1);
                *reinterpret_cast<uint2*>(dst + (NPLX > 1 ? 1 : 0) * (RPW * XLD)) = make_uint2(a2, b2);
                *reinterpret_cast<uint2*>(dst + (NPLX > 2 ? 2 : 0) * (RPW * XLD)) = make_uint2(a3, b3);
            } else {
                *reinterpret_cast<uint2*>(dst) = make_uint2(bf16_rne(v.x) | (bf16_rne(v.y) << 16), bf16_rne(v.z) | (bf16_rne(v.w) << 16));
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DX; ++d) issue_x(d, d);
#pragma unroll
    for (int d = 0; d < DW; ++d) issue_w(d, d);
    __builtin_amdgcn_sched_barrier(0);
    const LsCell& c = p.c;
    LfCellOps<NRT> co;
    lf_cell_prefetch<NRT>(c, co, row0, ug, ct, wave, lane);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { acc[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    auto load_a = [&](int buf, int rt, Frag8 (&a)[NPLX]) {
#pragma unroll
        for (int pl = 0; pl < NPLX; ++pl) {
            const uint4 av = *reinterpret_cast<const uint4*>(&xs[buf][pl][16 * rt + i16][16 * kq + 4 * q4]);      // 8 bf16: k = 32 kq + 8 q4 ..
            a[pl].u[0] = av.x; a[pl].u[1] = av.y; a[pl].u[2] = av.z; a[pl].u[3] = av.w;
        }
    };
    // one iteration: the wave's 32-k quarter of block `it` (weights in slot u % DW, activations in LDS buffer u & 1) times all row tiles
    // for its two column groups; block it + 1 is staged (split: VALU, + LDS stores) behind the terms of the first row tiles.
    // (Measured and dropped: the two waves of a SIMD staging at opposite ends of the iteration, so that one's VALU work meets the
    // other's matrix work: 29.1 vs 27.9 us at batch 240 - the loop is not issue bound)
    auto step = [&](int it, int u, bool stage_next) {
        const int d = u % DW;
        Frag8 wb[2][NPLX];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int pl = 0; pl < NPLX; ++pl) {
                const float4 w4 = wa[d][cc][pl];
                wb[cc][pl].u[0] = __float_as_uint(w4.x); wb[cc][pl].u[1] = __float_as_uint(w4.y); wb[cc][pl].u[2] = __float_as_uint(w4.z); wb[cc][pl].u[3] = __float_as_uint(w4.w);
            }
        Frag8 a[2][NPLX];
        load_a(u & 1, 0, a[0]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (rt + 1 < RT) load_a(u & 1, rt + 1, a[(rt + 1) & 1]);
            Frag8 (&ar)[NPLX] = a[rt & 1];
            if (PREC == 2) {       // six terms, small ones first (the order of gemm.hip / lstm_gates_body); the two column groups alternate
#define LF2_MM(PA, PB) acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[PA].v, wb[0][PB].v, acc[rt][0], 0, 0, 0); \
                       acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[PA].v, wb[1][PB].v, acc[rt][1], 0, 0, 0);
                LF2_MM(2, 0) LF2_MM(0, 2) LF2_MM(1, 1) LF2_MM(1, 0) LF2_MM(0, 1) LF2_MM(0, 0)
#undef LF2_MM
            } else {
                acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[0].v, wb[0][0].v, acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[0].v, wb[1][0].v, acc[rt][1], 0, 0, 0);
            }
            if (rt == (RT > 2 ? 1 : 0) && stage_next) stage(it + 1, (u + 1) % DX, (u + 1) & 1);
        }
    };
    // block it lives in weight slot it % DW, activation slot it % DX and LDS buffer it & 1 (all static in the body unrolled UNR times).
    // ONE barrier per block: it publishes the next block's LDS copy and retires this block's fragment reads before that buffer is
    // rewritten.
    stage(0, 0, 0);
    issue_x(DX, 0);                                         // block 0's slot is free again
    __syncthreads();
    const int nfull = (nit / UNR) * UNR;
    for (int it0 = 0; it0 < nfull; it0 += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            step(it0 + u, u, true);
            issue_w(it0 + u + DW, u % DW);                  // the slots this iteration has consumed: its weights, and the activations of
            issue_x(it0 + u + 1 + DX, (u + 1) % DX);        // block it + 1 it has just staged
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
        if (u < nit - nfull) {                              // (uniform condition; requests past the end re-read the last block)
            step(nfull + u, u, true);
            issue_w(nfull + u + DW, u % DW);
            issue_x(nfull + u + 1 + DX, (u + 1) % DX);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }

    // ---- the four k quarters meet (the loop's last barrier has retired every read of the activation planes: their space now holds
    //      the partial tiles): part[kq][column group][row][16]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[kq][2 * ctp + cc][16 * rt + 4 * q4 + r][i16] = acc[rt][cc][r];
    __syncthreads();
    f32x4 gsum[NRT];
#pragma unroll
    for (int n = 0; n < NRT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * NRT * rh + 16 * n + 4 * q4 + r;
            gsum[n][r] = (part[0][ct][row][i16] + part[1][ct][row][i16]) + (part[2][ct][row][i16] + part[3][ct][row][i16]);
        }
    __syncthreads();                                        // the space is rewritten by the epilogue's patches
    lf_cell_epilogue<NRT>(c, co, red, hs, gsum, row0, rg, ug, ct, rh, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two-layer prenet of ONE free-running step in one launch (reference Prenet.forward modules/tacotron2.py:37-46 called at :181):
//   y1 = dropout(relu(x W1^T + b1)),  y2 = dropout(relu(y1 W2^T + b2));  dropout always on (keep flags are inputs).
// Workgroup = 16 batch rows; wave w owns output columns {16 w .. } of both layers; y1 stays in LDS.  Exact fp32 MFMA.
// ---------------------------------------------------------------------------------------------------------------------------
struct Prenet2 {
    const float* x; int ldx; int Kin;           // [B, Kin] (Kin % 4 == 0)
    const float* w1; const float* b1; const float* w2; const float* b2;      // w1 / w2 in MFMA tile order (mtts_pack_weight)
    const uint8_t* m1; const uint8_t* m2; float scale;
    float* y1; float* y2;                        // [B, P] each
    int B, P;
};
constexpr int PN_MAXCT = 2;      // layer-1 column tiles per wave: P <= 8 waves x 2 x 16 = 256
constexpr int PN_MAXK1 = 6;      // 16-wide k-chunks of layer 1: Kin <= 96
constexpr int PN_NCG = 4;        // column groups of layer 2 (grid.y): more workgroups for a 22-MFLOP problem
constexpr int PN_MAXK2 = 8;      // layer-2 k-chunks per wave: the 8 waves are 4 column tiles x 2 K-halves, P <= 256

__global__ __launch_bounds__(512) void prenet2_kernel(Prenet2 p) {
    extern __shared__ __attribute__((aligned(16))) float psm[];      // y1 tile [16][P + 4], then the K-half partial sums [4][16][17]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // known wave-uniform: tile bases stay on the scalar unit
    const int i16 = lane & 15, q4 = lane >> 4;
    const int row0 = blockIdx.x * 16, cg = blockIdx.y;
    const int P = p.P, ldh = P + 4, nct = P >> 4, nk2 = P >> 4;
    const int arow = min(row0 + i16, p.B - 1);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nk1 = (p.Kin + 15) >> 4;
    // layer 2 geometry: this workgroup owns column tiles [cg * tpg, (cg + 1) * tpg), tpg <= 4; wave = (tile w & 3, K-half w >> 2)
    const int tpg = (nct + PN_NCG - 1) / PN_NCG;
    const int t2 = cg * tpg + (wave & 3), kh = wave >> 2;
    const bool t2_ok = (wave & 3) < tpg && t2 < nct;
    const int t2c = min(t2, nct - 1);
    const int kc_lo = kh * ((nk2 + 1) >> 1), kc_hi = kh ? nk2 : ((nk2 + 1) >> 1);
    // ---- EVERY global operand of both layers is requested here, before any arithmetic: one memory round trip (weights come
    //      from the Infinity Cache at best: eight other kernels ran since the previous step), then ~2.5 us of MFMA.
    //      Weights are in MFMA tile order ([column tile][k chunk][lane][4], mtts_pack_weight): one coalesced 1 KiB read per wave
    //      instruction - row-major rows would be 64 separate 16-byte requests each and leave the kernel texture-address bound.
    // Every load is UNCONDITIONAL on a clamped address and its value is not touched before the arithmetic (round 4): written as
    // `ok ? *p : 0` the compiler turned the x fragments into four flat dword loads each through a select against a zeroed scratch
    // slot, the weight fragments into dword pieces under exec branches, and put `s_waitcnt vmcnt(0)` behind every keep-flag byte
    // (five serial round trips in front of the first product).  K tails are zeroed on the x side when the fragment is used;
    // column tiles past the end compute on tile nct - 1 and are never stored.
    float4 a4[PN_MAXK1], b1f[PN_MAXCT][PN_MAXK1], b2f[PN_MAXK2];
    bool aok[PN_MAXK1];
    float bias1[PN_MAXCT];
    int keep1[PN_MAXCT][4], keep2[4];
#pragma unroll
    for (int kc = 0; kc < PN_MAXK1; ++kc) {
        const int k = 16 * kc + 4 * q4;
        aok[kc] = k < p.Kin;
        a4[kc] = *reinterpret_cast<const float4*>(p.x + (long)arow * p.ldx + (aok[kc] ? k : 0));
#pragma unroll
        for (int c = 0; c < PN_MAXCT; ++c)
            b1f[c][kc] = *reinterpret_cast<const float4*>(p.w1 + (((long)min(wave + 8 * c, nct - 1) * nk1 + min(kc, nk1 - 1)) * 64 + lane) * 4);
    }
#pragma unroll
    for (int kc = 0; kc < PN_MAXK2; ++kc)
        b2f[kc] = *reinterpret_cast<const float4*>(p.w2 + (((long)t2c * nk2 + min(kc_lo + kc, nk2 - 1)) * 64 + lane) * 4);
#pragma unroll
    for (int c = 0; c < PN_MAXCT; ++c) bias1[c] = p.b1[16 * min(wave + 8 * c, nct - 1) + i16];
    const float bias2 = p.b2[16 * t2c + i16];            // layer-2 epilogue operands travel with the rest
    // keep flags: loaded unconditionally as well (without masks the loads re-read byte 0 of the weights and are ignored) - a
    // branch around them made the compiler wait for everything in flight at the join
    const uint8_t* m1p = p.m1 ? p.m1 : reinterpret_cast<const uint8_t*>(p.w1);
    const uint8_t* m2p = p.m2 ? p.m2 : reinterpret_cast<const uint8_t*>(p.w2);
#pragma unroll
    for (int c = 0; c < PN_MAXCT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            keep1[c][r] = (int)m1p[p.m1 ? (long)min(row0 + 4 * q4 + r, p.B - 1) * P + 16 * min(wave + 8 * c, nct - 1) + i16 : 0];
#pragma unroll
    for (int r = 0; r < 4; ++r) keep2[r] = (int)m2p[p.m2 ? (long)min(row0 + 4 * q4 + r, p.B - 1) * P + 16 * t2c + i16 : 0];
    // ---- layer 1 (every workgroup of a row tile computes all of it: 0.65 MFLOP)
    f32x4 acc[PN_MAXCT];
#pragma unroll
    for (int c = 0; c < PN_MAXCT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < PN_MAXK1; ++kc) {
        const float4 az = aok[kc] ? a4[kc] : z4;
        const float av[4] = {az.x, az.y, az.z, az.w};
#pragma unroll
        for (int c = 0; c < PN_MAXCT; ++c) {
            const float bv[4] = {b1f[c][kc].x, b1f[c][kc].y, b1f[c][kc].z, b1f[c][kc].w};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bv[s2], acc[c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < PN_MAXCT; ++c) {
        const int ct = wave + 8 * c;
        if (ct < nct) {
            const int col = 16 * ct + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 4 * q4 + r, row = row0 + rr;
                float v = fmaxf(acc[c][r] + bias1[c], 0.f);
                if (row < p.B) {
                    if (p.m1) v = keep1[c][r] ? v * p.scale : 0.f;
                    if (cg == 0) p.y1[(long)row * P + col] = v;
                } else v = 0.f;
                psm[rr * ldh + col] = v;
            }
        }
    }
    __syncthreads();
    // ---- layer 2: this wave's (column tile, K-half)
    f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < PN_MAXK2; ++kc) {
        if (kc_lo + kc < kc_hi) {
            const float4 h4 = *reinterpret_cast<const float4*>(psm + i16 * ldh + 16 * (kc_lo + kc) + 4 * q4);
            const float av[4] = {h4.x, h4.y, h4.z, h4.w};
            const float bv[4] = {b2f[kc].x, b2f[kc].y, b2f[kc].z, b2f[kc].w};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], bv[s2], acc2, 0, 0, 0);
        }
    }
    __syncthreads();                                    // all reads of the y1 tile are done: reuse the buffer for the K-half exchange
    float* red = psm + (wave & 3) * (16 * 17);
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(4 * q4 + r) * 17 + i16] = acc2[r];
    }
    __syncthreads();
    if (kh == 0 && t2_ok) {
        const int col = 16 * t2 + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * q4 + r;
            if (row >= p.B) continue;
            float v = fmaxf(acc2[r] + red[(4 * q4 + r) * 17 + i16] + bias2, 0.f);
            if (p.m2) v = keep2[r] ? v * p.scale : 0.f;
            p.y2[(long)row * P + col] = v;
        }
    }
}

// Returns -1 when the shape is outside the kernel's bounds (the caller then runs the layers one by one), 0 on success.
int prenet2_launch(const float* x, int ldx, int Kin, const float* w1, const float* b1, const float* w2, const float* b2, const uint8_t* m1,
                   const uint8_t* m2, float scale, float* y1, float* y2, int B, int P, hipStream_t s) {
    if (!w1 || !w2 || (P & 15) != 0 || P > 16 * 8 * PN_MAXCT || Kin > 16 * PN_MAXK1 || (Kin & 15) != 0 || (ldx & 3) != 0 || (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2) & 15) != 0) return -1;
    Prenet2 p;
    p.x = x; p.ldx = ldx; p.Kin = Kin; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.m1 = m1; p.m2 = m2; p.scale = scale;
    p.y1 = y1; p.y2 = y2; p.B = B; p.P = P;
    hipLaunchKernelGGL(prenet2_kernel, dim3((B + 15) / 16, PN_NCG), dim3(512), sizeof(float) * 16 * (P + 4), s, p);
    if (hipGetLastError() != hipSuccess) return mtts_fail("launch prenet2_kernel failed");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
static int ls_check_segs(int nseg, const int* K, const int* ld, const char* what) {
    MTTS_REQUIRE(nseg >= 1 && nseg <= 3, "%s: 1..3 K segments", what);
    for (int i = 0; i < nseg; ++i)
        MTTS_REQUIRE(K[i] > 0 && (K[i] & 31) == 0 && (ld[i] & 3) == 0, "%s: segment %d needs K %% 32 == 0 and ld %% 4 == 0 (K=%d ld=%d)", what, i,
                     K[i], ld[i]);
    return 0;
}

MTTS_API int mtts_lstm_step_ksplit(int k_total) { return ls_ksplit((k_total + 31) / 32, 4); }      // upper bound over every nb_max >= 4

MTTS_API long mtts_lstm_step_partial_floats(int B, int H, int k_total) { return (long)mtts_lstm_step_ksplit(k_total) * B * 4 * H; }

MTTS_API long mtts_lstm_packed_weight_bytes(int H, int k_total, int precision) {
    return (long)4 * H * k_total * (precision == 2 ? 6 : precision ? 2 : 4);
}

MTTS_API int mtts_lstm_pack_weights(const LstmPackArgs* args, void* stream) {
    const LstmPackArgs& a = *args;
    MTTS_TRY(ls_check_segs(a.nseg, a.K, a.ldw, "mtts_lstm_pack_weights"));
    MTTS_REQUIRE(a.H > 0 && (a.H & 31) == 0, "mtts_lstm_pack_weights: H must be a multiple of 32 (H=%d)", a.H);
    LsPack p; memset(&p, 0, sizeof(p));
    p.w0 = a.w[0]; p.K0 = a.K[0]; p.ld0 = a.ldw[0];
    p.w1 = a.nseg > 1 ? a.w[1] : a.w[0]; p.K1 = a.nseg > 1 ? a.K[1] : 0; p.ld1 = a.nseg > 1 ? a.ldw[1] : 0;
    p.w2 = a.nseg > 2 ? a.w[2] : a.w[0]; p.K2 = a.nseg > 2 ? a.K[2] : 0; p.ld2 = a.nseg > 2 ? a.ldw[2] : 0;
    p.H = a.H; p.nkb = (p.K0 + p.K1 + p.K2) / 32; p.precision = a.precision; p.dst = a.dst;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(2048), dim3(256), 0, s, p);
    MTTS_CHECK_LAUNCH("lstm_pack_kernel");
    if (a.bias_u) {
        MTTS_REQUIRE(a.b_ih, "mtts_lstm_pack_weights: bias_u needs b_ih");
        hipLaunchKernelGGL(lstm_rows_unit_major_kernel, dim3(16), dim3(256), 0, s, a.b_ih, a.b_hh, 1, a.H, 1, a.bias_u);
        MTTS_CHECK_LAUNCH("lstm_rows_unit_major_kernel");
    }
    return 0;
}

MTTS_API int mtts_lstm_rows_unit_major(const float* src, int ld, int H, int K, float* dst, void* stream) {
    hipLaunchKernelGGL(lstm_rows_unit_major_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, src, (const float*)nullptr, ld, H, K, dst);
    MTTS_CHECK_LAUNCH("lstm_rows_unit_major_kernel");
    return 0;
}

static int ls_marshal(const LstmStepArgs& a, LsGates& g, LsCell& c) {
    MTTS_TRY(ls_check_segs(a.nseg, a.K, a.ldx, "mtts_lstm_step_fwd"));
    MTTS_REQUIRE(a.B > 0 && a.H > 0 && (a.H & 31) == 0, "mtts_lstm_step_fwd: H must be a multiple of 32 (H=%d)", a.H);
    MTTS_REQUIRE(a.w_packed && a.partials && a.c_prev && a.h_out && a.c_out, "mtts_lstm_step_fwd: missing buffers");
    MTTS_REQUIRE(!a.qpart || (a.w_query && (a.A & 15) == 0 && a.A <= 16 * 4 * LC_MAXCT), "mtts_lstm_step_fwd: query partials need A %% 16 == 0, A <= %d",
                 16 * 4 * LC_MAXCT);
    for (int i = 0; i < a.nseg; ++i) MTTS_REQUIRE(((uintptr_t)a.x[i] & 15) == 0, "mtts_lstm_step_fwd: x[%d] must be 16-byte aligned", i);
    memset(&g, 0, sizeof(g));
    g.x0 = a.x[0]; g.K0 = a.K[0]; g.ld0 = a.ldx[0];
    g.x1 = a.nseg > 1 ? a.x[1] : a.x[0]; g.K1 = a.nseg > 1 ? a.K[1] : 0; g.ld1 = a.nseg > 1 ? a.ldx[1] : a.ldx[0];
    g.x2 = a.nseg > 2 ? a.x[2] : a.x[0]; g.K2 = a.nseg > 2 ? a.K[2] : 0; g.ld2 = a.nseg > 2 ? a.ldx[2] : a.ldx[0];
    g.wp = a.w_packed; g.nkb = (g.K0 + g.K1 + g.K2) / 32; g.KS = ls_ksplit(g.nkb, a.nb_max); g.B = a.B; g.N = 4 * a.H; g.part = a.partials;
    g.nbmax = (g.nkb + g.KS - 1) / g.KS;
    memset(&c, 0, sizeof(c));
    c.part = a.partials; c.KS = g.KS; c.B = a.B; c.H = a.H; c.pre = a.pre; c.ldpre = a.ldpre; c.bias_u = a.bias_u;
    c.h_prev = a.h_prev; c.c_prev = a.c_prev; c.h_out = a.h_out; c.c_out = a.c_out; c.gates_out = a.gates_out;
    c.hmask = a.hmask; c.cmask = a.cmask; c.hscale = a.hscale; c.zone = a.zone; c.zh = a.zh; c.zc = a.zc;
    c.wq = a.qpart ? a.w_query : nullptr; c.A = a.A; c.qpart = a.qpart;
    MTTS_REQUIRE(!(c.zone && !c.h_prev), "mtts_lstm_step_fwd: zoneout needs h_prev");
    return 0;
}

static int ls_set_attrs() {
    // the attribute is per DEVICE (one process may drive several): one flag per device ordinal
    static bool attr_done_dev[64] = {false};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    bool& attr_done = attr_done_dev[dev_ & 63];
    if (!attr_done) {
        MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_gates_kernel<0, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 7 * LS_PLANE_B));
        MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_gates_kernel<0, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 10 * LS_PLANE_B));
        attr_done = true;
    }
    return 0;
}

static int lf2_set_attrs() {      // dynamic LDS above 64 KiB needs the per-device opt-in
    static bool done_dev[64] = {false};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    bool& done = done_dev[dev_ & 63];
    if (!done) {
        MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_fused2_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Lf2Geom<2, 2>::SM_BYTES));
        MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_fused2_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Lf2Geom<1, 2>::SM_BYTES));
        MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_fused2_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, Lf2Geom<2, 1>::SM_BYTES));
        MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)lstm_fused2_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, Lf2Geom<1, 1>::SM_BYTES));
        done = true;
    }
    return 0;
}

// batches of more than 64 rows take the fused kernel (lstm_fused_kernel): 64-row workgroups above 128 rows, 32-row ones up to 128
static bool ls_fused_ok(const LstmStepArgs& a) {
    return a.B > 64 && (a.H & 15) == 0 && (!a.qpart || a.A <= 256);
}
// weight-pack mode of a decoder LSTM step.  fp32 steps of the fused kernels take the weights as three pre-split bf16 planes (precision
// 2: six bf16 MFMA terms per fragment pair) instead of fp32 fragments on the fp32 matrix instruction wherever that is faster per launch:
//   lone chain (free-running schedule, lstm_fused2_kernel), every batch above 64 rows: 27.8 vs 34.4 us at batch 240 (K = 1312),
//     19.9 vs 25.6 us at batch 128 (K = 1568), 25.2 vs 34.4 us (K = 2336);
//   two chains side by side (teacher-forced schedule, lstm_fused_kernel): above 128 rows only (64-row workgroups: 31.1 vs 34.4 us at
//     batch 240; with 32-row workgroups six terms + 50 % more weight bytes lose: 27.5 vs 25.6 us at batch 128).
int ls_pack_mode(int B, int precision, bool lone_chain) { return (precision == 0 && (B > 128 || (lone_chain && B > 64))) ? 2 : precision; }

int lstm_step_launch(const LstmStepArgs& a, hipStream_t s) {
    LsGates g; LsCell c;
    MTTS_TRY(ls_marshal(a, g, c));
    MTTS_REQUIRE(a.precision != 2 || ls_fused_ok(a), "mtts_lstm_step_fwd: precision 2 (pre-split weight planes) is the form of batches above 64 rows (B=%d)", a.B);
    if (ls_fused_ok(a)) {
        LsFused f; memset(&f, 0, sizeof(f));
        f.x0 = g.x0; f.x1 = g.x1; f.x2 = g.x2; f.K0 = g.K0; f.K1 = g.K1; f.K2 = g.K2; f.ld0 = g.ld0; f.ld1 = g.ld1; f.ld2 = g.ld2;
        f.wp = g.wp; f.nkb = g.nkb; f.c = c;
        const int nug = a.H / 16;
        // nb_max == 4 is the caller's statement that another step kernel runs beside this one (the two LSTM chains of the teacher-forced
        // schedule on two streams): those launches keep F - 128 VGPRs, 51 KiB of LDS, two workgroups per CU - because the chains hide in
        // each other's stalls (decoder step at batch 240: 86.8 us with F against 91.0 with the faster-alone F2, which owns its CU).
        // A lone chain (the free-running schedule) takes F2.
        const bool lf_old = a.nb_max == 4;
        if (!lf_old && a.precision != 0) MTTS_TRY(lf2_set_attrs());
        if (a.B > 128) {
            const dim3 grid(nug * ((a.B + 63) / 64));
            if (a.precision == 2 && !lf_old) hipLaunchKernelGGL((lstm_fused2_kernel<2, 2>), grid, dim3(LS_THREADS), (Lf2Geom<2, 2>::SM_BYTES), s, f);
            else if (a.precision == 1 && !lf_old) hipLaunchKernelGGL((lstm_fused2_kernel<1, 2>), grid, dim3(LS_THREADS), (Lf2Geom<1, 2>::SM_BYTES), s, f);
            else if (a.precision == 2) hipLaunchKernelGGL((lstm_fused_kernel<2, 2, 4>), grid, dim3(LS_THREADS), 0, s, f);
            else if (a.precision) hipLaunchKernelGGL((lstm_fused_kernel<1, 2, 4>), grid, dim3(LS_THREADS), 0, s, f);
            else hipLaunchKernelGGL((lstm_fused_kernel<0, 2, 4>), grid, dim3(LS_THREADS), 0, s, f);
        } else {
            const dim3 grid(nug * ((a.B + 31) / 32));
            if (a.precision == 2 && !lf_old) hipLaunchKernelGGL((lstm_fused2_kernel<2, 1>), grid, dim3(LS_THREADS), (Lf2Geom<2, 1>::SM_BYTES), s, f);
            else if (a.precision == 1 && !lf_old) hipLaunchKernelGGL((lstm_fused2_kernel<1, 1>), grid, dim3(LS_THREADS), (Lf2Geom<1, 1>::SM_BYTES), s, f);
            else if (a.precision == 2) hipLaunchKernelGGL((lstm_fused_kernel<2, 1, 4>), grid, dim3(LS_THREADS), 0, s, f);
            else if (a.precision) hipLaunchKernelGGL((lstm_fused_kernel<1, 1, 4>), grid, dim3(LS_THREADS), 0, s, f);
            else hipLaunchKernelGGL((lstm_fused_kernel<0, 1, 4>), grid, dim3(LS_THREADS), 0, s, f);
        }
        MTTS_CHECK_LAUNCH("lstm_fused_kernel");
        return 0;
    }
    // one or two rows, inference (no saved gates, no training masks), a lone chain: the GEMV-shaped step (lstm_gemv_kernel)
    if (a.B <= LV_MAXB && a.precision == 0 && !a.gates_out && !a.hmask && !a.cmask && a.zone != 1 && a.nb_max != 4 && (a.H & 31) == 0 &&
        (!a.qpart || a.A == 64 || a.A == 128)) {
        LsGemv v; memset(&v, 0, sizeof(v));
        v.x0 = g.x0; v.x1 = g.x1; v.x2 = g.x2; v.K0 = g.K0; v.K1 = g.K1; v.K2 = g.K2; v.ld0 = g.ld0; v.ld1 = g.ld1; v.ld2 = g.ld2;
        v.wp = reinterpret_cast<const float*>(g.wp); v.nkb = g.nkb; v.c = c;
        const dim3 grid(a.H / 16), blk(LS_THREADS);
        if (a.B <= 1) hipLaunchKernelGGL((lstm_gemv_kernel<1, 8>), grid, blk, 0, s, v);
        else hipLaunchKernelGGL((lstm_gemv_kernel<2, 4>), grid, blk, 0, s, v);
        MTTS_CHECK_LAUNCH("lstm_gemv_kernel");
        return 0;
    }
    MTTS_TRY(ls_set_attrs());
    const int ntile = (4 * a.H) / LS_COLS;
    const dim3 grid(ntile * g.KS), blk(LS_THREADS);
#define LS_LAUNCH(PREC, NB) hipLaunchKernelGGL((lstm_gates_kernel<PREC, NB>), grid, blk, (size_t)(PREC ? 1 : 3) * NB * LS_PLANE_B, s, g)
    if (a.precision) {
        if (g.nbmax <= 4) LS_LAUNCH(1, 4); else if (g.nbmax <= 7) LS_LAUNCH(1, 7); else LS_LAUNCH(1, 10);
    } else {
        if (g.nbmax <= 4) LS_LAUNCH(0, 4); else if (g.nbmax <= 7) LS_LAUNCH(0, 7); else LS_LAUNCH(0, 10);
    }
#undef LS_LAUNCH
    MTTS_CHECK_LAUNCH("lstm_gates_kernel");
    hipLaunchKernelGGL(lstm_cell_q_kernel, dim3(a.H / 16, (a.B + 15) / 16), dim3(256), 0, s, c);
    MTTS_CHECK_LAUNCH("lstm_cell_q_kernel");
    return 0;
}

MTTS_API int mtts_lstm_step_fwd(const LstmStepArgs* args, void* stream) { return lstm_step_launch(*args, (hipStream_t)stream); }
