// fp32 GEMM for gfx950, one argument block, three cores:
//   * gemm_pipe_kernel (default for whole 32-wide K blocks and 16-byte aligned operands; plain GEMMs and the three implicit-GEMM
//     forms of a 1-D convolution): every fp32 operand element is split EXACTLY into three bf16 values x = h1 + h2 + h3
//     (+ < 2^-24 |x|) while its tile is written to LDS, and the product is evaluated on the bf16 matrix cores as
//     a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 with fp32 accumulation (v_mfma_f32_32x32x16_bf16).  The dropped terms are
//     <= 2^-23 |ab|, below one fp32 rounding of the product; measured error against fp64 is at or below that of an fp32 fma
//     chain (tests/test_gpu_more.py, tests/test_gpu_gemm_pipe.py).  One workgroup per CU runs a software pipeline: split, LDS
//     stores, global loads and fragment reads of the next tiles are issued in the shadow of the current tile's MFMAs
//     (generated stream, gemm_pipe_body.inc): 165-192 TFLOP/s-equivalent sustained.
//   * gemm_split_kernel: the same arithmetic, phase-alternating (MFMA | barrier | split + stores | barrier), for every other
//     shape and, with one bf16 plane, for the bf16 path: 145-155 TFLOP/s-equivalent with two workgroups per CU.
//   * gemm_mfma_kernel (MTTS_GEMM_EXACT_F32=1): v_mfma_f32_32x32x2_f32, exact f32 products, 101-112 sustained (157 TF peak).
//
//   C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// One kernel serves every *batched* dense contraction of the text->mel path:
//   * Linear layers            y = x W^T          (A,B both K-contiguous)
//   * input-gradient GEMMs     dx = dy W          (B "transposed": n contiguous)
//   * weight-gradient GEMMs    dW = dy^T x        (A and B "transposed": reduction over rows)
//   * 1-D convolution as an implicit GEMM over channel-last activations [rows=(n,l), C]:
//     K is decomposed as (tap t, channel c); tap t reads activation row r + shift_t and is
//     zero outside the sequence ('same' padding, dilation) -- no im2col buffer is materialised.
//   * grouped convolutions / per-tap weight gradients through the grid.z batch strides.
//
// Tile: 128x128x32 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles),
// register-staged global->LDS double buffering.  LDS image per operand is either
//   KC  [rows][32+4]   (k contiguous; ds_read_b128 feeds 4 MFMA k-steps via the k-slot trick), or
//   MC  [32][128+4]    (row contiguous, for transposed sources; ds_read_b32 per k-step).
// The k-slot trick: instruction s of group g uses slot q (=lane>>5) as k = 8g + 4q + s for BOTH
// operands, so a lane's float4 along k supplies four consecutive instructions.
#include "common.h"
#include <stdlib.h>


constexpr int BM = 128, BN = 128, BK = 32;

// Linear tile index (already XCD-contiguous: every XCD walks a contiguous run of indices) -> tile coordinates.  Row-major order made
// the 32 workgroups an XCD runs at a time cover one tile ROW: one A panel and 32 B panels - the whole B operand passes through every
// XCD's 4 MiB L2 once per tile row.  Column strips of four tiles, row-major inside a strip: the same 32 workgroups cover 8 x 4 tiles
// (8 A panels, 4 B panels), and a strip's four B panels stay L2-resident while its A panels stream by (38400 x 4096: exactly one
// strip per XCD).  Measured on the K loop of the pre-split core: 1.42 -> 1.31 us per step (scripts/mb/mb_gemm_planes.hip).  Results
// do not depend on the order (every tile is computed by one workgroup either way).
__device__ __forceinline__ void gemm_tile_block(int id, int ntx, int nty, int& tile_m, int& tile_n) {
    constexpr int W = 4;
    const int nfull = ntx / W, per_strip = W * nty;
    int strip = id / per_strip, rem = id - strip * per_strip, w = W;
    if (strip >= nfull) { strip = nfull; rem = id - nfull * per_strip; w = ntx - W * nfull; }
    tile_m = rem / w;
    tile_n = strip * W + (rem - tile_m * w);
}
constexpr int KC_LD = BK + 4;     // 36 floats: conflict-free ds_read_b128 (36*i mod 64 distinct over 16 rows)
constexpr int MC_LD = BM + 4;

template <bool T>
struct Tile {                      // registers holding one thread's share of a 128x32 tile (4 float4)
    float4 v[4];
};

// Load one float4 of operand X (rows = BM-tile rows, kk along K).  Generic + slow-path safe.
template <bool TRANS, bool IS_A, bool SPLIT = false>
__device__ __forceinline__ void load_tile(const GemmArgs& p, const float* __restrict__ base, int row0, int k0,
                                          int rows_total, int ld, bool vec_ok, int shift_z, Tile<TRANS>& t) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!TRANS) {
            // 8 threads cover one row's 32 k (128 B), 32 rows per pass
            const int r = row0 + it * 32 + (tid >> 3);
            const int kk = k0 + (tid & 7) * 4;
            if (r < rows_total && kk < p.K) {
                long off; bool valid = true; int kk_in = kk;
                if (IS_A && p.shift_mode == 1) {
                    const int tap = kk / p.Kc; kk_in = kk - tap * p.Kc;
                    const int sh = p.shift0 + tap * p.dshift;
                    const int l = r % p.seq_len;
                    valid = (l + sh >= 0) && (l + sh < p.seq_len);
                    off = (long)(r + sh) * ld + kk_in;
                } else {
                    off = (long)r * ld + kk;
                }
                if (valid) {
                    if (vec_ok && kk + 3 < p.K) {
                        out = *reinterpret_cast<const float4*>(base + off);
                    } else {
                        float tmp[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int j = 0; j < 4; ++j) {
                            if (kk + j < p.K) {
                                if (IS_A && p.shift_mode == 1) {
                                    const int kj = kk + j; const int tap = kj / p.Kc; const int c = kj - tap * p.Kc;
                                    const int sh = p.shift0 + tap * p.dshift; const int l = r % p.seq_len;
                                    if (l + sh >= 0 && l + sh < p.seq_len) tmp[j] = base[(long)(r + sh) * ld + c];
                                } else tmp[j] = base[(long)r * ld + kk + j];
                            }
                        }
                        out = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                    }
                }
            }
        } else {
            // transposed source: element (row, kk) at base[kk*ld + row] (+ tap offset); float4 along rows
            // exact-f32 core: k = it*8 + tid/32, rows 4*(tid%32)..+3;  split core: k = 4*(tid%8) + it, rows 4*(tid/8)..+3
            // (four consecutive k per thread, so that one row's 4 k-values form an 8-byte bf16 store)
            const int kk = k0 + (SPLIT ? (tid & 7) * 4 + it : it * 8 + (tid >> 5));
            const int r = row0 + (SPLIT ? (tid >> 3) * 4 : (tid & 31) * 4);
            if (kk < p.K && r < rows_total) {
                long koff; bool valid = true;
                if (!IS_A && p.shift_mode == 2) {          // wgrad: k index is an activation row, shifted
                    const int l = kk % p.seq_len;
                    valid = (l + shift_z >= 0) && (l + shift_z < p.seq_len);
                    koff = (long)(kk + shift_z) * ld;
                } else if (!IS_A && p.taps > 1 && p.shift_mode == 1) {   // conv bwd-data weights: kk = (tap, c)
                    const int tap = kk / p.Kc; const int c = kk - tap * p.Kc;
                    koff = (long)c * ld + (long)tap * p.b_tap;
                } else {
                    koff = (long)kk * ld;
                }
                if (valid) {
                    if (vec_ok && r + 3 < rows_total) {
                        out = *reinterpret_cast<const float4*>(base + koff + r);
                    } else {
                        float tmp[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int j = 0; j < 4; ++j) if (r + j < rows_total) tmp[j] = base[koff + r + j];
                        out = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                    }
                }
            }
        }
        t.v[it] = out;
    }
}

template <bool TRANS>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const Tile<TRANS>& t) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (!TRANS) {
            const int r = it * 32 + (tid >> 3);
            const int k = (tid & 7) * 4;
            *reinterpret_cast<float4*>(lds + r * KC_LD + k) = t.v[it];
        } else {
            const int k = it * 8 + (tid >> 5);
            const int r = (tid & 31) * 4;
            *reinterpret_cast<float4*>(lds + k * MC_LD + r) = t.v[it];
        }
    }
}

constexpr int LDS_A = (BM * KC_LD > BK * MC_LD) ? BM * KC_LD : BK * MC_LD;   // floats per operand buffer

// The scratch arena for split-K partial tiles is provided by the caller per device / per stream (mtts_set_workspace,
// mtts_set_stream_workspace in common.cpp); the library never allocates.

// C = epilogue(alpha * sum_s partial_s ...) for split-K launches
__global__ void gemm_splitk_reduce(GemmArgs p, const float* __restrict__ ws, int S) {
    const int z = blockIdx.z;
    const int zb = z / p.zt, ztap = z % p.zt;
    float* C = p.C + (long)zb * p.c_z + (long)ztap * p.c_ztap;
    const float* bias = p.bias ? p.bias + (long)zb * p.bias_z : nullptr;
    const long MN = (long)p.M * p.N;
    const float* W = ws + (long)z * S * MN;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    GridRC rc(i0, stride, p.N);
    for (long i = i0; i < MN; i += stride, rc.next()) {
        const long row = rc.row; const int col = rc.col;
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += W[s * MN + i];       // S is 2..8: a batch of clamped requests would re-read the last slab
        float v = p.alpha * a + (bias ? bias[col] : 0.f);
        float* cp = C + row * p.ldc + col;
        if (p.beta != 0.f) v += p.beta * (*cp);
        v = apply_act(p.act, v);
        if (p.mask) v = p.mask[row * p.ldmask + col] ? v * p.mask_scale : 0.f;
        *cp = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// split-bf16 core.  LDS image per operand: 3 planes x [128 rows][32 k] bf16 (64 B rows, no padding); the 16-byte
// chunk c of row r sits at chunk c ^ ((r >> 2) & 3): conflict-free for ds_read_b128 (lane groups of the 32x32x16
// fragment read) and at most 2-way for the 8-byte stores of both the k-contiguous and the transposed fill.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int SP_ROW_B = BK * 2;
constexpr int SP_PLANE_B = BM * SP_ROW_B;        // 8 KiB
constexpr int SP_LDS_B = 6 * SP_PLANE_B;         // A planes 0..2, B planes 3..5: 48 KiB

__device__ __forceinline__ void split_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
    const float rx = x - __uint_as_float(ux & 0xffff0000u), ry = y - __uint_as_float(uy & 0xffff0000u);   // exact
    const unsigned vx = __float_as_uint(rx), vy = __float_as_uint(ry);
    const float sx = rx - __uint_as_float(vx & 0xffff0000u), sy = ry - __uint_as_float(vy & 0xffff0000u);   // exact
    p1 = __builtin_amdgcn_perm(uy, ux, 0x07060302u);       // {hi16(x), hi16(y)}: truncation to bf16
    p2 = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}

__device__ __forceinline__ unsigned rne_pair(float x, float y) {      // two floats -> packed bf16, round to nearest even
    const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
    const unsigned rx = ux + 0x7fffu + ((ux >> 16) & 1u), ry = uy + 0x7fffu + ((uy >> 16) & 1u);
    return __builtin_amdgcn_perm(ry, rx, 0x07060302u);
}

// NPL = 3: exact three-way split (fp32 path); NPL = 1: one bf16 plane, round to nearest even (bf16 path)
template <int NPL>
__device__ __forceinline__ void store_split4(char* lds, int row, int k4, float4 v) {
    char* p = lds + row * SP_ROW_B + (((k4 >> 1) ^ ((row >> 2) & 3)) * 16) + (k4 & 1) * 8;
    if (NPL == 1) {
        *reinterpret_cast<uint2*>(p) = make_uint2(rne_pair(v.x, v.y), rne_pair(v.z, v.w));
        return;
    }
    unsigned a1, a2, a3, b1, b2, b3;
    split_pair(v.x, v.y, a1, a2, a3);
    split_pair(v.z, v.w, b1, b2, b3);
    *reinterpret_cast<uint2*>(p) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(p + SP_PLANE_B) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(p + 2 * SP_PLANE_B) = make_uint2(a3, b3);
}

template <bool TRANS, int NPL>
__device__ __forceinline__ void store_tile_split(char* lds, const Tile<TRANS>& t) {
    const int tid = threadIdx.x;
    if (!TRANS) {
#pragma unroll
        for (int it = 0; it < 4; ++it) store_split4<NPL>(lds, it * 32 + (tid >> 3), tid & 7, t.v[it]);
    } else {      // thread holds k = 4*(tid%8) + it (it = 0..3) of rows 4*(tid/8) + {x,y,z,w}
        store_split4<NPL>(lds, (tid >> 3) * 4 + 0, tid & 7, make_float4(t.v[0].x, t.v[1].x, t.v[2].x, t.v[3].x));
        store_split4<NPL>(lds, (tid >> 3) * 4 + 1, tid & 7, make_float4(t.v[0].y, t.v[1].y, t.v[2].y, t.v[3].y));
        store_split4<NPL>(lds, (tid >> 3) * 4 + 2, tid & 7, make_float4(t.v[0].z, t.v[1].z, t.v[2].z, t.v[3].z));
        store_split4<NPL>(lds, (tid >> 3) * 4 + 3, tid & 7, make_float4(t.v[0].w, t.v[1].w, t.v[2].w, t.v[3].w));
    }
}

// Epilogue arithmetic of one element: alpha, bias, beta * C, activation, keep-mask.  Everything that is uniform over the launch is
// decided outside the element loops (activation kind ACTK 0: none, 1: ReLU, 2: tanh / sigmoid as a template parameter; beta and the
// mask as loop-invariant flags whose loads are skipped and whose arithmetic is branch-free): evaluated per element they cost a lone
// wave ~60 branches per stored row - measured at 11 us per 128x128 tile, as much as eight K blocks.
struct EpiFlags { bool beta, mask; float scale; };
__device__ __forceinline__ EpiFlags epi_flags(const GemmArgs& p) { return {p.beta != 0.f, p.mask != nullptr, p.mask ? p.mask_scale : 1.f}; }
template <int ACTK>
__device__ __forceinline__ float epi_value(const GemmArgs& p, const EpiFlags& f, float acc, float bv, float c, unsigned m) {
    float v = p.alpha * acc + bv;
    v += p.beta * c;                       // c == 0 when beta == 0 (never loaded)
    if (ACTK == 1) v = fmaxf(v, 0.0f);
    if (ACTK == 2) v = apply_act(p.act, v);
    return m ? v * f.scale : 0.f;          // m == 1, scale == 1 without a mask
}

// MFMA layout of a wave's 64x64 block: C/D of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
template <int ACTK>
__device__ __forceinline__ void tile_epilogue_t(const GemmArgs& p, const f32x16 (&acc)[2][2], float* C, const float* bias, int row0, int col0,
                                                int li, int lq) {
    const EpiFlags f = epi_flags(p);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + j * 32 + li;
            if (col >= p.N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
                if (row >= p.M) continue;
                float* cp = C + (long)row * p.ldc + col;
                float c = 0.f; unsigned m = 1;
                if (f.beta) c = *cp;
                if (f.mask) m = p.mask[(long)row * p.ldmask + col];
                *cp = epi_value<ACTK>(p, f, acc[i][j][r], bv, c, m);
            }
        }
}

#define MTTS_EPI_DISPATCH(FN, ...)                                          \
    do {                                                                    \
        if (p.act == MTTS_ACT_NONE) FN<0>(__VA_ARGS__);                      \
        else if (p.act == MTTS_ACT_RELU) FN<1>(__VA_ARGS__);                 \
        else FN<2>(__VA_ARGS__);                                             \
    } while (0)

// Write one wave's 64x64 block of accumulators: raw partial tile into the split-K scratch, or the fused epilogue.
__device__ __forceinline__ void tile_epilogue(const GemmArgs& p, const f32x16 (&acc)[2][2], float* g_ws, float* C, const float* bias,
                                              int z, int row0, int col0, int li, int lq) {
    if (gridDim.y > 1) {
        float* W = g_ws + ((long)z * gridDim.y + blockIdx.y) * (long)p.M * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = col0 + j * 32 + li;
                if (col >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
                    if (row < p.M) W[(long)row * p.N + col] = acc[i][j][r];
                }
            }
        return;
    }
    MTTS_EPI_DISPATCH(tile_epilogue_t, p, acc, C, bias, row0, col0, li, lq);
}

template <bool TA, bool TB, int NPL = 3>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(GemmArgs p, float* g_ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* sa = reinterpret_cast<char*>(smem);
    char* sb = sa + NPL * SP_PLANE_B;

    const int ntx = (p.N + BN - 1) / BN, nty = (p.M + BM - 1) / BM;
    const int nt = ntx * nty;
    int id = blockIdx.x;
    {   // XCD-aware tile order (see gemm_mfma_kernel)
        const int q = nt / 8, r = nt % 8, xcd = id % 8, idx = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m, tile_n;
    gemm_tile_block(id, ntx, nty, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int z = blockIdx.z;
    const int zb = z / p.zt, ztap = z % p.zt;
    const float* A = p.A + (long)zb * p.a_z;
    const float* B = p.B + (long)zb * p.b_z;
    float* C = p.C + (long)zb * p.c_z + (long)ztap * p.c_ztap;
    const float* bias = p.bias ? p.bias + (long)zb * p.bias_z : nullptr;
    const int shift_z = p.shift0 + ztap * p.dshift;

    const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (p.shift_mode == 0 || (p.Kc & 3) == 0) &&
                      ((p.a_z & 3) == 0);
    const bool vecB = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (p.shift_mode == 0 || (p.Kc & 3) == 0) &&
                      ((p.b_z & 3) == 0) && ((p.b_tap & 3) == 0);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int li = lane & 31, lq = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Tile<TA> ta; Tile<TB> tb;
    const int nk_all = (p.K + BK - 1) / BK;
    const int per_split = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kb0 = blockIdx.y * per_split;
    const int nk = min(nk_all, kb0 + per_split);
    load_tile<TA, true, true>(p, A, m0, kb0 * BK, p.M, p.lda, vecA, shift_z, ta);
    load_tile<TB, false, true>(p, B, n0, kb0 * BK, p.N, p.ldb, vecB, shift_z, tb);
    store_tile_split<TA, NPL>(sa, ta);
    store_tile_split<TB, NPL>(sb, tb);
    __syncthreads();

    // fragment addresses: lane reads row (l & 31), k-chunk 2*ks + (l >> 5) of each plane
    int offa[2], offb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm + i * 32 + li, rb = wn + i * 32 + li;
        offa[i] = ra * SP_ROW_B + ((lq ^ ((ra >> 2) & 3)) * 16);
        offb[i] = rb * SP_ROW_B + ((lq ^ ((rb >> 2) & 3)) * 16);
    }

    for (int kb = kb0; kb < nk; ++kb) {
        if (kb + 1 < nk) {
            load_tile<TA, true, true>(p, A, m0, (kb + 1) * BK, p.M, p.lda, vecA, shift_z, ta);
            load_tile<TB, false, true>(p, B, n0, (kb + 1) * BK, p.N, p.ldb, vecB, shift_z, tb);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2][NPL], b[2][NPL];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {      // chunk (2*ks + lq) ^ s == (lq ^ s) ^ 2*ks
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(sa + pl * SP_PLANE_B + (offa[i] ^ (ks * 32)));
                    b[i][pl] = *reinterpret_cast<const bf16x8*>(sb + pl * SP_PLANE_B + (offb[i] ^ (ks * 32)));
                }
#define MTTS_MM(PA, PB)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =      \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0);
            if (NPL == 1) { MTTS_MM(0, 0) }
            else { MTTS_MM(NPL - 1, 0) MTTS_MM(0, NPL - 1) MTTS_MM(NPL / 2, NPL / 2) MTTS_MM(NPL / 2, 0) MTTS_MM(0, NPL / 2) MTTS_MM(0, 0) }
#undef MTTS_MM
        }
        __syncthreads();
        if (kb + 1 < nk) {
            store_tile_split<TA, NPL>(sa, ta);
            store_tile_split<TB, NPL>(sb, tb);
        }
        __syncthreads();
    }

    tile_epilogue(p, acc, g_ws, C, bias, z, m0 + wm, n0 + wn, li, lq);
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_pipe_kernel: the split-bf16 core as a software pipeline inside ONE wave per SIMD.
//
// gemm_split_kernel above alternates phases (48 MFMAs | barrier | split + ds_write | barrier): with one workgroup per CU - the
// only configuration the helper streams are allowed, DESIGN.md 3.1 - the matrix pipe idles during every split phase (MFMA
// busy ~50 %).  Here the LDS image is double buffered (2 x 48 KiB) and the 3-way split of tile kb+1, its LDS stores, the
// global loads of tile kb+2 and the fragment reads are issued IN THE SHADOW of tile kb's 48 MFMAs, about five instructions per
// MFMA (the measured issue budget of a lone wave, MI355X_MICROARCH.md).  The placement is generated
// (scripts/gen_gemm_pipe.py -> gemm_pipe_body.inc) and pinned with sched_barrier(0) after every MFMA group; MFMA order and
// accumulation order equal gemm_split_kernel's, so both cores return bit-identical results.
// Global loads are raw buffer loads (descriptor base per 32-row slab, one VGPR offset per operand, K position in an SGPR):
// no per-load address arithmetic.  Plain GEMMs only (shift_mode 0, K % 32 == 0, 16-byte aligned operands); everything else
// stays on gemm_split_kernel.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int PP_STAGE_B = 3 * SP_PLANE_B;            // one operand, one stage: 3 planes = 24 KiB
constexpr int PP_OPERAND_B = 2 * PP_STAGE_B;          // A: [2 stages][3 planes], B behind it: 96 KiB in all
constexpr unsigned PP_SEL = 0x07060302u, PP_MASK = 0xffff0000u;

struct PipeTmp { unsigned hx[8], hy[8], p1[8], p2[8], p3[8]; float rx[8], ry[8]; };

// The thread's 16 elements are split pair by pair, in processing order n = 0..7.  K-contiguous operand: pair n = float4 `n >> 1`,
// halves (x,y) / (z,w).  Transposed operand: the thread holds k = 4 * (tid % 8) + it of rows 4 * (tid / 8) + {x,y,z,w}; output row u
// needs (k0,k1) and (k2,k3): pairs are taken k-half by k-half (n = 0..3: float4 0,1 of rows 0..3; n = 4..7: float4 2,3), so that
// the first two float4 registers are free for the next loads after four pairs.  Pair index in the temporaries: 2 * row + half.
template <bool T> __device__ __forceinline__ constexpr int pipe_pair(int n) { return T ? 2 * (n & 3) + (n >> 2) : n; }
template <bool T> __device__ __forceinline__ unsigned pipe_x(const u32x4 (&R)[4], int j) { return T ? R[2 * (j & 1)][j >> 1] : R[j >> 1][2 * (j & 1)]; }
template <bool T> __device__ __forceinline__ unsigned pipe_y(const u32x4 (&R)[4], int j) { return T ? R[2 * (j & 1) + 1][j >> 1] : R[j >> 1][2 * (j & 1) + 1]; }

// which float4 of the tile after next may be requested after processing step n (slot 8: right after step 4, slot 9: after step 7)
template <bool T> __device__ __forceinline__ constexpr int pipe_load_slot(int n) {
    return T ? (n == 3 ? 0 : (n == 8 ? 1 : (n == 7 ? 2 : (n == 9 ? 3 : -1)))) : ((n & 1) && n < 8 ? n >> 1 : -1);
}

// LDS stores of a split tile, issued as soon as the planes they need exist.  K-contiguous operand: rows it and it + 1 of one plane
// lie 2 KiB apart (one ds_write2st64_b64): plane 0 after the first stage of pairs 0-3 / 4-7, planes 1 + 2 after their second
// stage.  Transposed operand: row u is complete after second-stage step 4 + u.
template <bool T>
__device__ __forceinline__ void pipe_store1(char* base, const PipeTmp& t, int n) {
    if (!T) {
        const int u = n == 3 ? 0 : 2;
        *reinterpret_cast<uint2*>(base + u * 2048) = make_uint2(t.p1[2 * u], t.p1[2 * u + 1]);
        *reinterpret_cast<uint2*>(base + (u + 1) * 2048) = make_uint2(t.p1[2 * u + 2], t.p1[2 * u + 3]);
    }
}
template <bool T>
__device__ __forceinline__ void pipe_store2(char* base, const PipeTmp& t, int n) {
    if (!T) {
        if (n != 3 && n != 7) return;
        const int u = n == 3 ? 0 : 2;
        *reinterpret_cast<uint2*>(base + SP_PLANE_B + u * 2048) = make_uint2(t.p2[2 * u], t.p2[2 * u + 1]);
        *reinterpret_cast<uint2*>(base + SP_PLANE_B + (u + 1) * 2048) = make_uint2(t.p2[2 * u + 2], t.p2[2 * u + 3]);
        *reinterpret_cast<uint2*>(base + 2 * SP_PLANE_B + u * 2048) = make_uint2(t.p3[2 * u], t.p3[2 * u + 1]);
        *reinterpret_cast<uint2*>(base + 2 * SP_PLANE_B + (u + 1) * 2048) = make_uint2(t.p3[2 * u + 2], t.p3[2 * u + 3]);
    } else if (n >= 4) {
        const int u = n - 4;
        *reinterpret_cast<uint2*>(base + u * SP_ROW_B) = make_uint2(t.p1[2 * u], t.p1[2 * u + 1]);
        *reinterpret_cast<uint2*>(base + u * SP_ROW_B + SP_PLANE_B) = make_uint2(t.p2[2 * u], t.p2[2 * u + 1]);
        *reinterpret_cast<uint2*>(base + u * SP_ROW_B + 2 * SP_PLANE_B) = make_uint2(t.p3[2 * u], t.p3[2 * u + 1]);
    }
}

#define PP_SB __builtin_amdgcn_sched_barrier(0);
#define PP_MFMA(F, i, j, pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F##a[i][pa], F##b[j][pb], acc[i][j], 0, 0, 0);
#define PP_RDA(F, ks, i, pl, STG) F##a[i][pl] = *reinterpret_cast<const bf16x8*>(lds + ((STG) * PP_STAGE_B + (pl) * SP_PLANE_B) + ra[ks][i]);
#define PP_RDB(F, ks, j, pl, STG) F##b[j][pl] = *reinterpret_cast<const bf16x8*>(lds + ((STG) * PP_STAGE_B + (pl) * SP_PLANE_B) + rb[ks][j]);
#define PP_S1A(O, n) { constexpr int j = pipe_pair<T##O>(n); tmp.hx[j] = pipe_x<T##O>(R##O[PP_SET], j) & PP_MASK; tmp.hy[j] = pipe_y<T##O>(R##O[PP_SET], j) & PP_MASK; }
#define PP_S1B(O, n) { constexpr int j = pipe_pair<T##O>(n); const unsigned ux = pipe_x<T##O>(R##O[PP_SET], j), uy = pipe_y<T##O>(R##O[PP_SET], j);   \
        tmp.rx[j] = __uint_as_float(ux) - __uint_as_float(tmp.hx[j]); tmp.ry[j] = __uint_as_float(uy) - __uint_as_float(tmp.hy[j]); \
        tmp.p1[j] = __builtin_amdgcn_perm(uy, ux, PP_SEL); }
#define PP_S2A(O, n) { constexpr int j = pipe_pair<T##O>(n); const unsigned vx = __float_as_uint(tmp.rx[j]), vy = __float_as_uint(tmp.ry[j]); \
        tmp.hx[j] = vx & PP_MASK; tmp.hy[j] = vy & PP_MASK; tmp.p2[j] = __builtin_amdgcn_perm(vy, vx, PP_SEL); }
#define PP_S2B(O, n) { constexpr int j = pipe_pair<T##O>(n);                                                                      \
        const float sx = tmp.rx[j] - __uint_as_float(tmp.hx[j]), sy = tmp.ry[j] - __uint_as_float(tmp.hy[j]);                     \
        tmp.p3[j] = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), PP_SEL); }
#define PP_LD(O, it) R##O[PP_SET][it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc##O[it], voff##O[it], soff##O, 0);
#define PP_LDS(O, n) { constexpr int it = pipe_load_slot<T##O>(n); if constexpr (it >= 0) { PP_LD(O, it) } }
#define PP_NEXT(O) next##O();
#define PP_OFF_A 0
#define PP_OFF_B PP_OPERAND_B
#define PP_ST1(O, n) pipe_store1<T##O>(lds + (PP_NXT * PP_STAGE_B + PP_OFF_##O) + wa##O, tmp, n);
#define PP_ST2(O, n) pipe_store2<T##O>(lds + (PP_NXT * PP_STAGE_B + PP_OFF_##O) + wa##O, tmp, n);
#define PP_BARRIER __syncthreads();

// Epilogue of the pipelined core.  A lone wave per SIMD has nothing to hide store latency behind, so the number of store instructions
// counts: the wave's 64x64 block goes through its private slice of the (now idle) LDS - rows of 72 floats: conflict-free ds_write_b32
// from the MFMA layout, conflict-free ds_read_b128 - and leaves as 16 global_store_dwordx4 per lane (four rows of 256 contiguous
// bytes each) instead of 64 global_store_dword.  Per-tile overhead (scripts/dbg_gemm_k.py: time per tile = a + b * K blocks):
// a = 15.7 us with the scalar, per-element-branching epilogue, 9.8 us now (5.3 us without any epilogue), b = 1.29 us.
// Needs 16-byte aligned rows (leading dimension and N multiples of 4); anything else takes tile_epilogue.
constexpr int EP_LD = 72;
template <int ACTK, int NPS>
__device__ __forceinline__ void pipe_rows_impl(const GemmArgs& p, const float* T, float* out, long ld, const float* bias, int row0, int col,
                                               int c4, int rr) {
    const EpiFlags f = epi_flags(p);
    float4 v[NPS];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) v[ps] = *reinterpret_cast<const float4*>(T + (ps * 4 + rr) * EP_LD + 4 * c4);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = make_float4(bias[col], bias[col + 1], bias[col + 2], bias[col + 3]);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int row = row0 + ps * 4 + rr;
        if (row >= p.M) continue;
        float* cp = out + (long)row * ld + col;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f.beta) c = *reinterpret_cast<const float4*>(cp);
        uchar4 m = make_uchar4(1, 1, 1, 1);
        if (f.mask) m = *reinterpret_cast<const uchar4*>(p.mask + (long)row * p.ldmask + col);
        float4 o;
        o.x = epi_value<ACTK>(p, f, v[ps].x, bv.x, c.x, m.x);
        o.y = epi_value<ACTK>(p, f, v[ps].y, bv.y, c.y, m.y);
        o.z = epi_value<ACTK>(p, f, v[ps].z, bv.z, c.z, m.z);
        o.w = epi_value<ACTK>(p, f, v[ps].w, bv.w, c.w, m.w);
        *reinterpret_cast<float4*>(cp) = o;
    }
}
template <int ACTK>
__device__ __forceinline__ void pipe_epilogue_rows(const GemmArgs& p, const float* T, float* out, long ld, const float* bias, int row0, int col,
                                                   int c4, int rr) {
    pipe_rows_impl<ACTK, 16>(p, T, out, ld, bias, row0, col, c4, rr);
}
template <int ACTK>
__device__ __forceinline__ void pipe_epilogue_rows_half(const GemmArgs& p, const float* T, float* out, long ld, const float* bias, int row0,
                                                        int col, int c4, int rr) {
    pipe_rows_impl<ACTK, 8>(p, T, out, ld, bias, row0, col, c4, rr);
}

__device__ __forceinline__ void pipe_epilogue(const GemmArgs& p, const f32x16 (&acc)[2][2], float* lds_f, float* g_ws, float* C,
                                              const float* bias, int z, int row0, int col0, int lane, int wave) {
    const bool raw = gridDim.y > 1;
    float* out = raw ? g_ws + ((long)z * gridDim.y + blockIdx.y) * (long)p.M * p.N : C;
    const long ld = raw ? p.N : p.ldc;
    const bool vec = (ld & 3) == 0 && (p.N & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                     (raw || !p.mask || ((p.ldmask & 3) == 0 && (reinterpret_cast<uintptr_t>(p.mask) & 3) == 0));
    if (!vec) {
        tile_epilogue(p, acc, g_ws, C, bias, z, row0, col0, lane & 31, lane >> 5);
        return;
    }
    __syncthreads();                      // every wave has read its last fragments: the stages may be overwritten
    float* T = lds_f + wave * (64 * EP_LD);
    const int li = lane & 31, lq = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq) * EP_LD + j * 32 + li] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();      // same wave, in-order LDS queue: the reads below see the stores above
    const int c4 = lane & 15, rr = lane >> 4;
    const int col = col0 + 4 * c4;
    if (col >= p.N) return;
    if (raw) {
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int rl = ps * 4 + rr;
            if (row0 + rl < p.M) *reinterpret_cast<float4*>(out + (long)(row0 + rl) * ld + col) = *reinterpret_cast<const float4*>(T + rl * EP_LD + 4 * c4);
        }
        return;
    }
    MTTS_EPI_DISPATCH(pipe_epilogue_rows, p, T, out, ld, bias, row0, col, c4, rr);
}

#ifndef MTTS_PIPE_SETS
#define MTTS_PIPE_SETS 1
#endif
#define PP_SET (PP_NXT % MTTS_PIPE_SETS)
#ifndef MTTS_PIPE_BODY
#define MTTS_PIPE_BODY "gemm_pipe_body.inc"      // scripts/ab_gemm_stream.sh builds variants of the stream
#endif

// CONV: 0 plain GEMM; the implicit-GEMM forms of a 1-D convolution over channel-last activations (kernels.conv1d_fwd / conv1d_bwd):
//   1 forward  (shift_mode 1): A = activations, K = (tap, channel); tap t reads row r + shift_t, zero outside the sequence
//   2 dgrad    (shift_mode 1, transposed weights with a tap stride b_tap)
//   3 wgrad    (shift_mode 2): both operands transposed, K = activation rows; B's row k is read at k + shift_z, zero outside
// A K block never straddles two taps (Kc % 32 == 0).  Rows that must read as zero get the offset 0x80000000: beyond every
// descriptor's num_records, the buffer load returns zeros without touching memory.
template <bool TA, bool TB, int CONV = 0>
__global__ __launch_bounds__(256, 2) void gemm_pipe_kernel(GemmArgs p, float* g_ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);

    const int ntx = (p.N + BN - 1) / BN, nty = (p.M + BM - 1) / BM;
    const int nt = ntx * nty;
    int id = blockIdx.x;
    {   // XCD-aware tile order (see gemm_mfma_kernel)
        const int q = nt / 8, r = nt % 8, xcd = id % 8, idx = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m, tile_n;
    gemm_tile_block(id, ntx, nty, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int z = blockIdx.z;
    const int zb = z / p.zt, ztap = z % p.zt;
    const float* A = p.A + (long)zb * p.a_z;
    const float* B = p.B + (long)zb * p.b_z;
    float* C = p.C + (long)zb * p.c_z + (long)ztap * p.c_ztap;
    const float* bias = p.bias ? p.bias + (long)zb * p.bias_z : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int li = lane & 31, lq = lane >> 5;

    // K range of this workgroup (split-K over blockIdx.y)
    const int nk_all = p.K / BK;
    const int per_split = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kb0 = blockIdx.y * per_split;
    const int nk = min(nk_all, kb0 + per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kb0 < nk) {
        // ---- global side: descriptor `it` starts at the operand's slab `it` of this tile; a thread's offset inside every slab is
        // the same, the K position rides in the scalar offset.  Plain GEMM: extents are exact, so rows past the end of a K-contiguous
        // operand read as zero without touching memory; a transposed operand clamps its row group (M % 4 == 0 there).
        constexpr unsigned OOB = 0x80000000u;
        const bool shiftA = CONV == 1 || CONV == 2, shiftB = CONV == 3;
        const long extA = shiftA ? 0x7fffffffL : (TA ? ((long)(p.K - 1) * p.lda + p.M) * 4 : ((long)(p.M - 1) * p.lda + p.K) * 4);
        const long extB = (shiftB || CONV == 2) ? 0x7fffffffL : (TB ? ((long)(p.K - 1) * p.ldb + p.N) * 4 : ((long)(p.N - 1) * p.ldb + p.K) * 4);
        // shifted operands: the most negative row shift goes into the descriptor base so that the scalar offset stays >= 0
        const int sh_first = p.shift0, sh_last = p.shift0 + (p.taps - 1) * p.dshift;
        const int min_sh = shiftA ? min(sh_first, sh_last) : 0;
        const int shift_z = p.shift0 + ztap * p.dshift;
        __amdgpu_buffer_rsrc_t rsrcA[4], rsrcB[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long sa = (TA ? (long)it * p.lda + m0 : ((long)m0 + it * 32 + min_sh) * p.lda) * 4;
            const long sb = (TB ? ((long)it + (shiftB ? shift_z : 0)) * p.ldb + n0 : ((long)n0 + it * 32) * p.ldb) * 4;
            rsrcA[it] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A)) + sa, 0,
                                                          (int)(shiftA ? extA : max(0L, extA - sa)), 0x00020000);
            rsrcB[it] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(B)) + sb, 0,
                                                          (int)((shiftB || CONV == 2) ? extB : max(0L, extB - sb)), 0x00020000);
        }
        const int q8 = tid >> 3, k8 = tid & 7;
        const unsigned voffA0 = TA ? (unsigned)(((long)(4 * k8) * p.lda + min(4 * q8, max(p.M - m0 - 4, 0))) * 4)
                                   : (unsigned)(((long)q8 * p.lda + 4 * k8) * 4);
        const unsigned voffB0 = TB ? (unsigned)(((long)(4 * k8) * p.ldb + min(4 * q8, max(p.N - n0 - 4, 0))) * 4)
                                   : (unsigned)(((long)q8 * p.ldb + 4 * k8) * 4);
        unsigned voffA[4] = {voffA0, voffA0, voffA0, voffA0}, voffB[4] = {voffB0, voffB0, voffB0, voffB0};
        const unsigned stepA = TA ? (unsigned)p.lda * (BK * 4) : BK * 4, stepB = TB ? (unsigned)p.ldb * (BK * 4) : BK * 4;
        const unsigned lastA = (unsigned)(nk - 1) * stepA, lastB = (unsigned)(nk - 1) * stepB;
        unsigned soffA = (unsigned)kb0 * stepA, soffB = (unsigned)kb0 * stepB;

        // K-position state of the shifted forms
        const int nbt = CONV == 1 || CONV == 2 ? p.Kc / BK : 1;        // K blocks per tap
        int cbA = kb0 % nbt, tapA = kb0 / nbt, cbB = cbA, tapB = tapA;   // block inside the tap, tap
        int leftA = nk - kb0 - 1, leftB = leftA;                         // further blocks the load stream may advance to
        int lA[4];                                                       // CONV 1, 2: sequence position of the thread's rows
        unsigned l0B = 0;                                                // CONV 3: sequence position of the thread's first k row
        auto set_voffA = [&](int sh) {                                   // rows whose shifted source lies outside the sequence read zero
#pragma unroll
            for (int it = 0; it < 4; ++it) voffA[it] = (unsigned)(lA[it] + sh) < (unsigned)p.seq_len ? voffA0 : OOB;
        };
        auto set_voffB = [&]() {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const unsigned w = l0B + it, t = min(w, w - (unsigned)p.seq_len);          // (l0B + it) mod seq_len
                voffB[it] = (t + (unsigned)shift_z) < (unsigned)p.seq_len ? voffB0 : OOB;
            }
        };
        if (shiftA) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = m0 + it * 32 + q8;
                lA[it] = r < p.M ? r % p.seq_len : 0x40000000;
            }
            const int sh = p.shift0 + tapA * p.dshift;
            soffA = (unsigned)(((long)(sh - min_sh) * p.lda + cbA * BK) * 4);
            set_voffA(sh);
        }
        if (CONV == 2) soffB = (unsigned)(((long)cbB * BK * p.ldb + (long)tapB * p.b_tap) * 4);
        if (shiftB) { l0B = (unsigned)((kb0 * BK + 4 * k8) % p.seq_len); set_voffB(); }
        auto nextA = [&]() {
            if (!shiftA) { soffA = min(soffA + stepA, lastA); return; }
            if (leftA <= 0) return;
            --leftA;
            if (++cbA == nbt) {
                cbA = 0; ++tapA;
                const int sh = p.shift0 + tapA * p.dshift;
                soffA = (unsigned)((long)(sh - min_sh) * p.lda * 4);
                set_voffA(sh);
            } else soffA += BK * 4;
        };
        auto nextB = [&]() {
            if (CONV == 2) {
                if (leftB <= 0) return;
                --leftB;
                if (++cbB == nbt) { cbB = 0; ++tapB; soffB = (unsigned)((long)tapB * p.b_tap * 4); }
                else soffB += stepB;
            } else if (shiftB) {
                if (leftB <= 0) return;
                --leftB;
                soffB += stepB;
                l0B += BK; l0B = min(l0B, l0B - (unsigned)p.seq_len);          // seq_len >= 32: one wrap at most
                set_voffB();
            } else soffB = min(soffB + stepB, lastB);
        };

        // ---- LDS side.  Store address of the thread's first row (pipe_store adds rows / planes); fragment read addresses.
        const int wrA = TA ? 4 * q8 : q8, wrB = TB ? 4 * q8 : q8;
        const unsigned waA = wrA * SP_ROW_B + (((k8 >> 1) ^ ((wrA >> 2) & 3)) * 16) + (k8 & 1) * 8;
        const unsigned waB = wrB * SP_ROW_B + (((k8 >> 1) ^ ((wrB >> 2) & 3)) * 16) + (k8 & 1) * 8;
        unsigned ra[2][2], rb[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rowa = wm + i * 32 + li, rowb = wn + i * 32 + li;
                ra[ks][i] = rowa * SP_ROW_B + (((2 * ks + lq) ^ ((rowa >> 2) & 3)) * 16);
                rb[ks][i] = rowb * SP_ROW_B + (((2 * ks + lq) ^ ((rowb >> 2) & 3)) * 16) + PP_OPERAND_B;
            }

        // MTTS_PIPE_SETS register sets per operand.  1 (default): a tile is requested one K block (~1.3 us) before it is split.
        // 2: tile kb0 + i lives in set i & 1 and is requested two blocks ahead (+32 VGPRs; no gain in isolation, 18 % of the wave
        // cycles are s_waitcnt either way - they wait on LDS, not on memory).
        u32x4 RA[MTTS_PIPE_SETS][4], RB[MTTS_PIPE_SETS][4];
        PipeTmp tmp;
        bf16x8 f0a[2][3], f0b[2][3], f1a[2][3], f1b[2][3];

        // prologue: tile kb0 -> stage 0, the next MTTS_PIPE_SETS tiles in flight, first fragments
#define PP_NXT 0
#pragma unroll
        for (int it = 0; it < 4; ++it) { PP_LD(A, it) PP_LD(B, it) }
        PP_NEXT(A) PP_NEXT(B)
#undef PP_NXT
#if MTTS_PIPE_SETS == 2
#define PP_NXT 1
#pragma unroll
        for (int it = 0; it < 4; ++it) { PP_LD(A, it) PP_LD(B, it) }
        PP_NEXT(A) PP_NEXT(B)
#undef PP_NXT
#endif
#define PP_NXT 0
#define PP_FILL(O, n) PP_S1A(O, n) PP_S1B(O, n) PP_S2A(O, n) PP_S2B(O, n)
#define PP_FILL_ALL(O) PP_FILL(O, 0) PP_FILL(O, 1) PP_FILL(O, 2) PP_FILL(O, 3) PP_FILL(O, 4) PP_FILL(O, 5) PP_FILL(O, 6) PP_FILL(O, 7)          \
        PP_ST1(O, 3) PP_ST1(O, 7) PP_ST2(O, 3) PP_ST2(O, 4) PP_ST2(O, 5) PP_ST2(O, 6) PP_ST2(O, 7)
        PP_FILL_ALL(A)
        PP_FILL_ALL(B)
#pragma unroll
        for (int it = 0; it < 4; ++it) { PP_LD(A, it) PP_LD(B, it) }
        PP_NEXT(A) PP_NEXT(B)
#undef PP_NXT
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)       // every ks=0 fragment of stage 0 (the stream re-reads all but the first four: harmless)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { PP_RDA(f0, 0, i, pl, 0) PP_RDB(f0, 0, i, pl, 0) }

        // K blocks in pairs (the stage is a compile-time constant inside each copy of the stream), an odd last block after the loop
        for (int n = (nk - kb0) >> 1; n > 0; --n) {
#define PP_CUR 0
#define PP_NXT 1
#include MTTS_PIPE_BODY
#undef PP_CUR
#undef PP_NXT
#define PP_CUR 1
#define PP_NXT 0
#include MTTS_PIPE_BODY
#undef PP_CUR
#undef PP_NXT
        }
        if ((nk - kb0) & 1) {
#define PP_CUR 0
#define PP_NXT 1
#include MTTS_PIPE_BODY
#undef PP_CUR
#undef PP_NXT
        }
    }
#ifdef MTTS_PIPE_NO_EPILOGUE          // ablation build (scripts/build_gemm_variant.sh): what the epilogue costs per tile
    if (acc[0][0][0] == 123.456f) C[0] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];
#elif defined(MTTS_PIPE_SCALAR_EPILOGUE)
    tile_epilogue(p, acc, g_ws, C, bias, z, m0 + wm, n0 + wn, li, lq);
#else
    pipe_epilogue(p, acc, smem, g_ws, C, bias, z, m0 + wm, n0 + wn, lane, wave);
#endif
}


template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmArgs p, float* g_ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // buffer b: A at smem + 2*b*LDS_A, B right behind it

    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); remap so that each XCD
    // walks a contiguous run of tiles (neighbouring tiles share A/B panels in that XCD's L2).
    const int ntx = (p.N + BN - 1) / BN, nty = (p.M + BM - 1) / BM;
    const int nt = ntx * nty;
    int id = blockIdx.x;
    {
        const int q = nt / 8, r = nt % 8, xcd = id % 8, idx = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m, tile_n;
    gemm_tile_block(id, ntx, nty, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int z = blockIdx.z;
    const int zb = z / p.zt, ztap = z % p.zt;
    const float* A = p.A + (long)zb * p.a_z;
    const float* B = p.B + (long)zb * p.b_z;
    float* C = p.C + (long)zb * p.c_z + (long)ztap * p.c_ztap;
    const float* bias = p.bias ? p.bias + (long)zb * p.bias_z : nullptr;
    const int shift_z = p.shift0 + ztap * p.dshift;

    const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (p.shift_mode == 0 || (p.Kc & 3) == 0) &&
                      ((p.a_z & 3) == 0);
    const bool vecB = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (p.shift_mode == 0 || (p.Kc & 3) == 0) &&
                      ((p.b_z & 3) == 0) && ((p.b_tap & 3) == 0);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int li = lane & 31, lq = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Tile<TA> ta; Tile<TB> tb;
    // split-K: blockIdx.y owns the K blocks [kb0, nk)
    const int nk_all = (p.K + BK - 1) / BK;
    const int per_split = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kb0 = blockIdx.y * per_split;
    const int nk = min(nk_all, kb0 + per_split);
    load_tile<TA, true>(p, A, m0, kb0 * BK, p.M, p.lda, vecA, shift_z, ta);
    load_tile<TB, false>(p, B, n0, kb0 * BK, p.N, p.ldb, vecB, shift_z, tb);
    store_tile<TA>(smem, ta);
    store_tile<TB>(smem + LDS_A, tb);
    __syncthreads();

    for (int kb = kb0; kb < nk; ++kb) {
        const int cur = (kb - kb0) & 1;
        if (kb + 1 < nk) {
            load_tile<TA, true>(p, A, m0, (kb + 1) * BK, p.M, p.lda, vecA, shift_z, ta);
            load_tile<TB, false>(p, B, n0, (kb + 1) * BK, p.N, p.ldb, vecB, shift_z, tb);
        }
        const float* as = smem + cur * 2 * LDS_A;
        const float* bs = as + LDS_A;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 a[2], b[2];
            float as_[2][4], bs_[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!TA) {
                    a[i] = *reinterpret_cast<const float4*>(as + (wm + i * 32 + li) * KC_LD + g * 8 + lq * 4);
                    as_[i][0] = a[i].x; as_[i][1] = a[i].y; as_[i][2] = a[i].z; as_[i][3] = a[i].w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) as_[i][s] = as[(g * 8 + lq * 4 + s) * MC_LD + wm + i * 32 + li];
                }
                if (!TB) {
                    b[i] = *reinterpret_cast<const float4*>(bs + (wn + i * 32 + li) * KC_LD + g * 8 + lq * 4);
                    bs_[i][0] = b[i].x; bs_[i][1] = b[i].y; bs_[i][2] = b[i].z; bs_[i][3] = b[i].w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bs_[i][s] = bs[(g * 8 + lq * 4 + s) * MC_LD + wn + i * 32 + li];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(as_[i][s], bs_[j][s], acc[i][j], 0, 0, 0);
        }
        if (kb + 1 < nk) {
            store_tile<TA>(smem + (cur ^ 1) * 2 * LDS_A, ta);
            store_tile<TB>(smem + (cur ^ 1) * 2 * LDS_A + LDS_A, tb);
        }
        __syncthreads();
    }

    // C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (gridDim.y > 1) {      // split-K: raw partial tile into the workspace, epilogue in gemm_splitk_reduce
        float* W = g_ws + ((long)z * gridDim.y + blockIdx.y) * (long)p.M * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn + j * 32 + li;
                if (col >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
                    if (row < p.M) W[(long)row * p.N + col] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + j * 32 + li;
            if (col >= p.N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
                if (row >= p.M) continue;
                float v = p.alpha * acc[i][j][r] + bv;
                float* cp = C + (long)row * p.ldc + col;
                if (p.beta != 0.f) v += p.beta * (*cp);
                v = apply_act(p.act, v);
                if (p.mask) v = p.mask[(long)row * p.ldmask + col] ? v * p.mask_scale : 0.f;
                *cp = v;
            }
        }
}

#include "gemm_planes.h"

static int g_default_precision = 0;
MTTS_API int mtts_set_precision(int precision) {
    MTTS_REQUIRE(precision == 0 || precision == 1, "mtts_set_precision: 0 (fp32) or 1 (bf16)");
    g_default_precision = precision;
    return 0;
}
MTTS_API int mtts_get_precision(void) { return g_default_precision; }

MTTS_API int mtts_gemm_ex(const GemmArgs* args, void* stream) {
    GemmArgs p = *args;
    p.precision = p.precision == 0 ? g_default_precision : (p.precision == 1 ? 1 : 0);
    if (p.M <= 0 || p.N <= 0) return 0;
    MTTS_REQUIRE(p.K >= 0 && p.taps >= 1 && p.taps * p.Kc == p.K, "mtts_gemm_ex: K=%d must equal taps*Kc=%d*%d", p.K,
                 p.taps, p.Kc);
    if (p.batch < 1) p.batch = 1;
    if (p.zt < 1) p.zt = 1;
    MTTS_REQUIRE(p.shift_mode == 0 || p.seq_len > 0, "mtts_gemm_ex: shift_mode needs seq_len");
    const int ntx = cdiv(p.N, BN), nty = cdiv(p.M, BM);
    // split-K when few output tiles face a long reduction: the kernel needs >= 2 workgroups per CU to keep the matrix pipe
    // busy (one workgroup alone spends 2/3 of a k-step outside its MFMA phase), i.e. >= 512 workgroups.
    // The scratch arena is cut into three regions so that launches of the caller's stream (nosplit 0), of the side stream
    // (nosplit 1) and of the weight-gradient stream (nosplit 2) never share partial tiles.
    int S = 1;
    size_t g_ws_bytes = 0;
    float* g_ws_host = workspace_for((hipStream_t)stream, &g_ws_bytes);
    // will the pre-split core really run?  (not during stream capture: its pack buffer may have to grow, which synchronises) - known
    // BEFORE the split-K factor is chosen, so that a launch that falls back to the split core keeps its split-K
    static const bool exact_f32 = [] { const char* e = getenv("MTTS_GEMM_EXACT_F32"); return e && e[0] == '1'; }();
    bool planes_can = !exact_f32 && planes_wanted(p, p.precision == 1);
    if (planes_can) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        planes_can = cap == hipStreamCaptureStatusNone;
    }
    const int region = p.nosplit < 0 || p.nosplit > 2 ? 0 : p.nosplit;
    const size_t region_bytes = (g_ws_bytes / 3) & ~(size_t)255;
    {
        const long tiles = (long)ntx * nty * p.batch * p.zt;
        const int nkb = cdiv(p.K, BK);
        if (g_ws_host && tiles < 512 && nkb >= 32) {
            const long target = region == 0 ? 1024L : 512L;      // workgroups wanted on the caller's stream / a helper stream
            S = (int)((target + tiles - 1) / tiles);
            if (S > nkb / 16) S = nkb / 16;
            if (S > 32) S = 32;
            while (S > 1 && (size_t)S * p.batch * p.zt * p.M * p.N * sizeof(float) > region_bytes) --S;
            if (S < 1) S = 1;
        }
        // bf16 GEMMs on the pre-split core: a K step is 96 wide and four times faster, so the split's partial slabs and its reduction
        // launch (20 us for a 4096 x 1568 weight gradient) cost more than the second half-round of workgroups they fill
        static const int planes_split = [] { const char* e = getenv("MTTS_PLANES_SPLITK"); return e ? atoi(e) : 0; }();
        static const int nosplit_tiles = [] { const char* e = getenv("MTTS_PLANES_NOSPLIT_TILES"); return e ? atoi(e) : 128; }();
        // (the same rule for the fp32 cores was measured and not kept: 4096 x 1024 x 3072 0.166 -> 0.138 ms per call, but 4096 x 1536 x 38400
        //  2.73 -> 3.04 and no change of the train step: profiles/r05_gemm_core.txt)
        if (S > 1 && !planes_split && p.precision == 1 && tiles >= nosplit_tiles && planes_can) S = 1;
    }
    dim3 grid(ntx * nty, S, p.batch * p.zt);
    // side-stream launches (nosplit) ask for > half of the CU's LDS so that only ONE GEMM workgroup sits on a CU and the
    // latency-critical step kernels of the main stream always find room next to it
    const size_t lds_base = exact_f32 ? 4 * LDS_A * sizeof(float) : (size_t)SP_LDS_B;
    // Helper-stream launches (side / weight-gradient stream) ask for 96 KiB of LDS: ONE GEMM workgroup per CU.  Every step
    // kernel of the decoder chains is built to fit beside it (<= 128 VGPRs, <= 64 KiB LDS: lstm_gates_kernel<., 4>,
    // skinny_kernel_lo, the attention kernels), so the latency-critical chain never waits for a 100-300 us GEMM workgroup to
    // retire.  Same-box A/B, 20 steps: 89.2-89.6 ms per train step with the reservation, 91.4-91.9 without, 95.5 with round 1's
    // 150-VGPR step kernels.
    const size_t lds = p.nosplit ? (size_t)96 * 1024 : lds_base;
    hipStream_t s = (hipStream_t)stream;
    // the attribute is per DEVICE (one process may drive several): one flag per device ordinal
    static bool attr_done_dev[64] = {false};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    bool& attr_done = attr_done_dev[dev_ & 63];
    if (!attr_done) {   // > 64 KiB of dynamic LDS needs the opt-in attribute
        const void* kernels[15] = {(const void*)gemm_pipe_kernel<false, false, 1>, (const void*)gemm_pipe_kernel<false, true, 2>,
                                   (const void*)gemm_pipe_kernel<true, true, 3>,(const void*)gemm_mfma_kernel<false, false>, (const void*)gemm_mfma_kernel<false, true>,
                                   (const void*)gemm_mfma_kernel<true, false>, (const void*)gemm_mfma_kernel<true, true>,
                                   (const void*)gemm_split_kernel<false, false>, (const void*)gemm_split_kernel<false, true>,
                                   (const void*)gemm_split_kernel<true, false>, (const void*)gemm_split_kernel<true, true>,
                                   (const void*)gemm_pipe_kernel<false, false>, (const void*)gemm_pipe_kernel<false, true>,
                                   (const void*)gemm_pipe_kernel<true, false>, (const void*)gemm_pipe_kernel<true, true>};
        for (const void* k : kernels) MTTS_CHECK_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_done = true;
    }
    float* ws = g_ws_host ? g_ws_host + (size_t)region * (region_bytes / sizeof(float)) : nullptr;
    // plain fp32 GEMMs with whole K blocks and 16-byte aligned operands run on the software-pipelined core (MTTS_GEMM_PIPE=0: off)
    static const bool pipe_on = [] { const char* e = getenv("MTTS_GEMM_PIPE"); return !(e && e[0] == '0'); }();
    const auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const long ext_a = p.transA ? ((long)(p.K - 1) * p.lda + p.M) : ((long)(p.M - 1) * p.lda + p.K);
    const long ext_b = p.transB ? ((long)(p.K - 1) * p.ldb + p.N) : ((long)(p.N - 1) * p.ldb + p.K);
    const bool pipe_common = pipe_on && !exact_f32 && p.precision != 1 && p.K >= BK && p.K % BK == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0 &&
                             (p.a_z & 3) == 0 && (p.b_z & 3) == 0 && aligned16(p.A) && aligned16(p.B) && (!p.transA || (p.M & 3) == 0) &&
                             (!p.transB || (p.N & 3) == 0) && ext_a < (1L << 29) && ext_b < (1L << 29);
    // 0: plain, 1-3: the convolution forms (see gemm_pipe_kernel), -1: not a shape of the pipelined core
    int pipe = -1;
    if (pipe_common) {
        if (p.shift_mode == 0 && p.taps == 1 && p.lda >= (p.transA ? p.M : p.K) && p.ldb >= (p.transB ? p.N : p.K)) pipe = 0;
        else if (p.shift_mode == 1 && !p.transA && p.Kc % BK == 0 && !p.transB) pipe = 1;
        else if (p.shift_mode == 1 && !p.transA && p.Kc % BK == 0 && p.transB && (p.b_tap & 3) == 0) pipe = 2;
        else if (p.shift_mode == 2 && p.transA && p.transB && p.taps == 1 && p.seq_len >= BK) pipe = 3;
    }
    // big plain GEMMs: operands split / rounded ONCE by a pack pass, K loop without vector arithmetic (gemm_planes.h)
    bool planes = false;
    if (planes_can) {
        if (p.precision == 1) MTTS_TRY(planes_gemm<true>(p, grid, ws, s, &planes));
        else MTTS_TRY(planes_gemm<false>(p, grid, ws, s, &planes));
    }
    if (planes) {
    } else if (pipe >= 0) {
        const size_t ldsp = 2 * PP_OPERAND_B;
        if (pipe == 1) hipLaunchKernelGGL((gemm_pipe_kernel<false, false, 1>), grid, dim3(256), ldsp, s, p, ws);
        else if (pipe == 2) hipLaunchKernelGGL((gemm_pipe_kernel<false, true, 2>), grid, dim3(256), ldsp, s, p, ws);
        else if (pipe == 3) hipLaunchKernelGGL((gemm_pipe_kernel<true, true, 3>), grid, dim3(256), ldsp, s, p, ws);
        else if (!p.transA && !p.transB) hipLaunchKernelGGL((gemm_pipe_kernel<false, false>), grid, dim3(256), ldsp, s, p, ws);
        else if (!p.transA && p.transB) hipLaunchKernelGGL((gemm_pipe_kernel<false, true>), grid, dim3(256), ldsp, s, p, ws);
        else if (p.transA && !p.transB) hipLaunchKernelGGL((gemm_pipe_kernel<true, false>), grid, dim3(256), ldsp, s, p, ws);
        else hipLaunchKernelGGL((gemm_pipe_kernel<true, true>), grid, dim3(256), ldsp, s, p, ws);
    } else if (exact_f32) {
        if (!p.transA && !p.transB) hipLaunchKernelGGL((gemm_mfma_kernel<false, false>), grid, dim3(256), lds, s, p, ws);
        else if (!p.transA && p.transB) hipLaunchKernelGGL((gemm_mfma_kernel<false, true>), grid, dim3(256), lds, s, p, ws);
        else if (p.transA && !p.transB) hipLaunchKernelGGL((gemm_mfma_kernel<true, false>), grid, dim3(256), lds, s, p, ws);
        else hipLaunchKernelGGL((gemm_mfma_kernel<true, true>), grid, dim3(256), lds, s, p, ws);
    } else if (p.precision == 1) {     // bf16 path: operands rounded to one bf16 plane, one MFMA product, fp32 accumulation
        const size_t lds1 = SP_LDS_B / 3;
        if (!p.transA && !p.transB) hipLaunchKernelGGL((gemm_split_kernel<false, false, 1>), grid, dim3(256), lds1, s, p, ws);
        else if (!p.transA && p.transB) hipLaunchKernelGGL((gemm_split_kernel<false, true, 1>), grid, dim3(256), lds1, s, p, ws);
        else if (p.transA && !p.transB) hipLaunchKernelGGL((gemm_split_kernel<true, false, 1>), grid, dim3(256), lds1, s, p, ws);
        else hipLaunchKernelGGL((gemm_split_kernel<true, true, 1>), grid, dim3(256), lds1, s, p, ws);
    } else {
        if (!p.transA && !p.transB) hipLaunchKernelGGL((gemm_split_kernel<false, false>), grid, dim3(256), lds, s, p, ws);
        else if (!p.transA && p.transB) hipLaunchKernelGGL((gemm_split_kernel<false, true>), grid, dim3(256), lds, s, p, ws);
        else if (p.transA && !p.transB) hipLaunchKernelGGL((gemm_split_kernel<true, false>), grid, dim3(256), lds, s, p, ws);
        else hipLaunchKernelGGL((gemm_split_kernel<true, true>), grid, dim3(256), lds, s, p, ws);
    }
    MTTS_CHECK_LAUNCH("gemm_mfma_kernel");
    if (S > 1) {
        const long MN = (long)p.M * p.N;
        int blocks = (int)((MN + 255) / 256); if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gemm_splitk_reduce, dim3(blocks, 1, p.batch * p.zt), dim3(256), 0, s, p, ws, S);
        MTTS_CHECK_LAUNCH("gemm_splitk_reduce");
    }
    return 0;
}

// Plain GEMM convenience wrapper used by host code in this library.
int gemm_plain(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, bool tA,
               bool tB, float alpha, float beta, const float* bias, int act, hipStream_t s) {
    GemmArgs p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.transA = tA; p.transB = tB; p.taps = 1; p.Kc = K; p.batch = 1; p.zt = 1;
    p.alpha = alpha; p.beta = beta; p.act = act; p.mask_scale = 1.f;
    return mtts_gemm_ex(&p, s);
}

MTTS_API int mtts_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       int transA, int transB, float alpha, float beta, const float* bias, int act, void* stream) {
    return gemm_plain(A, B, C, M, N, K, lda, ldb, ldc, transA != 0, transB != 0, alpha, beta, bias, act,
                      (hipStream_t)stream);
}
