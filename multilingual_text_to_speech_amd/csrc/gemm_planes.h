// gemm_planes_kernel: the GEMM core on PRE-SPLIT operands (round 5).  Included by gemm.hip (shares its tile constants and epilogues).
//
// Why: gemm_pipe_kernel splits every fp32 element into three bf16 planes while its tile passes through the registers - 176 vector
// instructions per operand tile and K block, placed in the shadow of 48 MFMAs.  With one wave per SIMD that is 3.7 vector instructions
// behind every 32-cycle MFMA: the stream is vector-ISSUE bound (matrix pipe 43-52 % busy inside the train step,
// profiles/r05_pmc_mfma.txt), and every one of the M/128 (N/128) workgroups that share an operand panel repeats the same split.
// Here the split happens ONCE per operand, in a bandwidth-bound pack pass (read 4 B, write 6 B per element), and the K loop of the
// GEMM issues only MFMAs, fragment reads, 16-byte LDS stores and 16-byte buffer loads.  The same pass transposes operands whose
// contraction runs over their rows (weight-gradient GEMMs), so that ONE kernel - both operands K-contiguous - serves all four
// (transA, transB) forms.  bf16 mode (mtts_set_precision(1), BASELINE configs[3]): one RNE-rounded plane per operand; the three
// "planes" of a record are then three consecutive 32-wide K blocks and the terms (0,0) (1,1) (2,2): 96 k per step, one MFMA term -
// replaces gemm_split_kernel<.., 1> (matrix pipe 10 % busy, profiles/r05_pmc_mfma.txt).
//
// Packed operand ("planes"), rows R, contraction length K: TILE-MAJOR blocks of 24 KiB, block (t, rec) = rows [128 t, 128 t + 128) of
// record rec, stored as the exact LDS image of one operand stage - [3 planes][128 rows][64 B], 16-byte chunk c of row r at chunk
// c ^ ((r >> 2) & 3) - so that a workgroup's tile load is ONE contiguous 24 KiB copy (every wave instruction reads 1 KiB of whole
// cache lines) and global offset == LDS offset:
//   fp32 mode : record = one 32-wide K block, plane pl = pl-th bf16 of the exact 3-way split, nrec = ceil(K / 32)
//   bf16 mode : record = three 32-wide K blocks ("plane" pl = K block 3 rec + pl), one RNE-rounded bf16 each, nrec = ceil(K / 96)
// k beyond K and rows beyond R (the last tile's padding) are written as zeros by the pack pass.
// (The first layout - 192-byte records row by row - made every wave instruction touch six row segments of 1.5 cache lines: the K loop
// without MFMAs took 1.10 us per step against 0.98 us for the MFMAs alone, scripts/mb/mb_gemm_planes.hip, profiles/r05_gemm_core.txt.)
//
// Arithmetic of the fp32 mode = gemm_pipe_kernel's: the same exact 3-way split (truncation, exact residuals), the same six terms in
// the same order, the same K order and split-K partition, the same epilogue: BIT-IDENTICAL results (tests/test_gpu_gemm_pipe.py).
#pragma once

constexpr int PLN_STAGE_B = 3 * SP_PLANE_B;          // one operand, one stage = one packed block: 24 KiB (as gemm_pipe_kernel)
constexpr int PLN_BLK_B = PLN_STAGE_B;
constexpr int PLN_OPERAND_B = 2 * PLN_STAGE_B;       // two stages; B behind A: 96 KiB in all

// ---- pack passes -----------------------------------------------------------------------------------------------------------------
// 8 consecutive k of one row -> NPL quanta of 16 bytes
// (output row r, 32-wide K block kb, 8-k chunk kq) -> its quantum / quanta inside the packed operand
template <bool BF16>
__device__ __forceinline__ void pln_store8(char* dst, int nrec, long r, int kb, int kq, const float (&f)[8]) {
    const int rin = (int)(r & 127);
    char* q = dst + ((r >> 7) * nrec + (BF16 ? kb / 3 : kb)) * (long)PLN_BLK_B + rin * SP_ROW_B + ((kq ^ ((rin >> 2) & 3)) * 16);
    if (BF16) {
        *reinterpret_cast<uint4*>(q + (kb % 3) * SP_PLANE_B) =
            make_uint4(rne_pair(f[0], f[1]), rne_pair(f[2], f[3]), rne_pair(f[4], f[5]), rne_pair(f[6], f[7]));
    } else {
        unsigned p1[4], p2[4], p3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split_pair(f[2 * e], f[2 * e + 1], p1[e], p2[e], p3[e]);
        *reinterpret_cast<uint4*>(q) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4*>(q + SP_PLANE_B) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        *reinterpret_cast<uint4*>(q + 2 * SP_PLANE_B) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
    }
}

// K-contiguous source: element (r, k) at src[r * ld + k].  One thread = one (row, 8 k) quantum; index order (kq, row inside the tile,
// K block, tile): 4 threads read 128 contiguous bytes of a row, a workgroup's 64 rows write 4 KiB contiguous per plane.
template <bool BF16>
__global__ __launch_bounds__(256) void pln_pack_plain_kernel(const float* __restrict__ src, long ld, int R, int K, char* __restrict__ dst, int nrec, int vec) {
    const int nkb = BF16 ? 3 * nrec : nrec;          // 32-wide K blocks written (bf16 mode: padded to whole records)
    const long total = (long)((R + 127) >> 7) * nkb * 512;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int kq = (int)(idx & 3), rin = (int)((idx >> 2) & 127);
        const long rest = idx >> 9;
        const int kb = (int)(rest % nkb);
        const long r = (rest / nkb) * 128 + rin;
        const int k0 = kb * 32 + kq * 8;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        if (r < R) {
            const float* s = src + r * ld + k0;
            if (vec && k0 + 8 <= K) {
                const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
                f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k0 + e < K) f[e] = s[e];
            }
        }
        pln_store8<BF16>(dst, nrec, r, kb, kq, f);
    }
}

// Transposed source: element (r, k) at src[k * ld + r] (the contraction runs over the source's ROWS: weight-gradient GEMMs).
// Workgroup = 64 output rows x one 32-wide K block through an LDS tile [32][65].
template <bool BF16>
__global__ __launch_bounds__(256) void pln_pack_trans_kernel(const float* __restrict__ src, long ld, int R, int K, char* __restrict__ dst, int nrec, int vec) {
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 64, kb = blockIdx.y, k0 = kb * 32;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = tid + 256 * it, kk = q >> 4, c4 = (q & 15) * 4;          // 16 threads x float4 = 64 output rows of one k
        const int k = k0 + kk, r = r0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            const float* s = src + (long)k * ld + r;
            if (vec && r + 4 <= R) v = *reinterpret_cast<const float4*>(s);
            else { if (r < R) v.x = s[0]; if (r + 1 < R) v.y = s[1]; if (r + 2 < R) v.z = s[2]; if (r + 3 < R) v.w = s[3]; }
        }
        tile[kk][c4] = v.x; tile[kk][c4 + 1] = v.y; tile[kk][c4 + 2] = v.z; tile[kk][c4 + 3] = v.w;
    }
    __syncthreads();
    const int rr = tid >> 2, kq = tid & 3, r = r0 + rr;      // rows beyond R (padding of the last tile) were loaded as zeros and are written
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = tile[kq * 8 + e][rr];
    pln_store8<BF16>(dst, nrec, r, kb, kq, f);
}

// ---- the GEMM on planes ----------------------------------------------------------------------------------------------------------
#define PL_SB __builtin_amdgcn_sched_barrier(0);
#define PL_MFMA(F, i, j, pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F##a[i][pa], F##b[j][pb], acc[i][j], 0, 0, 0);
#define PL_RDA(F, ks, i, pl, STG) F##a[i][pl] = *reinterpret_cast<const bf16x8*>(lds + ((STG) * PLN_STAGE_B + (pl) * SP_PLANE_B) + ra[ks][i]);
#define PL_RDB(F, ks, j, pl, STG) F##b[j][pl] = *reinterpret_cast<const bf16x8*>(lds + ((STG) * PLN_STAGE_B + (pl) * SP_PLANE_B) + rb[ks][j]);
#define PL_OFF_A 0
#define PL_OFF_B PLN_OPERAND_B
#define PL_ST(O, j) *reinterpret_cast<u32x4*>(lds + (PL_NXT * PLN_STAGE_B + PL_OFF_##O + (j) * 4096) + wl) = R##O[j];
#define PL_LD(O, j) R##O[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc##O, wl, so##O + (j) * 4096, 0);
#define PL_NEXT(O) so##O = min(so##O + (unsigned)PLN_BLK_B, last##O);
#define PL_BARRIER __syncthreads();

template <bool BF16>
__global__ __launch_bounds__(256, 1) void gemm_planes_kernel(GemmArgs p, const char* __restrict__ Ap, const char* __restrict__ Bp, int nrec, float* g_ws) {
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int li = lane & 31, lq = lane >> 5;

    // records of this workgroup (split-K over blockIdx.y: the partition of gemm_pipe_kernel)
    const int per_split = (nrec + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kb0 = blockIdx.y * per_split;
    const int nk = min(nrec, kb0 + per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kb0 < nk) {
        // ---- global side: the tile's blocks of an operand are contiguous (block = LDS image of a stage): quantum j of a thread sits
        // at byte (tid + 256 j) * 16 of the block AND of the stage; the record rides in the scalar offset
        const long tsb = (long)nrec * PLN_BLK_B;                       // bytes of one row tile (all its records)
        const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ap) + (long)tile_m * tsb, 0, (int)tsb, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Bp) + (long)tile_n * tsb, 0, (int)tsb, 0x00020000);
        const unsigned wl = (unsigned)tid * 16;
        const unsigned lastA = (unsigned)(nk - 1) * PLN_BLK_B, lastB = lastA;
        unsigned soA = (unsigned)kb0 * PLN_BLK_B, soB = soA;

        // ---- LDS side: fragment read addresses (as gemm_pipe_kernel)
        unsigned ra[2][2], rb[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rowa = wm + i * 32 + li, rowb = wn + i * 32 + li;
                ra[ks][i] = rowa * SP_ROW_B + (((2 * ks + lq) ^ ((rowa >> 2) & 3)) * 16);
                rb[ks][i] = rowb * SP_ROW_B + (((2 * ks + lq) ^ ((rowb >> 2) & 3)) * 16) + PLN_OPERAND_B;
            }

        u32x4 RA[6], RB[6];
        bf16x8 f0a[2][3], f0b[2][3], f1a[2][3], f1b[2][3];

        // prologue: record kb0 -> stage 0, record kb0 + 1 into the registers, every ks = 0 fragment of stage 0
#define PL_NXT 0
#pragma unroll
        for (int j = 0; j < 6; ++j) { PL_LD(A, j) PL_LD(B, j) }
        PL_NEXT(A) PL_NEXT(B)
#pragma unroll
        for (int j = 0; j < 6; ++j) { PL_ST(A, j) PL_ST(B, j) }
#pragma unroll
        for (int j = 0; j < 6; ++j) { PL_LD(A, j) PL_LD(B, j) }
        PL_NEXT(A) PL_NEXT(B)
#undef PL_NXT
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { PL_RDA(f0, 0, i, pl, 0) PL_RDB(f0, 0, i, pl, 0) }

        // records in pairs (the stage is a compile-time constant inside each copy of the stream), an odd last record after the loop
        for (int n = (nk - kb0) >> 1; n > 0; --n) {
#define PL_CUR 0
#define PL_NXT 1
            if constexpr (BF16) {
#include "gemm_planes_body_bf16.inc"
            } else {
#include "gemm_planes_body.inc"
            }
#undef PL_CUR
#undef PL_NXT
#define PL_CUR 1
#define PL_NXT 0
            if constexpr (BF16) {
#include "gemm_planes_body_bf16.inc"
            } else {
#include "gemm_planes_body.inc"
            }
#undef PL_CUR
#undef PL_NXT
        }
        if ((nk - kb0) & 1) {
#define PL_CUR 0
#define PL_NXT 1
            if constexpr (BF16) {
#include "gemm_planes_body_bf16.inc"
            } else {
#include "gemm_planes_body.inc"
            }
#undef PL_CUR
#undef PL_NXT
        }
    }
    pipe_epilogue(p, acc, smem, g_ws, p.C, p.bias, 0, m0 + wm, n0 + wn, lane, wave);
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
static long g_planes_launches = 0;
MTTS_API long mtts_gemm_planes_count(void) { return __atomic_load_n(&g_planes_launches, __ATOMIC_RELAXED); }

// pack buffer of (device, stream): grow-only, owned by the library (every stream that runs GEMMs - caller, side, weight-gradient -
// gets its own: their GEMMs run concurrently)
char* planes_buffer(hipStream_t s, size_t bytes);      // common.cpp; NULL when the allocation fails

// does this GEMM go through the planes core?  Plain GEMMs only (no convolution forms, no batches), big enough that the pack passes
// (two launches, (M + N) K (4 + 6) bytes of traffic) cost less than the vector issue they remove from the K loop.
// MTTS_GEMM_PLANES: 0 = never, 1 (default) = the bf16 path only, 2 = fp32 GEMMs as well.  Measured (profiles/r05_gemm_core.txt): in
// bf16 mode the pre-split core is 1.3-1.9x faster per call, pack passes included; in fp32 mode it is NOT faster than
// gemm_pipe_kernel although its K loop has no vector arithmetic and no exposed LDS wait - the six-term product runs into the chip's
// power limit, not into instruction issue - so the fp32 GEMMs keep the core that needs no pack pass.
static bool planes_wanted(const GemmArgs& p, bool bf16) {
    static const int on = [] { const char* e = getenv("MTTS_GEMM_PLANES"); return e ? atoi(e) : 1; }();
    if (!on || (!bf16 && on < 2)) return false;
    if (p.shift_mode != 0 || p.taps != 1 || p.batch != 1 || p.zt != 1) return false;
    static const double min_gflop_f32 = [] { const char* e = getenv("MTTS_PLANES_MIN_GFLOP"); return e ? atof(e) : 12.0; }();
    static const double min_gflop_bf16 = [] { const char* e = getenv("MTTS_PLANES_MIN_GFLOP_BF16"); return e ? atof(e) : 4.0; }();
    const double gflop = 2.0 * p.M * (double)p.N * p.K * 1e-9;
    // short reductions: the GEMM is dominated by its per-tile overhead either way; at K = 256 the pre-split core is 6 % ahead (0.342 vs 0.365 ms)
    static const int min_k_bf16 = [] { const char* e = getenv("MTTS_PLANES_MIN_K_BF16"); return e ? atoi(e) : 256; }();
    if (gflop < (bf16 ? min_gflop_bf16 : min_gflop_f32) || p.K < (bf16 && min_gflop_bf16 > 0 ? min_k_bf16 : 64)) return false;
    const int nrec = bf16 ? cdiv(p.K, 96) : cdiv(p.K, 32);
    const double bytes = ((double)cdiv(p.M, BM) + cdiv(p.N, BN)) * nrec * PLN_BLK_B;
    if (bytes > 1.6e9) return false;      // descriptor extents and buffer size
    return true;
}

template <bool BF16>
static int planes_pack(const float* src, long ld, bool trans, int R, int K, char* dst, int nrec, hipStream_t s) {
    const int vec = ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) ? 1 : 0;
    if (!trans) {
        const long total = (long)cdiv(R, 128) * (BF16 ? 3 * nrec : nrec) * 512;
        long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL((pln_pack_plain_kernel<BF16>), dim3((unsigned)blocks), dim3(256), 0, s, src, ld, R, K, dst, nrec, vec);
    } else {
        hipLaunchKernelGGL((pln_pack_trans_kernel<BF16>), dim3(2 * cdiv(R, 128), BF16 ? 3 * nrec : nrec), dim3(256), 0, s, src, ld, R, K, dst, nrec, vec);
    }
    MTTS_CHECK_LAUNCH("pln_pack_kernel");
    return 0;
}

// A(m, k): transA ? A[k * lda + m] : A[m * lda + k];  B(n, k): transB ? B[k * ldb + n] : B[n * ldb + k]  (GemmArgs convention)
template <bool BF16>
static int planes_gemm(const GemmArgs& p, dim3 grid, float* ws, hipStream_t s, bool* taken) {
    *taken = false;
    const int nrec = BF16 ? cdiv(p.K, 96) : cdiv(p.K, 32);
    const size_t bytesA = (size_t)cdiv(p.M, BM) * nrec * PLN_BLK_B, bytesB = (size_t)cdiv(p.N, BN) * nrec * PLN_BLK_B;
    char* buf = planes_buffer(s, bytesA + bytesB + 512);
    if (!buf) return 0;                       // no buffer: the caller falls back to the split-on-the-fly cores
    char* Ap = buf, *Bp = buf + ((bytesA + 255) & ~(size_t)255);
    MTTS_TRY(planes_pack<BF16>(p.A, p.lda, p.transA != 0, p.M, p.K, Ap, nrec, s));
    MTTS_TRY(planes_pack<BF16>(p.B, p.ldb, p.transB != 0, p.N, p.K, Bp, nrec, s));
    static bool attr_done_dev[64] = {false};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    if (!attr_done_dev[dev_ & 63]) {
        MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_planes_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_planes_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_done_dev[dev_ & 63] = true;
    }
    hipLaunchKernelGGL((gemm_planes_kernel<BF16>), grid, dim3(256), 2 * PLN_OPERAND_B, s, p, (const char*)Ap, (const char*)Bp, nrec, ws);
    MTTS_CHECK_LAUNCH("gemm_planes_kernel");
    *taken = true;
    __atomic_fetch_add(&g_planes_launches, 1L, __ATOMIC_RELAXED);
    return 0;
}
