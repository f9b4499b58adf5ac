// Micro-benchmark of the K loop of gemm_planes_kernel (csrc/gemm_planes.h): the product's generated instruction stream
// (csrc/gemm_planes_body.inc) under the harness's OWN macro definitions, so that stream knock-outs live here and not in the product:
//   -DKO_LD   no global loads in the loop          -DKO_LDS  no LDS reads / stores        -DKO_BAR  no block barrier
//   -DZERO_DATA all-zero operands (switching power of the matrix pipe)
//   -DKO_MFMA no MFMAs                             -DTILE_ORDER=0  row-major tile order (one tile row per 32 workgroups) instead of the product's column strips
// What is left when a stream is knocked out says what the step waits for.  Operands: random bf16 planes, M x N x K = 38400 x 4096 x 1536
// (the decoder's largest hoisted projection), no epilogue (accumulators folded into one store).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-D...] -I multilingual_text_to_speech_amd/csrc scripts/mb/mb_gemm_planes.hip -o scripts/mb/mb_gemm_planes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 128, SP_ROW_B = 64, SP_PLANE_B = BM * SP_ROW_B, PLN_STAGE_B = 3 * SP_PLANE_B, PLN_BLK_B = PLN_STAGE_B, PLN_OPERAND_B = 2 * PLN_STAGE_B;

#define PL_SB __builtin_amdgcn_sched_barrier(0);
#ifdef KO_MFMA
#define PL_MFMA(F, i, j, pa, pb)
#else
#define PL_MFMA(F, i, j, pa, pb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F##a[i][pa], F##b[j][pb], acc[i][j], 0, 0, 0);
#endif
#ifdef KO_LDS
#define PL_RDA(F, ks, i, pl, STG)
#define PL_RDB(F, ks, j, pl, STG)
#define PL_ST(O, j) asm volatile("" :: "v"(R##O[j]));
#else
#define PL_RDA(F, ks, i, pl, STG) F##a[i][pl] = *reinterpret_cast<const bf16x8*>(lds + ((STG) * PLN_STAGE_B + (pl) * SP_PLANE_B) + ra[ks][i]);
#define PL_RDB(F, ks, j, pl, STG) F##b[j][pl] = *reinterpret_cast<const bf16x8*>(lds + ((STG) * PLN_STAGE_B + (pl) * SP_PLANE_B) + rb[ks][j]);
#define PL_ST(O, j) *reinterpret_cast<u32x4*>(lds + (PL_NXT * PLN_STAGE_B + PL_OFF_##O + (j) * 4096) + wl) = R##O[j];
#endif
#define PL_OFF_A 0
#define PL_OFF_B PLN_OPERAND_B
#ifdef KO_LD
#define PL_LD(O, j)
#else
#define PL_LD(O, j) R##O[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc##O, wl, so##O + (j) * 4096, 0);
#endif
#define PL_NEXT(O) so##O = min(so##O + (unsigned)PLN_BLK_B, last##O);
#ifdef KO_BAR
#define PL_BARRIER
#else
#define PL_BARRIER __syncthreads();
#endif
#ifndef TILE_ORDER
#define TILE_ORDER 1      // the product's order: column strips of four tiles (gemm_tile_block); 0 = row-major
#endif

__global__ __launch_bounds__(256, 1) void k_planes(const char* __restrict__ Ap, const char* __restrict__ Bp, float* __restrict__ out, int M, int N, int nrec) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int ntx = N / BN, nty = M / BM, nt = ntx * nty;
    int id = blockIdx.x;
    int tile_m, tile_n;
    {
        const int q = nt / 8, r = nt % 8, xcd = id % 8, idx = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (TILE_ORDER == 0) { tile_m = id / ntx; tile_n = id % ntx; }
        else {      // column strips of four tiles, row-major inside a strip (gemm_tile_block of csrc/gemm.hip; ntx % 4 == 0 here)
            const int per_strip = 4 * nty, strip = id / per_strip, rem = id - strip * per_strip;
            tile_m = rem / 4; tile_n = strip * 4 + (rem & 3);
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, li = lane & 31, lq = lane >> 5;
    const int kb0 = 0, nk = nrec;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const long tsb = (long)nrec * PLN_BLK_B;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ap) + (long)tile_m * tsb, 0, (int)tsb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Bp) + (long)tile_n * tsb, 0, (int)tsb, 0x00020000);
    const unsigned wl = (unsigned)tid * 16;
    (void)m0; (void)n0;
    const unsigned lastA = (unsigned)(nk - 1) * PLN_BLK_B, lastB = lastA;
    unsigned soA = (unsigned)kb0 * PLN_BLK_B, soB = soA;
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
#pragma unroll
    for (int j = 0; j < 6; ++j) { RA[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, wl, soA + j * 4096, 0); RB[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, wl, soB + j * 4096, 0); }
    PL_NEXT(A) PL_NEXT(B)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        *reinterpret_cast<u32x4*>(lds + PL_OFF_A + j * 4096 + wl) = RA[j];
        *reinterpret_cast<u32x4*>(lds + PL_OFF_B + j * 4096 + wl) = RB[j];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) { RA[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, wl, soA + j * 4096, 0); RB[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, wl, soB + j * 4096, 0); }
    PL_NEXT(A) PL_NEXT(B)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            f0a[i][pl] = *reinterpret_cast<const bf16x8*>(lds + pl * SP_PLANE_B + ra[0][i]);
            f0b[i][pl] = *reinterpret_cast<const bf16x8*>(lds + pl * SP_PLANE_B + rb[0][i]);
            f1a[i][pl] = f0a[i][pl]; f1b[i][pl] = f0b[i][pl];
        }
    for (int n = (nk - kb0) >> 1; n > 0; --n) {
#define PL_CUR 0
#define PL_NXT 1
#include "gemm_planes_body.inc"
#undef PL_CUR
#undef PL_NXT
#define PL_CUR 1
#define PL_NXT 0
#include "gemm_planes_body.inc"
#undef PL_CUR
#undef PL_NXT
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
#ifdef KO_LD
    for (int j = 0; j < 6; ++j) s += __uint_as_float(RA[j].x + RB[j].y);
#endif
#ifdef KO_MFMA
    for (int i = 0; i < 2; ++i) for (int pl = 0; pl < 3; ++pl) s += (float)f0a[i][pl][0] + (float)f0b[i][pl][1] + (float)f1a[i][pl][2] + (float)f1b[i][pl][3];
#endif
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

int main() {
    const int M = 38400, N = 4096, K = 1536, nrec = K / 32;
    const size_t bA = (size_t)(M / BM) * nrec * PLN_BLK_B, bB = (size_t)(N / BN) * nrec * PLN_BLK_B;
    char *Ap, *Bp; float* out;
    hipMalloc(&Ap, bA); hipMalloc(&Bp, bB); hipMalloc(&out, (size_t)(M / BM) * (N / BN) * 256 * 4);
    {   // random bf16 values of moderate magnitude (the matrix pipe's power depends on the data)
        std::vector<unsigned short> h(bA / 2);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 3) & 0x8000u)); }
#ifdef ZERO_DATA      // all-zero operands: the same instruction stream at the matrix pipe's lowest switching power
        for (auto& v : h) v = 0;
#endif
        hipMemcpy(Ap, h.data(), bA, hipMemcpyHostToDevice);
        hipMemcpy(Bp, h.data(), bB, hipMemcpyHostToDevice);
    }
    hipFuncSetAttribute((const void*)k_planes, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const int grid = (M / BM) * (N / BN);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_planes, dim3(grid), dim3(256), 2 * PLN_OPERAND_B, 0, Ap, Bp, out, M, N, nrec);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k_planes, dim3(grid), dim3(256), 2 * PLN_OPERAND_B, 0, Ap, Bp, out, M, N, nrec);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double ksteps_per_cu = (double)grid / 256.0 * nrec;
    printf("%-28s %.3f ms  %.1f TF-eq  %.3f us per K step and CU (%d tiles, %d steps each)\n",
#if defined(KO_LD) || defined(KO_LDS) || defined(KO_BAR) || defined(KO_MFMA) || !TILE_ORDER || defined(ZERO_DATA)
           "variant"
#else
           "full stream"
#endif
           , ms, 2.0 * M * N * K / ms * 1e-9, ms * 1e3 / ksteps_per_cu, grid, nrec);
    return 0;
}
