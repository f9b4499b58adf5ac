// Micro-benchmark v7 (v3 + register prefetch 3 stages ahead, loads / split / LDS traffic interleaved with the MFMAs by sched_group_barrier): split-bf16 fp32 GEMM, LDS double-buffered stages of SK k-values, one barrier per stage,
// operand split (VALU) of stage s+1 interleaved with the MFMAs of stage s inside every wave.
// C[M,N] = A[M,K] * B[N,K]^T.   hipcc --offload-arch=gfx950 -O3 -DSK=16 -DOCC=2 mb_gemm_split3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#ifndef SK
#define SK 16
#endif
#ifndef OCC
#define OCC 2
#endif
#ifndef NV
#define NV 4
#endif
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128;
constexpr int ROW_B = SK * 2;             // bytes per LDS row per plane (no padding; 16-B chunks XOR-swizzled by row)
constexpr int PLANE_B = BM * ROW_B;
constexpr int STAGE_B = 6 * PLANE_B;      // A planes 0..2, B planes 3..5
constexpr int NCH = SK / 8;               // 16-B chunks per row
constexpr int NLD = SK / 8;               // float4 per thread per operand per stage  (128 rows * SK / 4 / 256)
constexpr int TPR = SK / 4;               // threads per row

__device__ __forceinline__ void split_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
    const float rx = x - __uint_as_float(ux & 0xffff0000u), ry = y - __uint_as_float(uy & 0xffff0000u);
    const unsigned vx = __float_as_uint(rx), vy = __float_as_uint(ry);
    const float sx = rx - __uint_as_float(vx & 0xffff0000u), sy = ry - __uint_as_float(vy & 0xffff0000u);
    p1 = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    p2 = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}

// swizzle: physical 16-B chunk = chunk ^ f(row);  f spreads 16 consecutive rows over all banks
__device__ __forceinline__ int swz(int row) { return NCH == 2 ? ((row >> 3) & 1) : ((row >> 2) & 3); }

__device__ __forceinline__ void store_split(char* lds, int row, int k4, float4 v) {
    unsigned a1, a2, a3, b1, b2, b3;
    split_pair(v.x, v.y, a1, a2, a3);
    split_pair(v.z, v.w, b1, b2, b3);
    char* p = lds + row * ROW_B + (((k4 >> 1) ^ swz(row)) * 16) + (k4 & 1) * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(p + PLANE_B) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(p + 2 * PLANE_B) = make_uint2(a3, b3);
}

__global__ __launch_bounds__(256, OCC) void gemm_split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                        int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntx = N / BN;
    const int nt = ntx * (M / BM);
    int id = blockIdx.x;
    { const int q = nt / 8, r = nt % 8, xcd = id % 8, idx = id / 8; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int m0 = (id / ntx) * BM, n0 = (id % ntx) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, li = lane & 31, lq = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = tid / TPR, lk4 = tid % TPR;          // rows lrow + it * (256/TPR)
    constexpr int RSTEP = 256 / TPR;
    const float* ap = A + (long)(m0 + lrow) * K + lk4 * 4;
    const float* bp = B + (long)(n0 + lrow) * K + lk4 * 4;
    const int ns = K / SK;

#define LOADS(RA, RB, S)                                                                                         \
    _Pragma("unroll") for (int it = 0; it < NLD; ++it) {                                                         \
        RA[it] = *reinterpret_cast<const float4*>(ap + (long)it * RSTEP * K + (S) * SK);                         \
        RB[it] = *reinterpret_cast<const float4*>(bp + (long)it * RSTEP * K + (S) * SK);                         \
    }
#define STORES(RA, RB, BUF)                                                                                      \
    _Pragma("unroll") for (int it = 0; it < NLD; ++it) {                                                         \
        store_split(smem + (BUF) * STAGE_B, it * RSTEP + lrow, lk4, RA[it]);                                     \
        store_split(smem + (BUF) * STAGE_B + 3 * PLANE_B, it * RSTEP + lrow, lk4, RB[it]);                       \
    }
#define MM(PA, PB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =      \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0);
#define MFMAS(BUF)                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < SK / 16; ++ks) {                                                     \
        bf16x8 a[2][3], b[2][3];                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
            const int ra_ = wm + i * 32 + li, rb_ = wn + i * 32 + li;                                            \
            const char* pa_ = smem + (BUF) * STAGE_B + ra_ * ROW_B + (((ks * 2 + lq) ^ swz(ra_)) * 16);          \
            const char* pb_ = smem + (BUF) * STAGE_B + 3 * PLANE_B + rb_ * ROW_B + (((ks * 2 + lq) ^ swz(rb_)) * 16); \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                   \
                a[i][pl] = *reinterpret_cast<const bf16x8*>(pa_ + pl * PLANE_B);                                 \
                b[i][pl] = *reinterpret_cast<const bf16x8*>(pb_ + pl * PLANE_B);                                 \
            }                                                                                                    \
        }                                                                                                        \
        MM(2, 0) MM(0, 2) MM(1, 1) MM(1, 0) MM(0, 1) MM(0, 0)                                                    \
    }

    // register sets r0..r3: stage s+1 is split out of set (s+1)&3 while the loads of stage s+4 refill set s&3
    float4 r0a[NLD], r0b[NLD], r1a[NLD], r1b[NLD], r2a[NLD], r2b[NLD], r3a[NLD], r3b[NLD];
    LOADS(r0a, r0b, 0)
    STORES(r0a, r0b, 0)
    LOADS(r1a, r1b, min(1, ns - 1))
    LOADS(r2a, r2b, min(2, ns - 1))
    LOADS(r3a, r3b, min(3, ns - 1))
    LOADS(r0a, r0b, min(4, ns - 1))
    __syncthreads();
#define SCHED()                                                                                                  \
    _Pragma("unroll") for (int g_ = 0; g_ < 24 * (SK / 16); ++g_) {                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* 1 MFMA */                                        \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  /* NV VALU */                                       \
        __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);   /* 1 DS read/write */                               \
        if ((g_ % 6) == 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   /* 1 VMEM read */                \
    }
#define STAGE(BUF, SA, SB, LA, LB, S)                                                                            \
    {                                                                                                            \
        MFMAS(BUF)                                                                                               \
        STORES(SA, SB, (BUF) ^ 1)                                                                                \
        LOADS(LA, LB, min((S) + 5, ns - 1))                                                                      \
        SCHED()                                                                                                  \
        __syncthreads();                                                                                         \
    }
    // stage s reads LDS buffer s&1; set (s+1)&3 holds stage s+1; after its split that set is refilled with stage s+5
    for (int s = 0; s < ns; s += 4) {
        STAGE(0, r1a, r1b, r1a, r1b, s)
        if (s + 1 >= ns) break;
        STAGE(1, r2a, r2b, r2a, r2b, s + 1)
        if (s + 2 >= ns) break;
        STAGE(0, r3a, r3b, r3a, r3b, s + 2)
        if (s + 3 >= ns) break;
        STAGE(1, r0a, r0b, r0a, r0b, s + 3)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
                C[(long)row * N + col] = acc[i][j][r];
            }
        }
}

static void run(int M, int N, int K) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : hA) x = rnd() * 1.7f;
    for (auto& x : hB) x = rnd() * 0.3f + 0.01f;
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dB, hB.size() * 4); (void)hipMalloc(&dC, hC.size() * 4);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void*)gemm_split, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B);
    dim3 grid((M / BM) * (N / BN));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_split, grid, dim3(256), 2 * STAGE_B, 0, dA, dB, dC, M, N, K);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 10;
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_split, grid, dim3(256), 2 * STAGE_B, 0, dA, dB, dC, M, N, K);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0;
    for (int t = 0; t < 256; ++t) {
        const int r = (t * 7919) % M, c = (t * 104729 + 13) % N;
        double ref = 0, mag = 0; float f32 = 0.f;
        for (int k = 0; k < K; ++k) {
            const double pa = hA[(size_t)r * K + k], pb = hB[(size_t)c * K + k];
            ref += pa * pb; mag += fabs(pa * pb); f32 = fmaf(hA[(size_t)r * K + k], hB[(size_t)c * K + k], f32);
        }
        worst = fmax(worst, fabs(hC[(size_t)r * N + c] - ref) / mag);
        worst32 = fmax(worst32, fabs((double)f32 - ref) / mag);
    }
    printf("M=%d N=%d K=%d  %.3f ms  %.1f TF-eq | err/sum|ab|: split %.2e  fp32 chain %.2e\n", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9, worst, worst32);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
}

int main() {
    printf("SK=%d OCC=%d lds=%d\n", SK, OCC, 2 * STAGE_B);
    run(4096, 4096, 4096);
    run(38400, 4096, 1536);
    run(38400, 512, 2560);
    run(3072, 4096, 1536);
    return 0;
}
