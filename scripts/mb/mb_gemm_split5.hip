// Micro-benchmark v3: split-bf16 fp32 GEMM, LDS double-buffered stages of SK k-values, one barrier per stage,
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
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef TBM
#define TBM 256
#endif
#ifndef TBN
#define TBN 128
#endif
constexpr int BM = TBM, BN = TBN;
constexpr int NTHR = (BM / 64) * (BN / 64) * 64;
constexpr int ROW_B = SK * 2;             // bytes per LDS row per plane (no padding; 16-B chunks XOR-swizzled by row)
constexpr int PLANE_A = BM * ROW_B, PLANE_Bn = BN * ROW_B;
constexpr int STAGE_B = 3 * (PLANE_A + PLANE_Bn);      // A planes 0..2, then B planes
constexpr int NCH = SK / 8;               // 16-B chunks per row
constexpr int NLDA = BM * SK / 4 / NTHR, NLDB = BN * SK / 4 / NTHR;
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

template <int PLANE_B>
__device__ __forceinline__ void store_split(char* lds, int row, int k4, float4 v) {
    unsigned a1, a2, a3, b1, b2, b3;
    split_pair(v.x, v.y, a1, a2, a3);
    split_pair(v.z, v.w, b1, b2, b3);
    char* p = lds + row * ROW_B + (((k4 >> 1) ^ swz(row)) * 16) + (k4 & 1) * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(p + PLANE_B) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(p + 2 * PLANE_B) = make_uint2(a3, b3);
}

__global__ __launch_bounds__(NTHR, OCC) void gemm_split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                        int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntx = N / BN;
    const int nt = ntx * (M / BM);
    int id = blockIdx.x;
    { const int q = nt / 8, r = nt % 8, xcd = id % 8, idx = id / 8; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int m0 = (id / ntx) * BM, n0 = (id % ntx) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / (BN / 64)) * 64, wn = (wave % (BN / 64)) * 64, li = lane & 31, lq = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = tid / TPR, lk4 = tid % TPR;          // rows lrow + it * (256/TPR)
    constexpr int RSTEP = NTHR / TPR;
    const float* ap = A + (long)(m0 + lrow) * K + lk4 * 4;
    const float* bp = B + (long)(n0 + lrow) * K + lk4 * 4;
    float4 pa[NLDA], pb[NLDB], qa[NLDA], qb[NLDB];
    const int ns = K / SK;

#define LOADS(RA, RB, S)                                                                                         \
    _Pragma("unroll") for (int it = 0; it < NLDA; ++it) RA[it] = *reinterpret_cast<const float4*>(ap + (long)it * RSTEP * K + (S) * SK); \
    _Pragma("unroll") for (int it = 0; it < NLDB; ++it) RB[it] = *reinterpret_cast<const float4*>(bp + (long)it * RSTEP * K + (S) * SK);
#define STORES(RA, RB, BUF)                                                                                      \
    _Pragma("unroll") for (int it = 0; it < NLDA; ++it) store_split<PLANE_A>(smem + (BUF) * STAGE_B, it * RSTEP + lrow, lk4, RA[it]); \
    _Pragma("unroll") for (int it = 0; it < NLDB; ++it) store_split<PLANE_Bn>(smem + (BUF) * STAGE_B + 3 * PLANE_A, it * RSTEP + lrow, lk4, RB[it]);
#define MM(PA, PB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =      \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0);
#define MFMAS(BUF)                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < SK / 16; ++ks) {                                                     \
        bf16x8 a[2][3], b[2][3];                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
            const int ra_ = wm + i * 32 + li, rb_ = wn + i * 32 + li;                                            \
            const char* pa_ = smem + (BUF) * STAGE_B + ra_ * ROW_B + (((ks * 2 + lq) ^ swz(ra_)) * 16);          \
            const char* pb_ = smem + (BUF) * STAGE_B + 3 * PLANE_A + rb_ * ROW_B + (((ks * 2 + lq) ^ swz(rb_)) * 16); \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                   \
                a[i][pl] = *reinterpret_cast<const bf16x8*>(pa_ + pl * PLANE_A);                                 \
                b[i][pl] = *reinterpret_cast<const bf16x8*>(pb_ + pl * PLANE_Bn);                                \
            }                                                                                                    \
        }                                                                                                        \
        MM(2, 0) MM(0, 2) MM(1, 1) MM(1, 0) MM(0, 1) MM(0, 0)                                                    \
    }

    LOADS(pa, pb, 0)
    STORES(pa, pb, 0)
    if (ns > 1) { LOADS(pa, pb, 1) }
    if (ns > 2) { LOADS(qa, qb, 2) }
    __syncthreads();
    // stage s lives in LDS buffer s&1; registers p hold stage s+1 (odd s: q), registers q hold stage s+2
    for (int s = 0; s < ns; s += 2) {
        MFMAS(0)
        if (s + 1 < ns) { STORES(pa, pb, 1) }
        if (s + 3 < ns) { LOADS(pa, pb, s + 3) }
        __syncthreads();
        if (s + 1 >= ns) break;
        MFMAS(1)
        if (s + 2 < ns) { STORES(qa, qb, 0) }
        if (s + 4 < ns) { LOADS(qa, qb, s + 4) }
        __syncthreads();
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
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_split, grid, dim3(NTHR), 2 * STAGE_B, 0, dA, dB, dC, M, N, K);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 10;
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_split, grid, dim3(NTHR), 2 * STAGE_B, 0, dA, dB, dC, M, N, K);
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
    printf("tile %dx%d SK=%d OCC=%d lds=%d threads=%d\n", BM, BN, SK, OCC, 2 * STAGE_B, NTHR);
    run(4096, 4096, 4096);
    run(38400, 4096, 1536);
    run(38400, 512, 2560);
    run(3072, 4096, 1536);
    return 0;
}
