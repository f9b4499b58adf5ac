// Micro-benchmark v8 (v1 + two register sets, the global loads of tile k+2 issued one by one between MFMA groups of tile k): fp32 GEMM through the bf16 matrix cores by exact 3-way operand splitting (6 partial products).
//   x = h1 + h2 + h3 (+ <2^-24 |x|), each h a bf16;  a*b ~= a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1   (dropped terms <= 2^-23 |ab|)
// C[M,N] = A[M,K] * B[N,K]^T, both K-contiguous.  hipcc --offload-arch=gfx950 -O3 mb_gemm_split.hip -o mb_gemm_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROW_B = 80;                 // bytes per LDS row: 32 bf16 + 16 B pad
constexpr int PLANE_B = BM * ROW_B;       // 10240

__device__ __forceinline__ void split_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
    const float rx = x - __uint_as_float(ux & 0xffff0000u), ry = y - __uint_as_float(uy & 0xffff0000u);
    const unsigned vx = __float_as_uint(rx), vy = __float_as_uint(ry);
    const float sx = rx - __uint_as_float(vx & 0xffff0000u), sy = ry - __uint_as_float(vy & 0xffff0000u);
    p1 = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    p2 = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}

__device__ __forceinline__ void store_split(char* lds, int row, int k4, float4 v) {
    unsigned a1, a2, a3, b1, b2, b3;
    split_pair(v.x, v.y, a1, a2, a3);
    split_pair(v.z, v.w, b1, b2, b3);
    const int phys = (k4 >> 1) ^ ((row >> 4) & 3);
    char* p = lds + row * ROW_B + phys * 16 + (k4 & 1) * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(p + PLANE_B) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(p + 2 * PLANE_B) = make_uint2(a3, b3);
}

__global__ __launch_bounds__(256, 2) void gemm_split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                      int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) char smem[6 * PLANE_B];
    char* sa = smem; char* sb = smem + 3 * PLANE_B;
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

    float4 ra[4], rb[4], qa[4], qb[4];
    const int lrow = tid >> 3, lk4 = tid & 7;
    const float* ap = A + (long)(m0 + lrow) * K + lk4 * 4;
    const float* bp = B + (long)(n0 + lrow) * K + lk4 * 4;
    const int nk = K / BK;
#define LD1(R, P, IT, KB) R[IT] = *reinterpret_cast<const float4*>(P + (long)(IT) * 32 * K + (long)min(KB, nk - 1) * BK);
#define CB() asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) { LD1(ra, ap, it, 0) LD1(rb, bp, it, 0) }
#pragma unroll
    for (int it = 0; it < 4; ++it) { store_split(sa, it * 32 + lrow, lk4, ra[it]); store_split(sb, it * 32 + lrow, lk4, rb[it]); }
#pragma unroll
    for (int it = 0; it < 4; ++it) { LD1(ra, ap, it, 1) LD1(rb, bp, it, 1) }
    __syncthreads();

#define MM(PA, PB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =      \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0);
#define FRAGS(KS)                                                                                                \
    bf16x8 a[2][3], b[2][3];                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                              \
        const int sw = (i * 2 + (li >> 4)) & 3;                                                                  \
        const int off = (i * 32 + li) * ROW_B + ((((KS) * 2 + lq) ^ sw) * 16);                                   \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                       \
            a[i][pl] = *reinterpret_cast<const bf16x8*>(sa + pl * PLANE_B + wm * ROW_B + off);                   \
            b[i][pl] = *reinterpret_cast<const bf16x8*>(sb + pl * PLANE_B + wn * ROW_B + off);                   \
        }                                                                                                        \
    }
    // tile kb is in LDS; set S holds tile kb+1 (stored at the end of this iteration); set L receives tile kb+2, one load per MFMA group
#define ITER(S_A, S_B, L_A, L_B, KB)                                                                             \
    {                                                                                                            \
        { FRAGS(0) MM(2, 0) CB() LD1(L_A, ap, 0, (KB) + 2) CB() MM(0, 2) CB() LD1(L_B, bp, 0, (KB) + 2) CB() MM(1, 1) CB()      \
          LD1(L_A, ap, 1, (KB) + 2) CB() MM(1, 0) CB() LD1(L_B, bp, 1, (KB) + 2) CB() MM(0, 1) MM(0, 0) }                    \
        { FRAGS(1) MM(2, 0) CB() LD1(L_A, ap, 2, (KB) + 2) CB() MM(0, 2) CB() LD1(L_B, bp, 2, (KB) + 2) CB() MM(1, 1) CB()      \
          LD1(L_A, ap, 3, (KB) + 2) CB() MM(1, 0) CB() LD1(L_B, bp, 3, (KB) + 2) CB() MM(0, 1) MM(0, 0) }                    \
        __syncthreads();                                                                                         \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) { store_split(sa, it * 32 + lrow, lk4, S_A[it]); store_split(sb, it * 32 + lrow, lk4, S_B[it]); } \
        __syncthreads();                                                                                         \
    }
    for (int kb = 0; kb < nk; kb += 2) {
        ITER(ra, rb, qa, qb, kb)
        if (kb + 1 >= nk) break;
        ITER(qa, qb, ra, rb, kb + 1)
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
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, hC.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    dim3 grid((M / BM) * (N / BN));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_split, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_split, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0, sum_abs = 0;
    for (int t = 0; t < 256; ++t) {
        const int r = (t * 7919) % M, c = (t * 104729 + 13) % N;
        double ref = 0, mag = 0; float f32 = 0.f;
        for (int k = 0; k < K; ++k) {
            const double pa = hA[(size_t)r * K + k], pb = hB[(size_t)c * K + k];
            ref += pa * pb; mag += fabs(pa * pb); f32 = fmaf(hA[(size_t)r * K + k], hB[(size_t)c * K + k], f32);
        }
        worst = fmax(worst, fabs(hC[(size_t)r * N + c] - ref) / mag);
        worst32 = fmax(worst32, fabs((double)f32 - ref) / mag);
        sum_abs += mag;
    }
    printf("M=%d N=%d K=%d  %.3f ms  %.1f TFLOP/s-equivalent | max err / sum|ab|: split %.3e   fp32 fma chain %.3e\n", M, N, K, ms,
           2.0 * M * N * K / ms * 1e-9, worst, worst32);
    hipFree(dA); hipFree(dB); hipFree(dC);
}

int main() {
    run(4096, 4096, 4096);
    run(38400, 4096, 1536);
    run(38400, 512, 2560);
    run(3072, 4096, 1536);
    return 0;
}
