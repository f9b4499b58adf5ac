// Peak issue-rate check for v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void peak_bf16(float* out, int iters, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
__global__ __launch_bounds__(256) void peak_f32(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x, b = threadIdx.x * 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4); long long* clk; hipMalloc(&clk, 16); long long hclk[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024}) {
        const int iters = 4000;
        float ms;
        hipLaunchKernelGGL(peak_bf16<4>, dim3(wgs), dim3(256), 0, 0, out, 10, (long long*)nullptr);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(peak_bf16<4>, dim3(wgs), dim3(256), 0, 0, out, iters, clk); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
        printf("bf16 32x32x16 nacc=4 wgs=%d: %.3f ms  %.0f TFLOP/s   shader cycles %lld, 100MHz ticks %lld -> %.2f GHz; cycles per MFMA %.1f\n", wgs, ms, (double)wgs * 4 * iters * 16 * 32768.0 / ms * 1e-9,
               hclk[0], hclk[1], hclk[0] / (hclk[1] * 10.0), (double)hclk[0] / (iters * 16.0));
        hipLaunchKernelGGL(peak_bf16<2>, dim3(wgs), dim3(256), 0, 0, out, 10, (long long*)nullptr);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(peak_bf16<2>, dim3(wgs), dim3(256), 0, 0, out, iters, (long long*)nullptr); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("bf16 32x32x16 nacc=2 wgs=%d: %.3f ms  %.0f TFLOP/s\n", wgs, ms, (double)wgs * 4 * iters * 8 * 32768.0 / ms * 1e-9);
        hipLaunchKernelGGL(peak_f32, dim3(wgs), dim3(256), 0, 0, out, 10);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(peak_f32, dim3(wgs), dim3(256), 0, 0, out, iters / 4); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("f32  32x32x2        wgs=%d: %.3f ms  %.0f TFLOP/s\n", wgs, ms, (double)wgs * 4 * (iters / 4) * 16 * 4096.0 / ms * 1e-9);
    }
    return 0;
}
