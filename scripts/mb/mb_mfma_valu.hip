// Do plain VALU instructions hide behind bf16 MFMAs on gfx950?  One wave per SIMD (256 workgroups x 256 threads) or two (512 wgs),
// NV independent VALU ops (v_and / v_sub_f32 / v_perm, the operand-split mix) issued after every v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    float v[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.5f + i; u[i] = threadIdx.x * 977u + i; }
    const unsigned mask = 0xffff0000u, sel = 0x07060302u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int r = (m * NV + j) & 7;
                if ((j % 3) == 0) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[r]) : "v"(u[(r + 3) & 7]), "v"(mask));
                else if ((j % 3) == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[r]) : "v"(v[(r + 3) & 7]), "v"(v[(r + 5) & 7]));
                else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[r]) : "v"(u[(r + 2) & 7]), "v"(u[(r + 5) & 7]), "v"(sel));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(float* out, int wgs) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(wgs), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0, 0); hipLaunchKernelGGL(k<NV>, dim3(wgs), dim3(256), 0, 0, out, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)wgs * 4 * iters * 16;
    printf("NV=%d wgs=%4d: %.3f ms  %.0f bf16-TFLOP/s  (%.1f ns per MFMA per wave-slot)\n", NV, wgs, ms, mfma * 32768.0 / ms * 1e-9, ms * 1e6 / (iters * 16.0 * (wgs / 256)));
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    for (int wgs : {256, 512}) { run<0>(out, wgs); run<2>(out, wgs); run<4>(out, wgs); run<6>(out, wgs); run<8>(out, wgs); }
    return 0;
}
