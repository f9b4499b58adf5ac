// Do two independent chains of short full-chip kernels on two HIP streams overlap on MI355X?  Each kernel: WGS workgroups x 512 threads that
// spin for ~DUR us (s_sleep), few registers, no LDS - so two of them fit on every CU side by side.  Prints us per step for chain A alone
// and for chains A and B together (same / lower priority for B), and with B's kernels narrower (fewer workgroups).
// hipcc --offload-arch=gfx950 -O3 mb_overlap.hip -o mb_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void spin(float* p, long ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f;
}
// same, but a workgroup as heavy as the decoder's step kernels: ~128 VGPRs x 8 waves and 48 KiB of LDS = half a CU
__global__ __launch_bounds__(512) void spin_heavy(float* p, long ticks) {
    extern __shared__ float sm[];
    float r[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) r[i] = p[16 + ((threadIdx.x + i) & 15)];
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 96; ++i) r[i] = r[i] * 1.0001f + r[(i + 7) % 96];
        __builtin_amdgcn_s_sleep(2);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += r[i];
    sm[threadIdx.x] = s;
    if (s == 123.456f) p[1] = sm[(threadIdx.x + 1) & 511];
}
int main() {
    float* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t a, b, c; hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
    hipStreamCreateWithPriority(&c, hipStreamNonBlocking, lo);
    hipEvent_t e0, e1, eb; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&eb);
    const int N = 600;
    const long ticks = 800;            // wall_clock64 runs at 100 MHz: 8 us
    auto run = [&](const char* name, hipStream_t sb, int wgs_a, int wgs_b, bool with_b) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, a);
            if (with_b) hipStreamWaitEvent(sb, e0, 0);
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin, dim3(wgs_a), dim3(512), 0, a, d, ticks);
                if (with_b) hipLaunchKernelGGL(spin, dim3(wgs_b), dim3(512), 0, sb, d + 8, ticks);
            }
            if (with_b) { hipEventRecord(eb, sb); hipStreamWaitEvent(a, eb, 0); }
            hipEventRecord(e1, a); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-58s %.2f us per step\n", name, ms * 1000 / N);
        }
    };
    auto runh = [&](const char* name, hipStream_t sb, int wgs_a, int wgs_b, bool with_b) {
        hipFuncSetAttribute((const void*)spin_heavy, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, a);
            if (with_b) hipStreamWaitEvent(sb, e0, 0);
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin_heavy, dim3(wgs_a), dim3(512), 48 * 1024, a, d, ticks);
                if (with_b) hipLaunchKernelGGL(spin_heavy, dim3(wgs_b), dim3(512), 48 * 1024, sb, d + 32, ticks);
            }
            if (with_b) { hipEventRecord(eb, sb); hipStreamWaitEvent(a, eb, 0); }
            hipEventRecord(e1, a); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-58s %.2f us per step\n", name, ms * 1000 / N);
        }
    };
    runh("heavy (128 VGPR, 48 KiB LDS) A alone: 256 wg", b, 256, 0, false);
    runh("heavy A alone: 416 wg", b, 416, 0, false);
    runh("heavy A alone: 512 wg", b, 512, 0, false);
    runh("heavy A + B same priority, 256 + 256 wg", b, 256, 256, true);
    runh("heavy A + B lower priority, 256 + 256 wg", c, 256, 256, true);
    runh("heavy A + B lower priority, 416 + 256 wg", c, 416, 256, true);
    run("A alone: 256 wg x 8 us", b, 256, 0, false);
    run("A alone: 512 wg x 8 us", b, 512, 0, false);
    run("A + B same priority, 256 + 256 wg", b, 256, 256, true);
    run("A + B lower priority, 256 + 256 wg", c, 256, 256, true);
    run("A + B same priority, 256 + 128 wg", b, 256, 128, true);
    run("A + B lower priority, 256 + 64 wg", c, 256, 64, true);
    run("A + B same priority, 416 + 256 wg", b, 416, 256, true);
    return 0;
}
