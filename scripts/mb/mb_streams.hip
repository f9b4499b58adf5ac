// Micro-benchmark: cost of a cross-stream dependency per step (event record + stream wait) vs same-stream launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(10); }
int main() {
    float* d; hipMalloc(&d, 4);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 600;
    hipEvent_t ev[2 * N]; for (int i = 0; i < 2 * N; ++i) hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    for (int work = 0; work <= 40; work += 40) {
        // (1) same stream: 3 dependent kernels per step
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, a);
            for (int i = 0; i < N; ++i) { for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, a, d, work); }
            hipEventRecord(e1, a); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("work=%d same-stream 3 kernels/step: %.2f us/step\n", work, ms * 1000 / N);
        }
        // (2) per step: A: k, k, k ; B: one kernel that depends on A's first kernel and that A's next step depends on
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, a);
            for (int i = 0; i < N; ++i) {
                if (i > 0) hipStreamWaitEvent(a, ev[2 * i - 1], 0);
                hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, a, d, work);
                hipEventRecord(ev[2 * i], a);
                hipStreamWaitEvent(b, ev[2 * i], 0);
                hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, b, d, work);
                hipEventRecord(ev[2 * i + 1], b);
                hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, a, d, work);
                hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, a, d, work);
            }
            hipEventRecord(e1, a); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("work=%d two-stream (A:3 + B:1 overlapped, 2 event edges/step): %.2f us/step\n", work, ms * 1000 / N);
        }
    }
    return 0;
}
