// Dependent-kernel chain: plain stream launches vs one captured hipGraph (what does graph replay save per boundary?)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void work(float* x, int n_iter) {
    float v = x[threadIdx.x + blockIdx.x * blockDim.x];
    for (int i = 0; i < n_iter; ++i) v = v * 1.0001f + 0.5f;
    x[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
int main() {
    float* x; hipMalloc(&x, 256 * 512 * 4); hipMemset(x, 0, 256 * 512 * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 1800;
    for (int iters : {1, 400, 2000}) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(work, dim3(256), dim3(512), 0, s, x, iters);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(256), dim3(512), 0, s, x, iters);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms_plain; hipEventElapsedTime(&ms_plain, e0, e1);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(256), dim3(512), 0, s, x, iters);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms_graph; hipEventElapsedTime(&ms_graph, e0, e1);
        printf("iters=%4d  plain %.2f us/kernel   graph %.2f us/kernel\n", iters, ms_plain * 1e3 / N, ms_graph * 1e3 / N);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
