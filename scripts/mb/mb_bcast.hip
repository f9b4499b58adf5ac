// Broadcast-read bandwidth: every one of 256 workgroups (512 threads, one per CU) reads the SAME region of `kb` KiB, `iters` times,
// with DEPTH 16-byte loads in flight per lane and almost no ALU work (xor fold).  What does "every CU reads the whole [B, K]
// operand of a decoder step from L2" cost?  Variants: plain loads / sc1 (L1-bypass) loads; with or without the region being
// re-written (sc1 write-through stores by its owner workgroups) between sweeps, which is what a persistent kernel does.
//   hipcc --offload-arch=gfx950 -O3 -o mb_bcast mb_bcast.hip && ./mb_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x27000);
}

template <int DEPTH, int SC1>
__global__ __launch_bounds__(512) void bcast_kernel(const unsigned* buf, unsigned bytes, int iters, unsigned* out) {
    const __amdgpu_buffer_rsrc_t r = rsrc_of(buf, bytes);
    const unsigned tid = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (unsigned o = tid * 16; o < bytes; o += DEPTH * 8192) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) v[u] = SC1 ? __builtin_amdgcn_raw_buffer_load_b128(r, o + u * 8192, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(r, o + u * 8192, 0, 0);
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) acc ^= v[u];
        }
        asm volatile("" ::: "memory");
        if (!SC1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }      // drop L1 so that the next sweep is L2-served again
    }
    out[blockIdx.x * 512 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int DEPTH, int SC1>
static void run(const unsigned* buf, unsigned kb, int nwg, unsigned* out) {
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((bcast_kernel<DEPTH, SC1>), dim3(nwg), dim3(512), 0, 0, buf, kb * 1024, 2, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((bcast_kernel<DEPTH, SC1>), dim3(nwg), dim3(512), 0, 0, buf, kb * 1024, iters, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    printf("depth=%2d sc1=%d wgs=%3d region=%4u KiB: %.2f us per sweep = %.1f GB/s per CU = %.1f B/clk@2.4GHz, aggregate %.1f TB/s\n", DEPTH, SC1, nwg, kb, us,
           kb * 1024.0 / us * 1e-3, kb * 1024.0 / us * 1e-3 / 2.4, nwg * kb * 1024.0 / us * 1e-6);
}

int main() {
    unsigned *buf, *out;
    (void)hipMalloc(&buf, 4 << 20); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemset(buf, 1, 4 << 20);
    for (unsigned kb : {128u, 256u, 512u, 1024u}) {
        run<8, 0>(buf, kb, 256, out);
        run<8, 1>(buf, kb, 256, out);
        run<16, 1>(buf, kb, 256, out);
        run<32, 1>(buf, kb, 256, out);
    }
    run<16, 1>(buf, 512, 64, out);
    run<16, 1>(buf, 512, 128, out);
    return 0;
}
