// Grid-wide barrier latency on MI355X (256 workgroups x 512 threads, cooperative launch): what would a persistent
// decoder-loop kernel pay per phase boundary?  Spins are BOUNDED (bail-out flag) so that a lost workgroup cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>

#ifndef LEVELS
#define LEVELS 2
#endif
// counters: [0] = global, [32 * (1 + g)] = group g (own 128-byte line).  Two levels: the last arriver of a group of
// `gsz` workgroups forwards ONE increment to the global counter, everybody spins on the global counter.
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned it1, unsigned* bail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned nwg = gridDim.x;
#if defined(NOFENCE)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = it1 * nwg;
#elif LEVELS == 1
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);
        const unsigned target = it1 * nwg;
#else
        const unsigned ngroups = nwg >= 8 ? 8 : 1, g = blockIdx.x % ngroups, gsz = nwg / ngroups;
        const unsigned prev = __atomic_fetch_add(counter + 32 * (1 + g), 1u, __ATOMIC_ACQ_REL);
        if (prev + 1 == it1 * gsz) __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);
        const unsigned target = it1 * ngroups;
#endif
        unsigned spins = 0;
#ifdef RELAXED
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
#else
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) {
#endif
#ifdef SLEEP
            __builtin_amdgcn_s_sleep(SLEEP);
#endif
            if (++spins > 4000000u) { *bail = 1; ok = false; break; }
        }
#if defined(RELAXED) && !defined(NOFENCE)
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
#endif
    }
    __syncthreads();
    return ok;
}

// every barrier is followed by a read of data another workgroup wrote before the barrier (checks visibility)
__global__ __launch_bounds__(512) void bar_kernel(unsigned* counter, float* data, unsigned* bail, int iters, float* out) {
    const int nwg = gridDim.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#ifdef NOFENCE
        if (threadIdx.x < 64) __hip_atomic_store(&data[(size_t)(it & 1) * nwg * 64 + blockIdx.x * 64 + threadIdx.x], (float)(it + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        if (threadIdx.x < 64) data[(size_t)(it & 1) * nwg * 64 + blockIdx.x * 64 + threadIdx.x] = (float)(it + blockIdx.x);
#endif
        if (!grid_barrier(counter, (unsigned)(it + 1), bail)) return;
        const int other = (blockIdx.x + 97) % nwg;
        if (threadIdx.x < 64) {
#ifdef NOFENCE
            const float v = __hip_atomic_load(&data[(size_t)(it & 1) * nwg * 64 + other * 64 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
            const float v = __builtin_nontemporal_load(&data[(size_t)(it & 1) * nwg * 64 + other * 64 + threadIdx.x]);
#endif
            acc += v - (float)(it + other);                                      // stays 0 when the write was visible
        }
    }
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    unsigned *counter, *bail; float *data, *out;
    hipMalloc(&counter, 4096); hipMalloc(&bail, 4); hipMalloc(&data, 2 * 256 * 64 * 4); hipMalloc(&out, 256 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nwg : {64, 256}) {
        for (int iters : {200, 2000}) {
            hipMemset(counter, 0, 4096); hipMemset(bail, 0, 4); hipMemset(out, 0, 256 * 64 * 4);
            void* args[] = {&counter, &data, &bail, &iters, &out};
            hipEventRecord(e0, 0);
            hipError_t err = hipLaunchCooperativeKernel((const void*)bar_kernel, dim3(nwg), dim3(512), args, 0, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned hb; float ho[256 * 64]; hipMemcpy(&hb, bail, 4, hipMemcpyDeviceToHost); hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
            double bad = 0; for (int i = 0; i < nwg * 64; ++i) bad += ho[i] != 0.f;
            printf("wgs=%3d iters=%4d: launch %s, %.3f ms total, %.2f us per barrier, bail=%u, stale reads=%.0f\n", nwg, iters,
                   hipGetErrorString(err), ms, ms * 1e3 / iters, hb, bad);
        }
    }
    return 0;
}
