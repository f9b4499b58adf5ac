// Optional in-library sampling of the dominant step kernel with HIP events (used by bench.py for the roofline
// line).  Events are created by mtts_prof_begin() OUTSIDE the timed launches; recording is a no-op otherwise.
#include "common.h"
#include <vector>

namespace {
struct Prof {
    bool on = false;
    int stride = 1, used = 0;
    std::vector<hipEvent_t> ec, e0, e1;     // ec -> e0: empty bracket (cost of one event packet); e0 -> e1: the kernel
} g_prof;
}

bool prof_sample(int t, hipStream_t s, int phase) {
    // phase 0: before the kernel, 1: after.  Returns true when this step is being sampled.
    if (!g_prof.on || (t % g_prof.stride) != 0) return false;
    if (phase == 0) {
        if (g_prof.used >= (int)g_prof.e0.size()) return false;
        (void)hipEventRecord(g_prof.ec[g_prof.used], s);
        (void)hipEventRecord(g_prof.e0[g_prof.used], s);
        return true;
    }
    (void)hipEventRecord(g_prof.e1[g_prof.used], s);
    g_prof.used++;
    return true;
}

MTTS_API int mtts_prof_begin(int max_samples, int stride) {
    for (auto e : g_prof.ec) (void)hipEventDestroy(e);
    for (auto e : g_prof.e0) (void)hipEventDestroy(e);
    for (auto e : g_prof.e1) (void)hipEventDestroy(e);
    g_prof.ec.assign(max_samples, nullptr);
    g_prof.e0.assign(max_samples, nullptr);
    g_prof.e1.assign(max_samples, nullptr);
    for (int i = 0; i < max_samples; ++i) {
        MTTS_CHECK_HIP(hipEventCreate(&g_prof.ec[i]));
        MTTS_CHECK_HIP(hipEventCreate(&g_prof.e0[i]));
        MTTS_CHECK_HIP(hipEventCreate(&g_prof.e1[i]));
    }
    g_prof.stride = stride < 1 ? 1 : stride;
    g_prof.used = 0;
    g_prof.on = true;
    return 0;
}

static float g_prof_empty_ms = 0.f;
// Summed duration (ms) of the empty brackets recorded in front of every sample of the last mtts_prof_end():
// what an event pair costs with nothing between, to be subtracted from the kernel brackets.
MTTS_API float mtts_prof_empty_ms(void) { return g_prof_empty_ms; }

// Synchronises the recorded events; returns the number of samples and their summed duration (ms).
MTTS_API int mtts_prof_end(float* total_ms, int* count) {
    g_prof.on = false;
    float tot = 0.f;
    g_prof_empty_ms = 0.f;
    for (int i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        MTTS_CHECK_HIP(hipEventSynchronize(g_prof.e1[i]));
        MTTS_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.e0[i], g_prof.e1[i]));
        tot += ms;
        MTTS_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.ec[i], g_prof.e0[i]));
        g_prof_empty_ms += ms;
    }
    *total_ms = tot;
    *count = g_prof.used;
    return 0;
}

// A distinctively named no-op kernel: bench.py's PMC passes cut the dispatch stream into regions at these launches.
__global__ void mtts_marker_kernel(int tag, int* sink) { if (sink) *sink = tag; }

MTTS_API int mtts_prof_marker(int tag, void* stream) {
    hipLaunchKernelGGL(mtts_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tag, (int*)nullptr);
    MTTS_CHECK_LAUNCH("mtts_marker_kernel");
    return 0;
}

// Test hook (tests/test_gpu_persist.py): a FOREIGN resident kernel - `workgroups` workgroups of 64 threads that each hold `lds_bytes`
// of LDS and sleep for `ms` milliseconds (constant 100 MHz wall clock).  With more than ~4 KB free LDS short of a persistent decoder
// workgroup's request it keeps those CUs closed to the persistent kernels for as long as it runs, the way a long kernel of another
// stream (or RCCL's resident channels) does.
__global__ void mtts_occupy_kernel(long long ticks, int* sink) {
    extern __shared__ char occ_lds[];
    occ_lds[threadIdx.x] = (char)threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
    if (sink && threadIdx.x == 0) *sink = occ_lds[1];
}

MTTS_API int mtts_debug_occupy(int workgroups, int lds_bytes, float ms, void* stream) {
    MTTS_REQUIRE(workgroups > 0 && lds_bytes >= 64 && lds_bytes <= 160 * 1024 && ms >= 0.f && ms <= 10000.f, "mtts_debug_occupy: bad arguments");
    MTTS_CHECK_HIP(hipFuncSetAttribute((const void*)mtts_occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(mtts_occupy_kernel, dim3(workgroups), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, (long long)(ms * 1e5f), (int*)nullptr);
    MTTS_CHECK_LAUNCH("mtts_occupy_kernel");
    return 0;
}
