// Price of one all-to-all edge inside a persistent decoder kernel on MI355X (256 workgroups x 512 threads, one per CU):
//   every workgroup publishes PUB floats with write-through (sc1) stores, crosses a grid barrier, then reads the WHOLE exchange
//   buffer (256 x PUB floats) with sc1 loads (L1 bypass: no acquire fence) and checks every word.
// Barrier variants:  0 flat counter   1 two-level (8 groups by blockIdx % 8 -> global)   2 flag all-gather (no atomics: every
// workgroup stores its epoch, wave 0 of every workgroup polls all 256 slots with one 16-byte sc1 load per lane)
// Every spin is bounded (bail flag): a lost workgroup cannot hang the GPU.
//   hipcc --offload-arch=gfx950 -O3 -o mb_pbar mb_pbar.hip && ./mb_pbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
constexpr unsigned SPIN_MAX = 2000000u;

struct Bar {
    unsigned* cnt;      // [0] global counter, [32 * (1 + g)] group counters, [32 * 16 + wg * 2] flag slots (8 bytes apart)
    unsigned* bail;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x27000);
}

template <int VAR>
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned epoch) {       // epoch = 1, 2, ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // every wave drains its sc1 stores
    __syncthreads();
    bool ok = true;
    const unsigned nwg = gridDim.x;
    if (VAR == 2) {
        if (threadIdx.x == 0) __hip_atomic_store(b.cnt + 512 + blockIdx.x, epoch, RLX, AGENT);
        if (threadIdx.x < 64) {
            const __amdgpu_buffer_rsrc_t r = rsrc_of(b.cnt + 512, nwg * 4);
            unsigned spins = 0;
            for (;;) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 16);      // out of range reads 0 ...
                const bool mine = (threadIdx.x * 4 >= nwg) || (v.x >= epoch && v.y >= epoch && v.z >= epoch && v.w >= epoch);      // ... and is ignored
                if (__all(mine)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_MAX) { if (threadIdx.x == 0) *b.bail = 1; ok = false; break; }
            }
        }
    } else if (threadIdx.x == 0) {
        unsigned target;
        if (VAR == 0) {
            __hip_atomic_fetch_add(b.cnt, 1u, RLX, AGENT);
            target = epoch * nwg;
        } else {
            const unsigned ng = 8, g = blockIdx.x % ng, gsz = nwg / ng;
            const unsigned prev = __hip_atomic_fetch_add(b.cnt + 32 * (1 + g), 1u, RLX, AGENT);
            if (prev + 1 == epoch * gsz) __hip_atomic_fetch_add(b.cnt, 1u, RLX, AGENT);
            target = epoch * ng;
        }
        unsigned spins = 0;
        while (__hip_atomic_load(b.cnt, RLX, AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_MAX) { *b.bail = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok && (__hip_atomic_load(b.bail, RLX, AGENT) == 0);
}

// PUB floats published per workgroup and step (multiple of 4 * 512 or less than 512 * 4 with idle threads)
template <int VAR>
__global__ __launch_bounds__(512) void bar_kernel(Bar b, float* xbuf, int pub, int iters, int read_all, float* out, int stagger) {
    const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const unsigned slab = (unsigned)nwg * pub;                       // floats per parity buffer
    float bad = 0.f;
    for (int it = 0; it < iters; ++it) {
        float* buf = xbuf + (size_t)(it & 1) * slab;
        if (pub > 0) {
            const __amdgpu_buffer_rsrc_t w = rsrc_of(buf, slab * 4);
            for (int i = tid * 4; i < pub; i += 512 * 4) {
                const float base = (float)(it * 7 + wg);
                u32x4 v; v.x = __float_as_uint(base + i); v.y = __float_as_uint(base + i + 1); v.z = __float_as_uint(base + i + 2); v.w = __float_as_uint(base + i + 3);
                __builtin_amdgcn_raw_buffer_store_b128(v, w, (wg * pub + i) * 4, 0, 16);
            }
        }
        if (!grid_barrier<VAR>(b, (unsigned)(it + 1))) break;
        if (read_all == 2) {        // (read_all == 3: plain loads without any acquire - timing only, may read stale lines) one lane's agent-scope acquire (buffer_inv sc1), then plain loads
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
        }
        if (pub > 0 && read_all >= 10) {          // timing path: 16 sc1 loads in flight per lane, xor fold (no per-word check)
            const __amdgpu_buffer_rsrc_t r = rsrc_of(buf, slab * 4);
            u32x4 acc = {0, 0, 0, 0};
            for (unsigned o = tid * 16; o < slab * 4; o += 16 * 8192) {
                u32x4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, o + u * 8192, 0, 16);
#pragma unroll
                for (int u = 0; u < 16; ++u) acc ^= v[u];
            }
            bad += (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u;
        } else
        if (pub > 0 && read_all) {
            const __amdgpu_buffer_rsrc_t r = rsrc_of(buf, slab * 4);
            // 8 independent 16-byte loads in flight per lane (out-of-range offsets read 0 and are skipped in the check)
            // stagger: workgroup wg starts its sweep at chunk (wg * stagger) of the buffer so that the 256 readers do not hit the same
            // L2 channel at the same moment
            const unsigned nchunk = (slab + 16383) / 16384, rot = stagger ? (wg * (unsigned)stagger) % nchunk : 0;
            for (unsigned c = 0; c < nchunk; ++c) {
                const unsigned i0 = ((c + rot) % nchunk) * 16384 + tid * 4;
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (read_all >= 2) ? __builtin_amdgcn_raw_buffer_load_b128(r, (i0 + u * 2048) * 4, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(r, (i0 + u * 2048) * 4, 0, 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned i = i0 + u * 2048;
                    if (i >= slab) continue;
                    const int src = i / pub, j = i - src * pub;
                    const float base = (float)(it * 7 + src);
                    bad += (__uint_as_float(v[u].x) != base + j) + (__uint_as_float(v[u].y) != base + j + 1) + (__uint_as_float(v[u].z) != base + j + 2) +
                           (__uint_as_float(v[u].w) != base + j + 3);
                }
            }
        }
    }
    out[wg * 512 + tid] = bad;
}

__global__ void xcc_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x;
    }
}

template <int VAR>
static void run(const char* name, Bar b, float* xbuf, float* out, int nwg, int pub, int iters, int read_all, int stagger = 0) {
    hipMemset(b.cnt, 0, 4096 * 4); hipMemset(b.bail, 0, 4); hipMemset(out, 0, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bar_kernel<VAR>, dim3(nwg), dim3(512), 0, 0, b, xbuf, pub, iters, read_all, out, stagger);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned hb; hipMemcpy(&hb, b.bail, 4, hipMemcpyDeviceToHost);
    std::vector<float> ho(256 * 512); hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double bad = 0; for (float v : ho) bad += v;
    printf("%-10s stagger=%d wgs=%3d pub=%5d floats/wg read_all=%d iters=%4d: %.2f us per step, bail=%u, wrong words=%.0f  (%s)\n", name, stagger, nwg, pub, read_all,
           iters, ms * 1e3 / iters, hb, bad, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    Bar b; float *xbuf, *out; unsigned* xcc;
    hipMalloc(&b.cnt, 4096 * 4); hipMalloc(&b.bail, 4); hipMalloc(&xbuf, 2 * 256 * 8192 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&xcc, 1024);
    hipLaunchKernelGGL(xcc_kernel, dim3(256), dim3(64), 0, 0, xcc);
    unsigned hx[256]; hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
    int ok = 0; for (int i = 0; i < 256; ++i) ok += (hx[i] & 15) == (unsigned)(i % 8);
    printf("xcc id == blockIdx %% 8 for %d of 256 workgroups (first 16:", ok);
    for (int i = 0; i < 16; ++i) printf(" %u", hx[i] & 15);
    printf(")\n");
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("flat", b, xbuf, out, 256, 0, iters, 0);
        run<1>("2level", b, xbuf, out, 256, 0, iters, 0);
        run<2>("flags", b, xbuf, out, 256, 0, iters, 0);
    }
    for (int pub : {256, 392, 608, 1024}) {
        run<1>("2level", b, xbuf, out, 256, pub, iters, 1, 1);
        run<1>("2level", b, xbuf, out, 256, pub, iters, 10, 0);
        run<1>("2level", b, xbuf, out, 256, pub, iters, 0, 0);
    }
    return 0;
}
