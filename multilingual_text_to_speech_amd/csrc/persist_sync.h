// Synchronisation primitives of the persistent kernels (persist.hip: decoder recurrences; pbwd.hip: decoder backward chains):
// spin bounds, the two-level grid barrier and its error word.  Device code only; every translation unit gets its own copy.
#pragma once
#include "common.h"

namespace {

constexpr unsigned PS_SPIN_MAX = 1u << 22;  // ~0.5 s of polling before a barrier gives up (error word, no hang)
// The FIRST hand-off of a launch waits for something else: for every workgroup to become RESIDENT.  When a foreign kernel (another
// stream's long-running kernel, RCCL's resident channels, another process) holds the LDS / registers of some CUs, the missing
// workgroups start when it leaves - seconds, not microseconds - and nothing is wrong.  Round 6: the start-up hand-offs (grid barrier
// epoch 1, pgen7's publish 1) get ~30 s of patience; every later hand-off is between resident workgroups and keeps the 0.5 s bound.
constexpr unsigned PS_SPIN_START = 1u << 28;
__device__ __forceinline__ unsigned ps_spin_limit(unsigned epoch) { return epoch <= 1u ? PS_SPIN_START : PS_SPIN_MAX; }
#define PS_RLX __ATOMIC_RELAXED
#define PS_AGENT __HIP_MEMORY_SCOPE_AGENT


// ---------------------------------------------------------------------------------------------------------------------------
// Grid barrier: two levels (8 groups by blockIdx % 8, observed = XCD; correctness does not depend on it), monotonic counters
// zeroed by the host before the launch, relaxed agent-scope atomics, sc1 payload drained by every wave before the arrive.
// Returns false when the spin bound was hit or another workgroup reported an error (every workgroup then leaves the kernel).
// ---------------------------------------------------------------------------------------------------------------------------
struct PsSync { unsigned* cnt; unsigned* err; };      // cnt[0] global, cnt[32 * (1 + g)] group g; err: device error word (0 = ok)


// arrive half: every wave drains its write-through stores, one lane bumps the counters (call BEFORE issuing loads that need not be
// complete at the barrier: the drain waits for everything this wave has in flight)
__device__ __forceinline__ void ps_bar_arrive(const PsSync& s, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned nwg = gridDim.x, ng = 8, g = blockIdx.x % ng, gsz = nwg / ng;
        const unsigned prev = __hip_atomic_fetch_add(s.cnt + 32 * (1 + g), 1u, PS_RLX, PS_AGENT);
        if (prev + 1 == epoch * gsz) __hip_atomic_fetch_add(s.cnt, 1u, PS_RLX, PS_AGENT);
    }
}
// wait half (thread 0 polls the top counter)
__device__ __forceinline__ bool ps_bar_wait(const PsSync& s, unsigned epoch) {
    if (threadIdx.x == 0) {
        const unsigned target = epoch * 8u, limit = ps_spin_limit(epoch);
        unsigned spins = 0;
        while (__hip_atomic_load(s.cnt, PS_RLX, PS_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && (spins > limit || __hip_atomic_load(s.err, PS_RLX, PS_AGENT) != 0)) {
                __hip_atomic_store(s.err, 2u, PS_RLX, PS_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    return __hip_atomic_load(s.err, PS_RLX, PS_AGENT) == 0;
}

__device__ __forceinline__ bool ps_barrier(const PsSync& s, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned nwg = gridDim.x, ng = 8, g = blockIdx.x % ng, gsz = nwg / ng;
        const unsigned prev = __hip_atomic_fetch_add(s.cnt + 32 * (1 + g), 1u, PS_RLX, PS_AGENT);
        if (prev + 1 == epoch * gsz) __hip_atomic_fetch_add(s.cnt, 1u, PS_RLX, PS_AGENT);
        const unsigned target = epoch * ng, limit = ps_spin_limit(epoch);
        unsigned spins = 0;
        while (__hip_atomic_load(s.cnt, PS_RLX, PS_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && (spins > limit || __hip_atomic_load(s.err, PS_RLX, PS_AGENT) != 0)) {
                __hip_atomic_store(s.err, 2u, PS_RLX, PS_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    return __hip_atomic_load(s.err, PS_RLX, PS_AGENT) == 0;
}


}  // namespace
