// Location-sensitive attention, one decoder step (forward).
// Reference: modules/attention.py:39-45 (forward), :67-74 (_attent), :76-83 (_normalize), :85-86 (_combine_weights).
//
// Critical path per step: energies e[l] = v . tanh(q + PL[l]) -> masked softmax -> context = w . memory.
// PL = M + bias + loc(cum) depends only on the cumulative alignment, so the workgroups of step t also
// produce PL for step t+1 (31-tap filter bank U = W_loc * W_conv over the LDS-staged cum window).
// Grid (B, nch): every workgroup of a sample recomputes the (cheap) energies/softmax so that the context
// columns and the PL_next rows of that sample can be split over nch CUs without a second launch.
//
// The step is latency bound (a few hundred KB per sample behind ~1-2 us memory round trips), so the fast
// kernel requests EVERYTHING it will need - query partials, PL, cumulative alignment, its memory columns,
// its rows of the memory transform, the filter bank - in one burst at entry (none of those addresses
// depends on computed data) and only then starts the dependent phases.  A generic kernel without the
// register-resident prefetch covers shapes outside the fast kernel's static bounds.
#include "common.h"
#include <stdlib.h>

constexpr int ATT_THREADS = 512;
constexpr int NC_MAX = 8;     // memory float4 per thread   (ceil(L/ng) <= NC_MAX)
constexpr int NU_MAX = 8;     // U elements per thread      (A*ksz    <= NU_MAX * ATT_THREADS)
constexpr int KQ_PER = 16;   // query partial slabs per thread: kq <= KQ_PER * (ATT_THREADS / A)

struct AttLds {
    float *q, *vv, *bias, *w, *cumw, *Us, *part;
};

__device__ __forceinline__ AttLds att_carve(float* sm, int A, int L, int ksz) {
    AttLds s;
    s.q = sm; s.vv = s.q + A; s.bias = s.vv + A; s.w = s.bias + A; s.cumw = s.w + L; s.Us = s.cumw + L + ksz - 1;
    s.part = sm + (((3 * A + 2 * L + ksz - 1 + A * ksz) + 3) & ~3);
    return s;
}
static inline size_t att_lds_bytes(int A, int L, int ksz) {
    return sizeof(float) * ((((size_t)3 * A + 2 * L + ksz - 1 + (size_t)A * ksz + 3) & ~(size_t)3) + 4 * ATT_THREADS);
}

// floor(a / b) for 0 <= a <= 2048, 1 <= b <= 2048 without the integer-division sequence (~25 vector instructions): float reciprocal
// + one correction step each way
__device__ __forceinline__ int small_div(int a, int b) {
    int q = (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b));
    q -= (q * b > a);
    q += ((q + 1) * b <= a);
    return q;
}
// tanh with the hardware reciprocal (1 ulp) instead of the IEEE division sequence (10 instructions): the energies of the large-batch
// kernel evaluate it 4 NTE times per lane
__device__ __forceinline__ float tanh_rcp_(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

// ------------------------------------------------------------------------------------------------------------
// fast kernel: all global loads up front; energies reduced with DPP; the location filter bank runs on MFMA
//   loc[l, a] = sum_k cumwin[l][k] * U[a][k]   (rows of this workgroup x A x 32 taps  ->  v_mfma_f32_16x16x4_f32)
// Wave w owns attention columns [16w, 16w+16) (A <= 128) for both 16-row tiles of the chunk.
// ------------------------------------------------------------------------------------------------------------
constexpr int UP_LD = 36;     // filter-bank row: 32 taps + 4 pad floats (conflict-free ds_read_b128)
// NE4_MAX = PL float4 per thread (L*A/4 <= NE4_MAX * ATT_THREADS): 8 covers the training shapes (L <= 128 at A = 128) inside 128
// VGPRs, 16 covers synthesis inputs up to L = 256.

template <int G, int NE4_MAX, int NMT = 2>     // G = A / 4 lanes per position (16 or 32); NMT 16-row tiles of PL_next per workgroup
__global__ __launch_bounds__(ATT_THREADS) void attn_step_kernel(AttnStepArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, ch = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.L, A = p.A, Dm = p.Dm, ksz = p.ksz, pad = (ksz - 1) / 2;
    float* q = sm;                       // [A]
    float* vv = q + A;                   // [A]
    float* bias = vv + A;                // [A]
    float* w = bias + A;                 // [L]
    float* cumw = w + L;                 // [L + 64]  cum_out with zero halo / slack for the 32-tap MFMA window
    float* Up = sm + ((3 * A + 2 * L + 64 + 3) & ~3);          // [A][UP_LD]
    float* part = Up + A * UP_LD;        // [4 * ATT_THREADS]
    const int LA4 = (L * A) >> 2;

    // ---- geometry of this workgroup's shares
    // (round 4, as in the large-batch kernel below: no run-time integer division on the vector pipe, addresses = wave-uniform base +
    //  32-bit per-lane byte offset advanced by additions, every request unconditional, hardware reciprocal in the energies' tanh: at
    //  batch 1 - four workgroups - this launch was 15.8 us of 4 000 instructions per wave)
    const int dc = ((int)small_div(Dm + p.nch - 1, p.nch) + 3) & ~3;
    const int d0 = ch * dc, d1 = min(Dm, d0 + dc);
    const int nc4 = max(0, (d1 - d0) >> 2);
    const int ng = nc4 > 0 ? max(1, small_div(ATT_THREADS, nc4)) : 1;
    const int cg = nc4 > 0 ? small_div(tid, nc4) : ng, c4 = nc4 > 0 ? tid - cg * nc4 : 0;
    const int lc = small_div(L + p.nch - 1, p.nch);
    const int l0 = ch * lc, l1 = min(L, l0 + lc);
    const int i16 = lane & 15, q4 = lane >> 4;
    const int log2A = A == 128 ? 7 : 6;

    // ---- burst of independent loads
    const int len = min(p.lengths[b], L);
    // query partials: thread (group g = tid / A, channel a = tid % A) takes slabs g, g + ngrp, ...
    const int q_ngrp = ATT_THREADS >> log2A, q_g = tid >> log2A, q_a = tid & (A - 1);
    float qp[KQ_PER];
    {
        const char* qb = reinterpret_cast<const char*>(p.qpart + (long)b * A);
        const unsigned q_ks = (unsigned)p.q_ks * 4u;
        const unsigned qo_max = (unsigned)(p.kq - 1) * q_ks + (unsigned)q_a * 4u, qo_step = (unsigned)q_ngrp * q_ks;
        unsigned qo = (unsigned)q_g * q_ks + (unsigned)q_a * 4u;
#pragma unroll
        for (int k = 0; k < KQ_PER; ++k) { qp[k] = *reinterpret_cast<const float*>(qb + min(qo, qo_max)); qo += qo_step; }
    }
    const float v_r = p.v[min(tid, A - 1)];
    const float bias_r = p.bias[min(tid, A - 1)];
    const float cum_r = p.cum_in[(long)b * L + min(tid, L - 1)];
    float4 pl4[NE4_MAX];
    {
        const char* PLb = reinterpret_cast<const char*>(p.PL + (long)b * L * A);
        const unsigned po_max = (unsigned)(LA4 - 1) * 16u;
        unsigned po = (unsigned)tid * 16u;
#pragma unroll
        for (int j = 0; j < NE4_MAX; ++j) { pl4[j] = *reinterpret_cast<const float4*>(PLb + min(po, po_max)); po += ATT_THREADS * 16u; }
    }
    float4 mem4[NC_MAX];
    {
        const char* mb = reinterpret_cast<const char*>(p.memory + (long)b * L * Dm + (nc4 > 0 ? d0 : 0));
        const unsigned eo_max = (unsigned)((L - 1) * Dm + c4 * 4) * 4u, eo_step = (unsigned)(ng * Dm) * 4u;
        unsigned eo = (unsigned)(cg * Dm + c4 * 4) * 4u;
#pragma unroll
        for (int j = 0; j < NC_MAX; ++j) { mem4[j] = *reinterpret_cast<const float4*>(mb + min(eo, eo_max)); eo += eo_step; }
    }
    float mtD[NMT][4], us[NU_MAX];
    const int a_own = min(16 * wave + i16, A - 1);
    // filter bank: thread (channel ua = tid >> 2, taps 8 (tid & 3) .. + 7)
    const int ua = tid >> 2, uj = (tid & 3) * 8;
    if (p.PL_next) {
        const char* mtb = reinterpret_cast<const char*>(p.Mt + (long)b * L * A);
        const unsigned rowb = (unsigned)A * 4u, omax = (unsigned)((L - 1) * A + a_own) * 4u;
        unsigned o = (unsigned)((l0 + 4 * q4) * A + a_own) * 4u;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { mtD[mt][r] = *reinterpret_cast<const float*>(mtb + min(o, omax)); o += rowb; }
            o += 12u * rowb;
        }
        const int u0 = min(ua, A - 1) * ksz + uj, umax = A * ksz - 1;
#pragma unroll
        for (int j = 0; j < NU_MAX; ++j) us[j] = p.U[min(u0 + j, umax)];
    }

    // ---- q partial sums, v, bias, filter bank -> LDS
    {
        float qs = 0.f;
#pragma unroll
        for (int k = 0; k < KQ_PER; ++k) qs += (q_g + k * q_ngrp < p.kq) ? qp[k] : 0.f;
        part[tid] = qs;
    }
    if (tid < A) { vv[tid] = v_r; bias[tid] = bias_r; }
    if (p.PL_next) {
        if (ua < A) {
#pragma unroll
            for (int j = 0; j < NU_MAX; ++j) Up[ua * UP_LD + uj + j] = (uj + j < ksz) ? us[j] : 0.f;
        }
        if (tid < A) *reinterpret_cast<float4*>(Up + tid * UP_LD + 32) = make_float4(0.f, 0.f, 0.f, 0.f);      // pad floats of the row
    }
    __syncthreads();
    if (tid < A) {
        float qs = 0.f;
        for (int g = 0; g < q_ngrp; ++g) qs += part[g * A + tid];
        q[tid] = qs;
        if (ch == 0 && p.q_out) p.q_out[(long)b * A + tid] = qs;
    }
    __syncthreads();

    // ---- energies: lane holds 4 consecutive attention channels of one position; G lanes cover the position
    {
        const int a0 = 4 * (tid % G);
        const float4 q4v = *reinterpret_cast<const float4*>(q + a0);
        const float4 v4v = *reinterpret_cast<const float4*>(vv + a0);
#pragma unroll
        for (int j = 0; j < NE4_MAX; ++j) {
            const int i4 = tid + j * ATT_THREADS;
            float e = v4v.x * tanh_rcp_(q4v.x + pl4[j].x) + v4v.y * tanh_rcp_(q4v.y + pl4[j].y) + v4v.z * tanh_rcp_(q4v.z + pl4[j].z) +
                      v4v.w * tanh_rcp_(q4v.w + pl4[j].w);
            e = group_sum<G>(e);
            if ((lane % G) == 0 && i4 < LA4) w[i4 / G] = e;
        }
    }
    __syncthreads();

    // ---- masked softmax (wave 0), then the cumulative-alignment window
    if (tid < 64) {
        float mx = -INFINITY;
        for (int l = lane; l < len; l += 64) mx = fmaxf(mx, w[l]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int l = lane; l < L; l += 64) { const float ex = (l < len) ? __expf(w[l] - mx) : 0.f; w[l] = ex; sum += ex; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int l = lane; l < L; l += 64) w[l] *= inv;
    }
    __syncthreads();
    if (tid < L) {
        const float wl = w[tid];
        const float cn = cum_r + wl;
        cumw[pad + tid] = cn;
        if (ch == 0) { p.w_out[(long)b * L + tid] = wl; p.cum_out[(long)b * L + tid] = cn; }
    }
    if (tid < pad) cumw[tid] = 0.f;
    if (tid < 64 - pad) cumw[pad + L + tid] = 0.f;

    // ---- context columns of this chunk (memory rows are already in registers)
    float4* part4 = reinterpret_cast<float4*>(part);
    {
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NC_MAX; ++j) {
            const int l = cg + j * ng;
            const float wl = (cg < ng && l < L) ? w[l] : 0.f;
            s4.x += wl * mem4[j].x; s4.y += wl * mem4[j].y; s4.z += wl * mem4[j].z; s4.w += wl * mem4[j].w;
        }
        part4[tid] = s4;
    }
    __syncthreads();
    if (tid < nc4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < ng; ++k) { const float4 v4 = part4[k * nc4 + tid]; t.x += v4.x; t.y += v4.y; t.z += v4.z; t.w += v4.w; }
        *reinterpret_cast<float4*>(p.ctx_out + (long)b * Dm + d0 + tid * 4) = t;
        if (p.ctx_pack_out) {       // MFMA tile order copy: row b, columns d .. d+3 are exactly one lane's float4
            const int d = d0 + tid * 4;
            *reinterpret_cast<float4*>(p.ctx_pack_out + ((((long)(b >> 4) * (Dm >> 4) + (d >> 4)) * 64) + 4 * (d & 12) + (b & 15)) * 4) = t;
        }
    }

    // ---- PL for the next step, rows [l0, l1): loc = cumwin x U^T on MFMA, + M + bias
    if (p.PL_next && 16 * wave < A) {
        f32x4 acc[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 bf = *reinterpret_cast<const float4*>(Up + (16 * wave + i16) * UP_LD + 16 * c + 4 * q4);
            const float bv[4] = {bf.x, bf.y, bf.z, bf.w};
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const float* cw = cumw + l0 + 16 * mt + i16 + 16 * c + 4 * q4;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cw[s2], bv[s2], acc[mt], 0, 0, 0);
            }
        }
        const int a = 16 * wave + i16;
        const float bb = bias[min(a, A - 1)];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = l0 + 16 * mt + 4 * q4 + r;
                if (l < l1 && a < A) p.PL_next[((long)b * L + l) * A + a] = acc[mt][r] + mtD[mt][r] + bb;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// large-batch kernel (B >= 128): ONE 1024-thread workgroup per sample (two when the context is too wide for one), and NO
// processed-memory buffer: PL[l, a] = Mt[l, a] + bias[a] + loc(cum)[l, a] is formed in registers from the cumulative alignment of
// THIS step (location filter bank on MFMA, as in the persistent kernel) right in front of the energies, instead of being written by
// step t - 1 and read back by step t.  Per sample and step that removes the read and the write of a [L, A] array (122 KB of the
// 353 KB at L = 120) - at batch 240 the step kernels are bandwidth bound (profiles/r04_fwd_decoder_b240_*).
// Wave w = (column tile w % (A / 16), row group w / (A / 16)) owns the row tiles {group + ngroups j}; the partial energies of a
// column tile are reduced over its 16 lanes (DPP) and summed over the column tiles in a fixed order by the softmax wave.
// ------------------------------------------------------------------------------------------------------------
constexpr int ATT_BIG = 1024;        // threads
constexpr int ATT_BIG_NCM = 9;       // memory float4 per thread
constexpr int ATT_BIG_ES = 272;      // row of the partial-energy buffer (>= 16 * 16 positions + pad)

template <int NTE>                   // row tiles of 16 positions per wave: L <= 16 * NTE * (16 / (A / 16))
__global__ __launch_bounds__(ATT_BIG) void attn_step_big_kernel(AttnStepArgs p) {
    constexpr int NT = ATT_BIG;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, ch = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.L, A = p.A, Dm = p.Dm, ksz = p.ksz, pad = (ksz - 1) / 2;
    float* q = sm;                       // [A]
    float* vv = q + A;                   // [A]
    float* bias = vv + A;                // [A]
    float* w = bias + A;                 // [L]
    float* cumw = w + L;                 // [L + 64]  cum_in with zero halo / slack for the 32-tap MFMA window
    float* Up = sm + ((3 * A + 2 * L + 64 + 3) & ~3);          // [A][UP_LD]
    float* part = Up + A * UP_LD;        // [4 * NT]
    float* es = part + 4 * NT;           // [A / 16][ATT_BIG_ES] partial energies per column tile

    // ---- geometry.  The kernel is bound by the instructions its 16 waves issue (4 per SIMD: 3 400 instructions per wave, 2 200 of
    //      them on the vector pipe, took 17 of the launch's 21 us), so nothing here divides by a run-time value on the vector pipe
    //      (A is 64 or 128; the column split of the context goes through a float reciprocal), and the address of every request is
    //      a wave-uniform base plus a 32-bit per-lane byte offset that advances by additions (round 4: 139 quarter-rate integer
    //      multiplies and 83 64-bit multiply-adds per wave before).
    const int dc = ((p.nch == 2 ? (Dm + 1) >> 1 : Dm) + 3) & ~3;
    const int d0 = ch * dc, d1 = min(Dm, d0 + dc);
    const int nc4 = max(0, (d1 - d0) >> 2);
    const int ng = nc4 > 0 ? max(1, small_div(NT, nc4)) : 1;
    const int cg = nc4 > 0 ? small_div(tid, nc4) : ng, c4 = nc4 > 0 ? tid - cg * nc4 : 0;
    const int i16 = lane & 15, q4 = lane >> 4;
    const int log2A = A == 128 ? 7 : 6;
    const int nct = A >> 4, ngroups = A == 128 ? 2 : 4;           // (NT / 64) / nct
    const int wcol = wave & (nct - 1), wgrp = wave >> (log2A - 4);
    const int a_own = 16 * wcol + i16;

    // ---- burst of independent loads.  Every one of them is UNCONDITIONAL on a clamped address and none of the values is touched
    //      before the last request has gone out (round 4): as `cond ? *p : 0` the query partials and the memory rows came out of the
    //      compiler as separate exec-masked blocks with `s_waitcnt vmcnt(0)` between them, and the sequence length - a per-sample
    //      scalar - was waited for right behind its load: five serial round trips in front of the first product.
    int len_raw = p.lengths[b];
    const int q_ngrp = NT >> log2A, q_g = tid >> log2A, q_a = tid & (A - 1);
    constexpr int KQP = KQ_PER / 2;
    float qp[KQP];
    {
        const char* qb = reinterpret_cast<const char*>(p.qpart + (long)b * A);
        const unsigned q_ks = (unsigned)p.q_ks * 4u;
        const unsigned qo_max = (unsigned)(p.kq - 1) * q_ks + (unsigned)q_a * 4u, qo_step = (unsigned)q_ngrp * q_ks;
        unsigned qo = (unsigned)q_g * q_ks + (unsigned)q_a * 4u;
#pragma unroll
        for (int k = 0; k < KQP; ++k) { qp[k] = *reinterpret_cast<const float*>(qb + min(qo, qo_max)); qo += qo_step; }
    }
    const float v_r = p.v[min(tid, A - 1)];
    const float bias_r = p.bias[min(tid, A - 1)];
    const float cum_r = p.cum_in[(long)b * L + min(tid, L - 1)];
    float mtD[NTE][4], us[4];
    {
        const char* mtb = reinterpret_cast<const char*>(p.Mt + (long)b * L * A);
        const unsigned mo_max = (unsigned)((L - 1) * A + a_own) * 4u, rowb = (unsigned)A * 4u;
        unsigned mo[4];
        mo[0] = (unsigned)(4 * q4 * A + a_own) * 4u;
#pragma unroll
        for (int r = 1; r < 4; ++r) mo[r] = mo[r - 1] + rowb;
#pragma unroll
        for (int j = 0; j < NTE; ++j) {
            const unsigned tb = (unsigned)(16 * (wgrp + ngroups * j)) * rowb;           // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) mtD[j][r] = *reinterpret_cast<const float*>(mtb + min(tb + mo[r], mo_max));
        }
    }
    // filter bank: thread (channel tid >> 3, taps 4 (tid & 7) .. + 3)
    const int ua = tid >> 3, uj = (tid & 7) * 4;
    {
        const int u0 = min(ua, A - 1) * ksz + uj, umax = A * ksz - 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) us[e] = p.U[min(u0 + e, umax)];
    }

    // ---- q partial sums, v, bias, filter bank, cumulative alignment -> LDS
    {
        float qs = 0.f;
#pragma unroll
        for (int k = 0; k < KQP; ++k) qs += (q_g + k * q_ngrp < p.kq) ? qp[k] : 0.f;
        part[tid] = qs;
    }
    if (tid < A) {
        vv[tid] = v_r; bias[tid] = bias_r;
        *reinterpret_cast<float4*>(Up + tid * UP_LD + 32) = make_float4(0.f, 0.f, 0.f, 0.f);            // pad floats of the row
    }
    if (ua < A)
        *reinterpret_cast<float4*>(Up + ua * UP_LD + uj) = make_float4(uj < ksz ? us[0] : 0.f, uj + 1 < ksz ? us[1] : 0.f,
                                                                       uj + 2 < ksz ? us[2] : 0.f, uj + 3 < ksz ? us[3] : 0.f);
    if (tid < L) cumw[pad + tid] = cum_r;
    if (tid < pad) cumw[tid] = 0.f;
    if (tid < 64 - pad) cumw[pad + L + tid] = 0.f;
    __syncthreads();
    // the memory rows of the context are only needed after the softmax: requested here, where the registers of the first burst
    // (query partials, filter bank) are free again, they travel under the location features, the energies and the softmax
    float4 mem4[ATT_BIG_NCM];
    {
        const char* mb = reinterpret_cast<const char*>(p.memory + (long)b * L * Dm + (nc4 > 0 ? d0 : 0));
        const unsigned eo_max = (unsigned)((L - 1) * Dm + c4 * 4) * 4u, eo_step = (unsigned)(ng * Dm) * 4u;
        unsigned eo = (unsigned)(cg * Dm + c4 * 4) * 4u;
#pragma unroll
        for (int j = 0; j < ATT_BIG_NCM; ++j) { mem4[j] = *reinterpret_cast<const float4*>(mb + min(eo, eo_max)); eo += eo_step; }
    }
    if (tid < A) {
        float qs = 0.f;
        for (int g = 0; g < q_ngrp; ++g) qs += part[g * A + tid];
        q[tid] = qs;
        if (ch == 0 && p.q_out) p.q_out[(long)b * A + tid] = qs;
    }
    __syncthreads();

    // ---- location features on MFMA + partial energies of this wave's tiles
    {
        const float qa = q[a_own] + bias[a_own], va = vv[a_own];
        float4 bf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[c] = *reinterpret_cast<const float4*>(Up + a_own * UP_LD + 16 * c + 4 * q4);
#pragma unroll
        for (int j = 0; j < NTE; ++j) {
            const int l0t = 16 * (wgrp + ngroups * j);
            if (l0t < L) {                                                       // wave-uniform
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float bv[4] = {bf[c].x, bf[c].y, bf[c].z, bf[c].w};
                    const float* cw = cumw + min(l0t + i16, L) + 16 * c + 4 * q4;
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cw[s2], bv[s2], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = row16_sum(va * tanh_rcp_(qa + mtD[j][r] + acc[r]));
                    const int l = l0t + 4 * q4 + r;
                    if (i16 == 0 && l < L) es[wcol * ATT_BIG_ES + l] = e;
                }
            }
        }
    }
    __syncthreads();

    // ---- masked softmax (wave 0): energies = sum over the column tiles in a fixed order
    asm volatile("" : "+v"(len_raw));                 // the length stays an opaque register value until here
    const int len = min(len_raw, L);
    if (tid < 64) {
        float mx = -INFINITY;
        for (int l = lane; l < L; l += 64) {
            float e = 0.f;
            for (int ct = 0; ct < nct; ++ct) e += es[ct * ATT_BIG_ES + l];
            w[l] = e;
            if (l < len) mx = fmaxf(mx, e);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int l = lane; l < L; l += 64) { const float ex = (l < len) ? __expf(w[l] - mx) : 0.f; w[l] = ex; sum += ex; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int l = lane; l < L; l += 64) w[l] *= inv;
    }
    __syncthreads();
    if (tid < L && ch == 0) {
        const float wl = w[tid];
        p.w_out[(long)b * L + tid] = wl;
        p.cum_out[(long)b * L + tid] = cumw[pad + tid] + wl;      // (the LDS copy: one register less across the kernel)
    }

    // ---- context columns of this chunk (memory rows are already in registers)
    float4* part4 = reinterpret_cast<float4*>(part);
    {
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < ATT_BIG_NCM; ++j) {
            const int l = cg + j * ng;
            const float wl = (cg < ng && l < L) ? w[l] : 0.f;
            s4.x += wl * mem4[j].x; s4.y += wl * mem4[j].y; s4.z += wl * mem4[j].z; s4.w += wl * mem4[j].w;
        }
        part4[tid] = s4;
    }
    __syncthreads();
    if (tid < nc4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < ng; ++k) { const float4 v4 = part4[k * nc4 + tid]; t.x += v4.x; t.y += v4.y; t.z += v4.z; t.w += v4.w; }
        *reinterpret_cast<float4*>(p.ctx_out + (long)b * Dm + d0 + tid * 4) = t;
        if (p.ctx_pack_out) {
            const int d = d0 + tid * 4;
            *reinterpret_cast<float4*>(p.ctx_pack_out + ((((long)(b >> 4) * (Dm >> 4) + (d >> 4)) * 64) + 4 * (d & 12) + (b & 15)) * 4) = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// generic kernel (any shape that fits the LDS budget)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ATT_THREADS) void attn_step_generic_kernel(AttnStepArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, ch = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = ATT_THREADS / 64;
    const int L = p.L, A = p.A, Dm = p.Dm, ksz = p.ksz, pad = (ksz - 1) / 2;
    const AttLds s = att_carve(sm, A, L, ksz);
    const int len = min(p.lengths[b], L);

    for (int a = tid; a < A; a += ATT_THREADS) {
        float qs = 0.f;
        for (int k = 0; k < p.kq; ++k) qs += p.qpart[(long)k * p.q_ks + (long)b * A + a];
        s.q[a] = qs; s.vv[a] = p.v[a]; s.bias[a] = p.bias[a];
        if (ch == 0 && p.q_out) p.q_out[(long)b * A + a] = qs;
    }
    if (p.PL_next)
        for (int i = tid; i < A * ksz; i += ATT_THREADS) s.Us[i] = p.U[i];
    __syncthreads();

    const float* PLb = p.PL + (long)b * L * A;
    for (int l = wave; l < L; l += nwaves) {
        float e = 0.f;
        if (l < len)
            for (int a = lane; a < A; a += 64) e += s.vv[a] * tanhf_(s.q[a] + PLb[(long)l * A + a]);
        e = wave_sum(e);
        if (lane == 0) s.w[l] = (l < len) ? e : -INFINITY;
    }
    __syncthreads();
    if (wave == 0) {
        float mx = -INFINITY;
        for (int l = lane; l < L; l += 64) mx = fmaxf(mx, s.w[l]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int l = lane; l < L; l += 64) { const float ex = (l < len) ? __expf(s.w[l] - mx) : 0.f; s.w[l] = ex; sum += ex; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int l = lane; l < L; l += 64) {
            const float wl = s.w[l] * inv;
            s.w[l] = wl;
            const float cn = p.cum_in[(long)b * L + l] + wl;
            s.cumw[pad + l] = cn;
            if (ch == 0) { p.w_out[(long)b * L + l] = wl; p.cum_out[(long)b * L + l] = cn; }
        }
        for (int i = lane; i < pad; i += 64) { s.cumw[i] = 0.f; s.cumw[pad + L + i] = 0.f; }
    }
    __syncthreads();
    {
        const int dc = (((Dm + p.nch - 1) / p.nch) + 3) & ~3;
        const int d0 = ch * dc, d1 = min(Dm, d0 + dc);
        const int nc4 = (d1 - d0) >> 2;
        if (nc4 > 0) {
            const int ng = max(1, ATT_THREADS / nc4);
            const int g = tid / nc4, c4 = tid % nc4;
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < ng) {
                const float* mem = p.memory + (long)b * L * Dm + d0 + c4 * 4;
                for (int l = g; l < len; l += ng) {
                    const float4 m4 = *reinterpret_cast<const float4*>(mem + (long)l * Dm);
                    const float wl = s.w[l];
                    s4.x += wl * m4.x; s4.y += wl * m4.y; s4.z += wl * m4.z; s4.w += wl * m4.w;
                }
            }
            float4* part4 = reinterpret_cast<float4*>(s.part);
            part4[tid] = s4;
            __syncthreads();
            if (tid < nc4) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = 0; k < ng; ++k) { const float4 v4 = part4[k * nc4 + tid]; t.x += v4.x; t.y += v4.y; t.z += v4.z; t.w += v4.w; }
                *reinterpret_cast<float4*>(p.ctx_out + (long)b * Dm + d0 + tid * 4) = t;
                if (p.ctx_pack_out) {
                    const int d = d0 + tid * 4;
                    *reinterpret_cast<float4*>(p.ctx_pack_out + ((((long)(b >> 4) * (Dm >> 4) + (d >> 4)) * 64) + 4 * (d & 12) + (b & 15)) * 4) = t;
                }
            }
        }
    }
    if (p.PL_next) {
        const int lc = (L + p.nch - 1) / p.nch;
        const int l0 = ch * lc, l1 = min(L, l0 + lc);
        const float* Mb = p.Mt + (long)b * L * A;
        float* out = p.PL_next + (long)b * L * A;
        for (int i = tid; i < (l1 - l0) * A; i += ATT_THREADS) {
            const int l = l0 + i / A, a = i % A;
            float acc = Mb[(long)l * A + a] + s.bias[a];
            const float* u = s.Us + a * ksz;
            const float* cw = s.cumw + l;
            for (int j = 0; j < ksz; ++j) acc += u[j] * cw[j];
            out[(long)l * A + a] = acc;
        }
    }
}

// PL0 = M + bias (cum = 0)
__global__ void attn_pl_init_kernel(const float* __restrict__ Mt, const float* __restrict__ bias, float* __restrict__ PL,
                                    long total, int A) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        PL[i] = Mt[i] + bias[i % A];
}

int attn_pl_init(const float* Mt, const float* bias, float* PL, long total, int A, hipStream_t s) {
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(attn_pl_init_kernel, dim3(blocks), dim3(256), 0, s, Mt, bias, PL, total, A);
    MTTS_CHECK_LAUNCH("attn_pl_init_kernel");
    return 0;
}

// ---- launch geometry ---------------------------------------------------------------------------------------------
static size_t att_big_lds(int A, int L) {
    return sizeof(float) * ((((size_t)3 * A + 2 * L + 64 + 3) & ~(size_t)3) + (size_t)A * UP_LD + 4 * ATT_BIG + (size_t)(A / 16) * ATT_BIG_ES);
}
static size_t att_fast_lds(int A, int L, int nt) {
    return sizeof(float) * ((((size_t)3 * A + 2 * L + 64 + 3) & ~(size_t)3) + (size_t)A * UP_LD + 4 * nt);
}
// can the large-batch kernel take this shape with `nch` workgroups per sample?
static bool att_big_ok(int B, int L, int A, int Dm, int ksz, int kq, int nch) {
    if (B < 128 || (A != 64 && A != 128) || nch < 1 || nch > 2 || L > 256 || ksz > 32 || (Dm & 3)) return false;
    const int dc = (((Dm + nch - 1) / nch) + 3) & ~3, nc4 = dc / 4;
    if (nc4 < 1 || nc4 > ATT_BIG) return false;
    const int ng = ATT_BIG / nc4, ngroups = (ATT_BIG / 64) / (A / 16);
    return (L + ng - 1) / ng <= ATT_BIG_NCM && (L + 15) / 16 <= 8 * ngroups && (long)A * ksz <= (long)(NU_MAX / 2) * ATT_BIG &&
           kq <= (KQ_PER / 2) * (ATT_BIG / A) && att_big_lds(A, L) <= 64 * 1024;
}

// workgroups per sample for a decoder step (the caller sizes nothing by it: every chunk of a sample writes its own rows / columns)
int attn_step_nch(int B, int L, int A, int Dm, int ksz, int kq) {
    // (two workgroups per sample where one fits were measured at batch 240: 86.5 -> 93.9 us per step fp32, 55.7 -> 66.0 bf16 - profiles/r05_attn_big_nch.txt)
    for (int nch = 1; nch <= 2; ++nch)
        if (att_big_ok(B, L, A, Dm, ksz, kq, nch)) return nch;          // large batch: one (two) 1024-thread workgroup(s) per sample
    int nch = (Dm + 511) / 512;
    if (nch < 4 && B * 4 <= 1024) nch = 4;                              // small batch: four workgroups per sample fill the chip
    if ((L + nch - 1) / nch > 64 && L <= 256) nch = (L + 63) / 64;      // <= 64 rows of PL_next per workgroup (MFMA path)
    return nch;
}

int attn_step_launch(const AttnStepArgs& p, hipStream_t s) {
    MTTS_REQUIRE((p.ksz & 1) == 1, "attention kernel size must be odd (got %d)", p.ksz);
    const int dc = (((p.Dm + p.nch - 1) / p.nch) + 3) & ~3;
    MTTS_REQUIRE(dc / 4 <= ATT_THREADS && (p.Dm & 3) == 0, "attn_step: Dm/nch = %d too wide or Dm %% 4 != 0", dc);
    const int G = p.A / 4;
    if (att_big_ok(p.B, p.L, p.A, p.Dm, p.ksz, p.kq, p.nch)) {
        const dim3 grid(p.B, p.nch), blk(ATT_BIG);
        const int ngroups = (ATT_BIG / 64) / (p.A / 16), nte = ((p.L + 15) / 16 + ngroups - 1) / ngroups;
        if (nte <= 4) hipLaunchKernelGGL((attn_step_big_kernel<4>), grid, blk, att_big_lds(p.A, p.L), s, p);
        else hipLaunchKernelGGL((attn_step_big_kernel<8>), grid, blk, att_big_lds(p.A, p.L), s, p);
        MTTS_CHECK_LAUNCH("attn_step_big_kernel");
        return 0;
    }
    const size_t lds = att_lds_bytes(p.A, p.L, p.ksz);
    const size_t lds_fast = att_fast_lds(p.A, p.L, ATT_THREADS);
    MTTS_REQUIRE(lds <= 64 * 1024, "attn_step: LDS request %zu too large", lds);
    const int nc4 = dc / 4, ng = nc4 > 0 ? (ATT_THREADS / nc4 > 0 ? ATT_THREADS / nc4 : 1) : 1;
    const int lc = (p.L + p.nch - 1) / p.nch;
    const long la4 = (long)p.L * p.A / 4;
    const bool fast = (p.A == 64 || p.A == 128) && p.L <= ATT_THREADS && la4 <= 16L * ATT_THREADS &&
                      (p.L + ng - 1) / ng <= NC_MAX && lc <= 64 && p.ksz <= 32 && lds_fast <= 64 * 1024 &&
                      (long)p.A * p.ksz <= (long)NU_MAX * ATT_THREADS && p.kq <= KQ_PER * (ATT_THREADS / p.A);
    const bool small = la4 <= 8L * ATT_THREADS && lc <= 32;
    const dim3 grid(p.B, p.nch), blk(ATT_THREADS);
    if (fast && G == 32 && small) hipLaunchKernelGGL((attn_step_kernel<32, 8>), grid, blk, lds_fast, s, p);
    else if (fast && G == 32) hipLaunchKernelGGL((attn_step_kernel<32, 16, 4>), grid, blk, lds_fast, s, p);
    else if (fast && G == 16 && small) hipLaunchKernelGGL((attn_step_kernel<16, 8>), grid, blk, lds_fast, s, p);
    else if (fast && G == 16) hipLaunchKernelGGL((attn_step_kernel<16, 16, 4>), grid, blk, lds_fast, s, p);
    else hipLaunchKernelGGL(attn_step_generic_kernel, grid, blk, lds, s, p);
    MTTS_CHECK_LAUNCH("attn_step_kernel");
    return 0;
}

MTTS_API int mtts_attn_step_form(int B, int L, int A, int Dm, int ksz, int kq, int nch) {
    return att_big_ok(B, L, A, Dm, ksz, kq, nch) ? 1 : 0;
}

MTTS_API int mtts_attn_step_fwd(const AttnStepArgs* args, void* stream) {
    return attn_step_launch(*args, (hipStream_t)stream);
}
