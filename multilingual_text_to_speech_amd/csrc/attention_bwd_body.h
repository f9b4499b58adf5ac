// Fast (MFMA) attention-step backward body, shared by attention_bwd.hip and the fused step launches.
#pragma once
#include "common.h"

constexpr int ATB_THREADS = 512;


__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < ATB_THREADS / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------------------------------------------------
// fast kernel (A = 64 or 128): every global operand is requested at entry; the three contractions over ds run on
// v_mfma_f32_16x16x4_f32 from LDS:
//   loc [rows x A]   = cumwin [rows x 32] * U^T          (PL recompute)
//   dU  [A x 32]     = ds^T   [A x rows]  * cumwin       (filter-bank gradient)
//   g   [rows x 32]  = ds     [rows x A]  * U            (cumulative-alignment gradient, then anti-diagonal sums)
// Wave w owns attention columns [16w, 16w+16).
// ------------------------------------------------------------------------------------------------------------
constexpr int BNU_MAX = 8;    // filter-bank elements per thread
constexpr int BNR_MAX = 4;    // own rows per wave
constexpr int BND_MAX = 9;    // memory floats per lane per row (Dm <= 576)
constexpr int BNP_MAX = 8;    // partial slabs (fused launch, 128-VGPR budget)
constexpr int BNX_MAX = 2;    // context floats per thread (Dm <= 1024)
constexpr int BUP_LD = 36;    // U row (32 taps + pad)
constexpr int BROWS = 32;     // padded row count of a chunk
constexpr int BG_LD = 34;     // g tile row (32 taps + pad)

template <int NP = BNP_MAX>
__device__ __forceinline__ void attn_bwd_body(const AttnBwdArgs& p, float* sm, const int b, const int ch) {
    const int tid = threadIdx.x, lane = tid & 63, nwaves = ATB_THREADS / 64;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.L, A = p.A, Dm = p.Dm, ksz = p.ksz, pad = (ksz - 1) / 2;
    const int lc = (L + p.nch - 1) / p.nch;
    const int l0 = ch * lc, l1 = min(L, l0 + lc), nl = max(0, l1 - l0);
    const int AK = A * ksz, DS_LD = A + 4;
    const int i16 = lane & 15, q4 = lane >> 4;
    float* q = sm;                       // [A]
    float* vv = q + A;                   // [A]
    float* bias = vv + A;                // [A]
    float* accq = bias + A;              // [A]
    float* accv = accq + A;              // [A]
    float* w = accv + A;                 // [L]
    float* dex = w + L;                  // [L]
    float* cumw = dex + L;               // [L + 64]
    float* de = cumw + L + 64;           // [BROWS]
    float* red = de + BROWS;             // [16]
    float* dctx_s = red + 16;            // [Dm]
    float* Up = sm + ((5 * A + 3 * L + 64 + BROWS + 64 + 16 + Dm + 3) & ~3);     // [A][BUP_LD]   U[a][tap]
    float* UT = Up + A * BUP_LD;         // [32][DS_LD]   U^T[tap][a]
    float* dsL = UT + 32 * DS_LD;        // [BROWS][DS_LD]
    float* gL = Up;                      // [2][BROWS][BG_LD]  g tiles of the two K halves; U[a][tap] is dead after the PL recompute
    const long slab = (long)b * p.nch + ch;

    // ---- burst of independent loads
    const int ac = min(tid, A - 1), lcl = min(tid, L - 1);
    const float q_r = p.q[(long)b * A + ac], v_r = p.v[ac], bias_r = p.bias[ac];
    const float w_r = p.w[(long)b * L + lcl], cum_r = p.cum_in[(long)b * L + lcl];
    const float dco_r = p.dcum_out[(long)b * L + lcl];
    const float dal_r = p.dalign ? p.dalign[(long)b * L + lcl] : 0.f;
    float dcx[BNX_MAX], cx[BNX_MAX];
#pragma unroll
    for (int j = 0; j < BNX_MAX; ++j) {
        const int d = min(tid + j * ATB_THREADS, Dm - 1);
        float g = p.dctx[(long)b * Dm + d];
        float pp[NP];
        const char* pb = reinterpret_cast<const char*>(p.part + (long)b * p.part_ld);
        const unsigned pstep = (unsigned)p.part_ks * 4u;
        unsigned po = (unsigned)d * 4u;
#pragma unroll
        for (int k = 0; k < NP; ++k) { pp[k] = (k < p.n_part) ? *reinterpret_cast<const float*>(pb + po) : 0.f; po += pstep; }
#pragma unroll
        for (int k = 0; k < NP; ++k) g += pp[k];
        dcx[j] = g;
        cx[j] = p.ctx[(long)b * Dm + d];
    }
    // Addresses of the burst = wave-uniform base + 32-bit per-lane byte offset (round 4: the kernel issues ~1 100 vector instructions
    // per wave at four waves per SIMD; a third of them were 64-bit address arithmetic and run-time integer divisions).
    float memr[BNR_MAX][BND_MAX];
    {
        unsigned mo[BND_MAX];
#pragma unroll
        for (int k = 0; k < BND_MAX; ++k) mo[k] = (unsigned)min(lane + 64 * k, Dm - 1) * 4u;
#pragma unroll
        for (int j = 0; j < BNR_MAX; ++j) {
            const int l = min(l0 + wave + j * nwaves, L - 1);                          // wave-uniform
            const char* mem = reinterpret_cast<const char*>(p.memory + ((long)b * L + l) * Dm);
#pragma unroll
            for (int k = 0; k < BND_MAX; ++k) memr[j][k] = *reinterpret_cast<const float*>(mem + mo[k]);
        }
    }
    // operands in the MFMA accumulator layout: rows 16*mt + 4*q4 + r, column 16*wave + i16
    const int a_own = min(16 * wave + i16, A - 1);
    float mtD[2][4], dmtD[2][4];
    {
        const char* mtb = reinterpret_cast<const char*>(p.Mt + (long)b * L * A);
        const char* dmb = reinterpret_cast<const char*>(p.dMt + (long)b * L * A);
        const unsigned rowb = (unsigned)A * 4u, omax = (unsigned)((L - 1) * A + a_own) * 4u;
        unsigned o = (unsigned)((l0 + 4 * q4) * A + a_own) * 4u;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned oc = min(o, omax);
                mtD[mt][r] = *reinterpret_cast<const float*>(mtb + oc); dmtD[mt][r] = *reinterpret_cast<const float*>(dmb + oc);
                o += rowb;
            }
            o += 12u * rowb;                                                           // next 16-row tile
        }
    }
    // The v / bias slabs of this (sample, chunk) have ONE writer per launch: no-return float atomics at L2 give the same result as
    // load + add + store without a load in the entry burst.  (Not so for dMt and the filter-bank slab: a million single-dword atomics per
    // launch are L2-rate bound - 15.6 against 12.6 us per launch, scripts/mb/mb_attn_bwd.hip.)

    // ---- stage in LDS
    if (tid < A) { q[tid] = q_r; vv[tid] = v_r; bias[tid] = bias_r; }
    if (tid < L) { w[tid] = w_r; dex[tid] = dal_r + dco_r; cumw[pad + tid] = cum_r; }
    if (tid < pad) cumw[tid] = 0.f;
    if (tid < 64 - pad) cumw[pad + L + tid] = 0.f;
    float sdot = 0.f;
#pragma unroll
    for (int j = 0; j < BNX_MAX; ++j) {
        const int d = tid + j * ATB_THREADS;
        if (d < Dm) {
            dctx_s[d] = dcx[j];
            if (ch == 0) p.dctx_total[(long)b * Dm + d] = dcx[j];
            sdot += dcx[j] * cx[j];
        }
    }
    if (tid < L) sdot += w_r * (dal_r + dco_r);
    sdot = wave_total_hi(sdot);
    if (lane == 63) red[wave] = sdot;
    __syncthreads();
    // the filter bank (the same 16 KB for every workgroup and step: L2 hits) is requested now and staged behind the dw stage, which
    // covers its latency; in the entry burst its 8 registers per thread pushed the 128-VGPR fused launch into spills
    // thread (channel ua = tid >> 2, taps 8 (tid & 3) .. + 7): no division by the tap count, and the zero taps ksz..31 come with it
    float us[BNU_MAX];
    const int ua = tid >> 2, uj = (tid & 3) * 8;
    {
        const int u0 = min(ua, A - 1) * ksz + uj;
#pragma unroll
        for (int j = 0; j < BNU_MAX; ++j) us[j] = p.U[min(u0 + j, AK - 1)];
    }
    float S = 0.f;                                        // softmax-backward scalar: sum_d dctx ctx + sum_l w (dalign + dcum)
#pragma unroll
    for (int i = 0; i < ATB_THREADS / 64; ++i) S += red[i];

    // ---- dw for the own rows (wave per row, memory rows already in registers), de = w (dw - S); padded rows -> 0
    float dwr[BNR_MAX];                                   // select instead of branch on the Dm tail: one LDS read per k, no exec-mask regions
#pragma unroll
    for (int j = 0; j < BNR_MAX; ++j) dwr[j] = 0.f;
#pragma unroll
    for (int k = 0; k < BND_MAX; ++k) {
        const int d = lane + 64 * k;
        const float xr = dctx_s[min(d, Dm - 1)], x = d < Dm ? xr : 0.f;
#pragma unroll
        for (int j = 0; j < BNR_MAX; ++j) dwr[j] += x * memr[j][k];       // memr beyond Dm is a clamped (finite) re-read times 0
    }
#pragma unroll
    for (int j = 0; j < BNR_MAX; ++j) {
        const int r = wave + j * nwaves;
        float acc = dwr[j];
        acc = wave_total_hi(acc);
        if (lane == 63 && r < BROWS) de[r] = (r < nl) ? w[l0 + r] * (dex[l0 + r] + acc - S) : 0.f;
    }
    if (ua < A) {
#pragma unroll
        for (int j = 0; j < BNU_MAX; ++j) {
            const int jj = uj + j;
            const float u = jj < ksz ? us[j] : 0.f;
            Up[ua * BUP_LD + jj] = u; UT[jj * DS_LD + ua] = u;
        }
    }
    // dU slab in the accumulator layout of the dU contraction (rows a = 16*wave + 4*q4 + r, columns tap = 16*nt + i16): requested here,
    // two stages ahead of its use, so that the entry burst stays inside the register budget
    float dusD[2][4];
    {
        const float* dUs_in = p.dU_slab + slab * AK;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = min(16 * wave + 4 * q4 + r, A - 1), jj = min(16 * nt + i16, ksz - 1);
                dusD[nt][r] = dUs_in[a * ksz + jj];
            }
    }
    __syncthreads();

    // ---- PL recompute on MFMA, ds = de * v * (1 - tanh^2), dMt accumulation, dq / dv column sums
    if (16 * wave < A) {
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 bf = *reinterpret_cast<const float4*>(Up + (16 * wave + i16) * BUP_LD + 16 * c + 4 * q4);
            const float bv[4] = {bf.x, bf.y, bf.z, bf.w};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float* cw = cumw + l0 + 16 * mt + i16 + 16 * c + 4 * q4;
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cw[s2], bv[s2], acc[mt], 0, 0, 0);
            }
        }
        const int a = 16 * wave + i16;
        const float qa = q[a_own], va = vv[a_own], ba = bias[a_own];
        float sq = 0.f, sv = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * mt + 4 * q4 + r;
                const float th = tanhf_(qa + mtD[mt][r] + ba + acc[mt][r]);
                const float der = de[rr];                         // 0 for padded rows
                const float dsv = der * va * (1.f - th * th);
                dsL[rr * DS_LD + a_own] = dsv;
                if (rr < nl && a < A) p.dMt[((long)b * L + l0 + rr) * A + a] = dmtD[mt][r] + dsv;
                sq += dsv; sv += der * th;
            }
        // column sums over the 4 row groups of this lane column (lanes i16, i16+16, i16+32, i16+48)
        sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
        sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
        if (q4 == 0 && a < A) { accq[a] = sq; accv[a] = sv; }
    }
    __syncthreads();
    if (tid < A) {
        atomicAdd(p.dq + (long)b * A + tid, accq[tid]);
        atomicAdd(p.dbias_slab + slab * A + tid, accq[tid]);
        atomicAdd(p.dv_slab + slab * A + tid, accv[tid]);
    }

    // ---- dU[a, tap] += sum_rows ds[row, a] * cumwin[row][tap]      (A operand ds^T, B operand Toeplitz window)
    if (16 * wave < A) {
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float av[4];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) av[s2] = dsL[(16 * c + 4 * q4 + s2) * DS_LD + 16 * wave + i16];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float* cw = cumw + l0 + 16 * c + 4 * q4 + 16 * nt + i16;       // cumwin[row][tap] = cumw[l0 + row + tap]
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s2], cw[s2], acc[nt], 0, 0, 0);
            }
        }
        float* dUs = p.dU_slab + slab * AK;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * wave + 4 * q4 + r, jj = 16 * nt + i16;
                if (a < A && jj < ksz) dUs[a * ksz + jj] = dusD[nt][r] + acc[nt][r];
            }
    }

    // ---- g[row, tap] = sum_a ds[row, a] * U[a][tap]; dcum window: dcum[row + tap] += g.  4 tiles x 2 K-halves over 8 waves; the tiles go to
    //      LDS and the anti-diagonals are summed in a fixed order (round 4: LDS float atomics here cost 2 us per launch and made the sum
    //      order-dependent)
    {
        const int tile = wave & 3, mt = tile >> 1, nt = tile & 1, kh = wave >> 2;
        const int nchunk = A >> 4, c_lo = kh * (nchunk >> 1), c_hi = kh ? nchunk : (nchunk >> 1);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int c = c_lo; c < c_hi; ++c) {
            const float4 af = *reinterpret_cast<const float4*>(dsL + (16 * mt + i16) * DS_LD + 16 * c + 4 * q4);
            const float4 bf = *reinterpret_cast<const float4*>(UT + (16 * nt + i16) * DS_LD + 16 * c + 4 * q4);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 16 * mt + 4 * q4 + r, jj = 16 * nt + i16;
            gL[(kh * BROWS + rr) * BG_LD + jj] = acc[r];
        }
    }
    __syncthreads();
    {   // window position i = row + tap: 8 lanes per position, rows = sub + 8 k
        const int i = tid >> 3, sub = tid & 7;
        float sg = 0.f;
#pragma unroll
        for (int k = 0; k < BROWS / 8; ++k) {
            const int rr = sub + 8 * k, jj = i - rr;
            const bool ok = rr < nl && jj >= 0 && jj < ksz;
            const int off = ok ? rr * BG_LD + jj : 0;
            const float g0 = gL[off], g1 = gL[BROWS * BG_LD + off];
            sg += ok ? g0 + g1 : 0.f;
        }
        sg += dpp_f<0xB1>(sg); sg += dpp_f<0x4E>(sg); sg += dpp_f<0x141>(sg);       // the 8 lanes of a position
        const int m = l0 - pad + i;
        if (sub == 0 && i < nl + ksz - 1 && m >= 0 && m < L) atomicAdd(p.dcum_in + (long)b * L + m, sg);
    }
    if (tid >= l0 && tid < l1) atomicAdd(p.dcum_in + (long)b * L + tid, dco_r);     // carry: cum_out = cum_in + w
}


static inline size_t attn_bwd_fast_lds(const AttnBwdArgs& p) {
    return sizeof(float) * ((((size_t)5 * p.A + 3 * p.L + 64 + BROWS + 64 + 16 + p.Dm + 3) & ~(size_t)3) + (size_t)p.A * BUP_LD +
                            (size_t)(32 + BROWS) * (p.A + 4));
}
static inline bool attn_bwd_fast_ok(const AttnBwdArgs& p, int max_part = BNP_MAX) {
    const int lc = (p.L + p.nch - 1) / p.nch;
    return (p.A == 64 || p.A == 128) && p.L <= ATB_THREADS && lc <= BROWS && lc <= BNR_MAX * (ATB_THREADS / 64) && p.ksz <= 32 &&
           (long)p.A * p.ksz <= (long)BNU_MAX * ATB_THREADS && p.Dm <= 64 * BND_MAX && p.Dm <= BNX_MAX * ATB_THREADS &&
           p.n_part <= max_part && attn_bwd_fast_lds(p) <= 64 * 1024;
}
