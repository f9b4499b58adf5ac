// Shared device/host helpers for libmtts_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mtts.h"

#define MTTS_API extern "C" __attribute__((visibility("default")))

// Thread-local last-error text, readable through mtts_last_error().
extern thread_local char g_mtts_err[512];
int mtts_fail(const char* fmt, ...);

#define MTTS_CHECK_HIP(expr)                                                         \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) return mtts_fail("%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define MTTS_CHECK_LAUNCH(name)                                                      \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) return mtts_fail("launch %s: %s", name, hipGetErrorString(_e)); \
    } while (0)

#define MTTS_REQUIRE(cond, ...)                                                      \
    do {                                                                             \
        if (!(cond)) return mtts_fail(__VA_ARGS__);                                  \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// tanh through exp: accurate to ~1e-7 abs, saturates cleanly.
__device__ __forceinline__ float tanhf_(float x) {
    float ax = fabsf(x);
    float e = __expf(-2.0f * ax);
    float r = (1.0f - e) / (1.0f + e);
    return copysignf(r, x);
}

// Wave-wide sum / max on DPP alone (quad_perm, row mirrors, gfx9 row_bcast15 / row_bcast31) + one v_readlane: no LDS round trips
// (the __shfl_xor butterfly is six dependent ds_bpermute round trips, ~0.3 us per reduction on a lone wave - round-4 timelines of the
// attention backward and of pdec's softmax).  Every lane gets the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_rows(float old, float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_rows<0xB1, 0xF>(0.f, v);      // quad_perm [1,0,3,2]
    v += dpp_rows<0x4E, 0xF>(0.f, v);      // quad_perm [2,3,0,1]
    v += dpp_rows<0x141, 0xF>(0.f, v);     // row_half_mirror
    v += dpp_rows<0x140, 0xF>(0.f, v);     // row_mirror: every lane of a 16-lane row holds the row's sum
    v += dpp_rows<0x142, 0xA>(0.f, v);     // row_bcast15: lane 15 of rows 0, 2 -> rows 1, 3
    v += dpp_rows<0x143, 0xC>(0.f, v);     // row_bcast31: lane 31 -> rows 2, 3; the total sits in row 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_rows<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_rows<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_rows<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_rows<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_rows<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_rows<0x143, 0xC>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}


__device__ __forceinline__ float apply_act(int act, float v) {
    switch (act) {
        case MTTS_ACT_RELU: return fmaxf(v, 0.0f);
        case MTTS_ACT_TANH: return tanhf_(v);
        case MTTS_ACT_SIGMOID: return sigmoidf_(v);
        default: return v;
    }
}

// (row, col) of the flat index of a grid-stride loop over a [rows, cols] array WITHOUT a division per element: two 64-bit divisions
// per thread at the start, then additions with one carry.  (`row = i / cols` on a 64-bit index is ~100 vector instructions per
// element - gemm_splitk_reduce and bn_apply were bound by it, not by their memory traffic; round 4.)
struct GridRC {
    long row, dr; int col, dc, cols;
    __device__ __forceinline__ GridRC(long i0, long stride, int cols_) {
        cols = cols_; row = i0 / cols; col = (int)(i0 - row * cols); dr = stride / cols; dc = (int)(stride - dr * cols);
    }
    __device__ __forceinline__ void next() { col += dc; row += dr; if (col >= cols) { col -= cols; ++row; } }
};

// ---- internal cross-file entry points (host) ----
int gemm_plain(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, bool tA,
               bool tB, float alpha, float beta, const float* bias, int act, hipStream_t s);
int skinny_launch(const SkinnyArgs& p, hipStream_t s);
int attn_step_launch(const AttnStepArgs& p, hipStream_t s);
int attn_step_nch(int B, int L, int A, int Dm, int ksz, int kq);
int attn_pl_init(const float* Mt, const float* bias, float* PL, long total, int A, hipStream_t s);
int copy2d(const float* in, float* out, int rows, int cols, int ldi, int ldo, hipStream_t s);

#define MTTS_TRY(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc) return _rc;      \
    } while (0)
int transpose2d(const float* in, float* out, int rows, int cols, hipStream_t s);
int transpose2d_ld(const float* in, float* out, int rows, int cols, int ldi, int ldo, hipStream_t s);
int colsum(const float* x, float* out, int rows, int cols, int ld, float* ws, hipStream_t s);
int relu_mask_bwd(const float* dy, const float* y, float* dz, long n, float scale, hipStream_t s);
bool prof_sample(int t, hipStream_t s, int phase);
// per-(device, caller stream) helper streams, ordering events and split-K scratch (common.cpp)
hipStream_t side_stream(hipStream_t s);
hipStream_t wgrad_stream(hipStream_t s);
hipEvent_t pool_event(hipStream_t s);
float* workspace_for(hipStream_t s, size_t* bytes);
int decoder_chunk();

// ---- DPP (no LDS traffic) reductions inside 16-lane rows, then across rows -----------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// every lane of a 16-lane row ends up with the row's sum
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_f<0xB1>(x);      // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);      // quad_perm [2,3,0,1]
    x += dpp_f<0x141>(x);     // row_half_mirror
    x += dpp_f<0x140>(x);     // row_mirror
    return x;
}
// wave_sum without the final broadcast: the total lands in lanes 48..63
__device__ __forceinline__ float wave_total_hi(float x) {
    x = row16_sum(x);
    x += dpp_rows<0x142, 0xA>(0.f, x);
    x += dpp_rows<0x143, 0xC>(0.f, x);
    return x;
}
// sum over groups of G consecutive lanes (G = 16, 32 or 64); every lane gets its group's sum
template <int G>
__device__ __forceinline__ float group_sum(float x) {
    x = row16_sum(x);
    if (G >= 32) x += __shfl_xor(x, 16, 64);
    if (G >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}
int add3(float* out, int ldo, const float* a, int lda, const float* b, int ldb, const float* c3, int ldc, int rows, int cols,
         bool accumulate, hipStream_t s);
int attn_bwd_plus_skinny(const AttnBwdArgs& a, const SkinnyArgs& k, hipStream_t s);
int lstm_step_launch(const LstmStepArgs& a, hipStream_t s);
int ls_pack_mode(int B, int precision, bool lone_chain);      // 2 = fp32 weights as three pre-split bf16 planes (lstm_step.hip)
// persistent recurrences (persist.hip)
bool persist_enabled();
bool pgen_supported(const DecoderArgs& a);
int pgen_launch(const DecoderArgs& a, int t0, int t1, hipStream_t s);
bool pdec_supported(const DecoderArgs& a);
int pdec_launch(const DecoderArgs& a, int t0, int t1, hipStream_t s);
bool ps_ready_ext(const void* fn, int threads, size_t lds);
int ps_run_launch(const DecoderArgs& a, hipStream_t s, void (*go)(void* ctx, unsigned* cnt, unsigned* err, hipStream_t s), void* ctx);
// persistent decoder backward: chain A (attention + attention LSTM) of a chunk in one launch (pbwd.hip)
struct PbwdChunk { int a0, a1; };      // chain A steps [a0, a1)
bool pbwd_supported(const DecoderArgs& a, const DecoderGradArgs& g);
int pbwd_launch(const DecoderArgs& a, const DecoderGradArgs& g, const PbwdChunk& c, hipStream_t s);
int prenet2_launch(const float* x, int ldx, int Kin, const float* w1, const float* b1, const float* w2, const float* b2, const uint8_t* m1,
                   const uint8_t* m2, float scale, float* y1, float* y2, int B, int P, hipStream_t s);
