// Data-movement kernels: embedding gather / scatter-add, strided 2-D copy, conv weight repack.
#include "common.h"

// ids outside [0, vocab) never touch memory: the row reads as zeros / contributes no gradient and *err (nullable) is raised,
// which the host turns into an exception at its next check (torch.nn.Embedding device-asserts instead).
__global__ void embedding_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids, float* __restrict__ out,
                                     int rows, int D, int ldo, int col0, long vocab, int* __restrict__ err) {
    const long total = (long)rows * D;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    GridRC rc(i0, stride, D);
    for (long i = i0; i < total; i += stride, rc.next()) {
        const long r = rc.row; const int d = rc.col;
        const int64_t id = ids[r];
        const bool ok = id >= 0 && id < vocab;
        if (!ok && d == 0 && err) atomicOr(err, 1);
        out[r * ldo + col0 + d] = ok ? table[id * D + d] : 0.f;
    }
}

__global__ void embedding_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ ids, float* __restrict__ dtable,
                                     int rows, int D, int ldo, int col0, int padding_idx, long vocab) {
    const long total = (long)rows * D;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    GridRC rc(i0, stride, D);
    for (long i = i0; i < total; i += stride, rc.next()) {
        const long r = rc.row; const int d = rc.col;
        const int64_t id = ids[r];
        if (id == padding_idx || id < 0 || id >= vocab) continue;
        atomicAdd(dtable + id * D + d, dout[r * ldo + col0 + d]);
    }
}

__global__ void copy2d_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int ldi, int ldo) {
    const long total = (long)rows * cols;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    GridRC rc(i0, stride, cols);
    for (long i = i0; i < total; i += stride, rc.next()) {
        const long r = rc.row; const int c = rc.col;
        out[r * ldo + c] = in[r * ldi + c];
    }
}

// torch conv weight [O][I][k] <-> implicit-GEMM layout [O][k][I]
__global__ void conv_pack_kernel(const float* __restrict__ in, float* __restrict__ out, int O, int I, int k, int to_packed) {
    const long total = (long)O * I * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i enumerates the packed layout (o, t, c)
        const long o = i / ((long)I * k); const long rem = i - o * (long)I * k;
        const int t = (int)(rem / I), c = (int)(rem - (long)t * I);
        const long torch_idx = (o * I + c) * k + t;
        if (to_packed) out[i] = in[torch_idx]; else out[torch_idx] = in[i];
    }
}

static inline int nblocks(long total) { long b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

MTTS_API int mtts_embedding_fwd(const float* table, const int64_t* ids, float* out, int rows, int D, int ldo, int col0,
                                long vocab, int* err, void* stream) {
    MTTS_REQUIRE(vocab > 0, "embedding_fwd: vocab must be positive");
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(nblocks((long)rows * D)), dim3(256), 0, (hipStream_t)stream, table, ids, out,
                       rows, D, ldo, col0, vocab, err);
    MTTS_CHECK_LAUNCH("embedding_fwd");
    return 0;
}

MTTS_API int mtts_embedding_bwd(const float* dout, const int64_t* ids, float* dtable, int rows, int D, int ldo, int col0,
                                int padding_idx, long vocab, void* stream) {
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(nblocks((long)rows * D)), dim3(256), 0, (hipStream_t)stream, dout, ids,
                       dtable, rows, D, ldo, col0, padding_idx, vocab);
    MTTS_CHECK_LAUNCH("embedding_bwd");
    return 0;
}

int copy2d(const float* in, float* out, int rows, int cols, int ldi, int ldo, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(copy2d_kernel, dim3(nblocks((long)rows * cols)), dim3(256), 0, s, in, out, rows, cols, ldi, ldo);
    MTTS_CHECK_LAUNCH("copy2d");
    return 0;
}

MTTS_API int mtts_copy2d(const float* in, float* out, int rows, int cols, int ldi, int ldo, void* stream) {
    return copy2d(in, out, rows, cols, ldi, ldo, (hipStream_t)stream);
}

MTTS_API int mtts_conv_weight_pack(const float* in, float* out, int O, int I, int k, int to_packed, void* stream) {
    hipLaunchKernelGGL(conv_pack_kernel, dim3(nblocks((long)O * I * k)), dim3(256), 0, (hipStream_t)stream, in, out, O, I, k,
                       to_packed);
    MTTS_CHECK_LAUNCH("conv_pack");
    return 0;
}

// Gradient reversal backward: out = clamp(g, -c, c) * (-l)   (reference modules/classifier.py:16-18)
__global__ void grad_reverse_kernel(const float* __restrict__ g, float* __restrict__ out, long n, float l, float c) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = -l * fminf(fmaxf(g[i], -c), c);
}

MTTS_API int mtts_grad_reverse_clamp(const float* g, float* out, long n, float l, float c, void* stream) {
    hipLaunchKernelGGL(grad_reverse_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, g, out, n, l, c);
    MTTS_CHECK_LAUNCH("grad_reverse");
    return 0;
}

// out[c*rows + r] = in[r*cols + c], LDS-tiled 32x32
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int ldi, int ldo) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[i][threadIdx.x] = in[(long)r * ldi + c];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[(long)c * ldo + r] = tile[threadIdx.x][i];
    }
}

int transpose2d_ld(const float* in, float* out, int rows, int cols, int ldi, int ldo, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, s, in, out, rows, cols, ldi, ldo);
    MTTS_CHECK_LAUNCH("transpose");
    return 0;
}

int transpose2d(const float* in, float* out, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(32, 8), 0, s, in, out, rows, cols, cols, rows);
    MTTS_CHECK_LAUNCH("transpose");
    return 0;
}

MTTS_API int mtts_transpose(const float* in, float* out, int rows, int cols, void* stream) {
    return transpose2d(in, out, rows, cols, (hipStream_t)stream);
}

// column sums, two deterministic stages: partials [chunks, cols] then final
constexpr int CS_CHUNKS = 64;
__global__ void colsum_part_kernel(const float* __restrict__ x, float* __restrict__ ws, int rows, int cols, int ld) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int nch = gridDim.y;
    const int per = (rows + nch - 1) / nch;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + threadIdx.y; r < r1; r += 4) s += x[(long)r * ld + c];
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < cols) ws[(long)blockIdx.y * cols + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void colsum_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int cols, int nch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k0 = 0; k0 < nch; k0 += 16) {          // sixteen slabs per memory round trip, added in slab order
        float ps[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) ps[j] = ws[(long)min(k0 + j, nch - 1) * cols + c];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (k0 + j < nch) s += ps[j];
    }
    out[c] = s;
}

int colsum(const float* x, float* out, int rows, int cols, int ld, float* ws, hipStream_t s) {
    int nch = cdiv(rows, 256); nch = nch < 1 ? 1 : (nch > CS_CHUNKS ? CS_CHUNKS : nch);
    hipLaunchKernelGGL(colsum_part_kernel, dim3(cdiv(cols, 64), nch), dim3(64, 4), 0, s, x, ws, rows, cols, ld);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, ws, out, cols, nch);
    MTTS_CHECK_LAUNCH("colsum");
    return 0;
}

MTTS_API long mtts_colsum_workspace_floats(int cols) { return (long)CS_CHUNKS * cols; }
MTTS_API int mtts_colsum(const float* x, float* out, int rows, int cols, int ld, float* ws, void* stream) {
    return colsum(x, out, rows, cols, ld, ws, (hipStream_t)stream);
}

__global__ void relu_mask_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz, long n, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dz[i] = y[i] > 0.f ? dy[i] * scale : 0.f;
}

int relu_mask_bwd(const float* dy, const float* y, float* dz, long n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, s, dy, y, dz, n, scale);
    MTTS_CHECK_LAUNCH("relu_mask_bwd");
    return 0;
}

MTTS_API int mtts_relu_mask_bwd(const float* dy, const float* y, float* dz, long n, float scale, void* stream) {
    return relu_mask_bwd(dy, y, dz, n, scale, (hipStream_t)stream);
}

// ---- MFMA tile order ("packed") copies: [tiles][K/16][64 lanes][4]; lane 16q+i <-> row 16*tile+i, columns 16c+4q..+3 ----
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int ld, int N, int K, int lstm_H, int ntile) {
    const int nc = K >> 4;
    const long total = (long)ntile * nc * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long tc = i >> 6;
        const int c = (int)(tc % nc), cb = (int)(tc / nc);
        const int j = lane & 15, q = lane >> 4;
        int row; bool ok;
        if (lstm_H > 0) { const int u = cb * 4 + (j & 3); row = (j >> 2) * lstm_H + u; ok = u < lstm_H; }
        else { row = cb * 16 + j; ok = row < N; }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            const float* sp = src + (long)row * ld + c * 16 + 4 * q;
            v = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
        *reinterpret_cast<float4*>(dst + i * 4) = v;
    }
}

MTTS_API int mtts_pack_weight(const float* src, int ld, int N, int K, int lstm_H, float* dst, void* stream) {
    MTTS_REQUIRE((K & 15) == 0, "mtts_pack_weight: K = %d must be a multiple of 16", K);
    const int ntile = lstm_H > 0 ? cdiv(lstm_H, 4) : cdiv(N, 16);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(nblocks((long)ntile * (K >> 4) * 64)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       ld, N, K, lstm_H, ntile);
    MTTS_CHECK_LAUNCH("pack_weight");
    return 0;
}

// bf16 tile order of the per-step backward products in bf16 mode (skinny_body.h, PK = 3): [tiles][K/32][64 lanes][16 B]; lane 16q+i
// holds, RNE-rounded, columns 32c + 4q .. + 3 (low 8 bytes: the operand of the first v_mfma_f32_16x16x16_bf16 of the pair) and
// 32c + 16 + 4q .. + 3 (high 8 bytes: the second) of row 16 tile + i.
__device__ __forceinline__ unsigned pw_bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ void pack_weight_bf16_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int ld, int N, int K, int ntile) {
    const int nc = K >> 5;
    const long total = (long)ntile * nc * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long tc = i >> 6;
        const int c = (int)(tc % nc), cb = (int)(tc / nc);
        const int j = lane & 15, q = lane >> 4, row = cb * 16 + j;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < N) {
            const float* sp = src + (long)row * ld + c * 32 + 4 * q;
            v.x = pw_bf16_rne(sp[0]) | (pw_bf16_rne(sp[1]) << 16); v.y = pw_bf16_rne(sp[2]) | (pw_bf16_rne(sp[3]) << 16);
            v.z = pw_bf16_rne(sp[16]) | (pw_bf16_rne(sp[17]) << 16); v.w = pw_bf16_rne(sp[18]) | (pw_bf16_rne(sp[19]) << 16);
        }
        dst[i] = v;
    }
}

MTTS_API int mtts_pack_weight_bf16(const float* src, int ld, int N, int K, void* dst, void* stream) {
    MTTS_REQUIRE((K & 31) == 0, "mtts_pack_weight_bf16: K = %d must be a multiple of 32", K);
    const int ntile = cdiv(N, 16);
    hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(nblocks((long)ntile * (K >> 5) * 64)), dim3(256), 0, (hipStream_t)stream, src,
                       (uint4*)dst, ld, N, K, ntile);
    MTTS_CHECK_LAUNCH("pack_weight_bf16");
    return 0;
}

MTTS_API int mtts_pack_rows(const float* src, int ld, int rows, int K, float* dst, void* stream) {
    MTTS_REQUIRE((K & 15) == 0, "mtts_pack_rows: K = %d must be a multiple of 16", K);
    const int ntile = cdiv(rows, 16);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(nblocks((long)ntile * (K >> 4) * 64)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       ld, rows, K, 0, ntile);
    MTTS_CHECK_LAUNCH("pack_rows");
    return 0;
}

// out[r, c] (+)= a[r, c] + b[r, c] + c3[r, c]  with independent leading dimensions (NULL operands are skipped)
__global__ void add3_kernel(float* __restrict__ out, int ldo, const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                            const float* __restrict__ c3, int ldc, int rows, int cols, int accumulate) {
    const long total = (long)rows * cols;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    GridRC rc(i0, stride, cols);
    for (long i = i0; i < total; i += stride, rc.next()) {
        const long r = rc.row; const int c = rc.col;
        float v = accumulate ? out[r * ldo + c] : 0.f;
        if (a) v += a[r * lda + c];
        if (b) v += b[r * ldb + c];
        if (c3) v += c3[r * ldc + c];
        out[r * ldo + c] = v;
    }
}

int add3(float* out, int ldo, const float* a, int lda, const float* b, int ldb, const float* c3, int ldc, int rows, int cols,
         bool accumulate, hipStream_t s) {
    hipLaunchKernelGGL(add3_kernel, dim3(nblocks((long)rows * cols)), dim3(256), 0, s, out, ldo, a, lda, b, ldb, c3, ldc, rows, cols,
                       accumulate ? 1 : 0);
    MTTS_CHECK_LAUNCH("add3");
    return 0;
}


// ---- dropout keep flags ---------------------------------------------------------------------------------------------------
// uint8 keep mask (1 = keep) with P(keep) = 1 - p, drawn with Philox4x32-10 (Salmon et al., SC'11; counter = element block,
// key = seed): one call yields 128 bits = eight 16-bit uniforms = eight flags.  Replaces torch.rand(shape) >= p + cast
// (write 4 B, read 4 B, write 1 B per element) on the host side of every Dropout / zoneout site (modules/layers.py:27,37-40).
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], const uint32_t (&k)[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void dropout_keep_kernel(uint8_t* __restrict__ out, long n, uint32_t thresh16, uint64_t seed, uint64_t offset) {
    const long nblk = (n + 7) / 8;
    for (long blk = (long)blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += (long)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset + (uint64_t)blk;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k);
            k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
        }
        uint8_t f[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = (c[i] & 0xffffu) >= thresh16; f[2 * i + 1] = (c[i] >> 16) >= thresh16; }
        const long e0 = blk * 8;
        if (e0 + 8 <= n && (((uintptr_t)(out + e0)) & 7) == 0) {
            uint64_t w = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) w |= (uint64_t)f[i] << (8 * i);
            *reinterpret_cast<uint64_t*>(out + e0) = w;
        } else {
            for (int i = 0; i < 8 && e0 + i < n; ++i) out[e0 + i] = f[i];
        }
    }
}

MTTS_API int mtts_dropout_keep_mask(uint8_t* out, long n, float p, uint64_t seed, uint64_t offset, void* stream) {
    if (n <= 0) return 0;
    MTTS_REQUIRE(p >= 0.f && p < 1.f, "dropout probability %f outside [0, 1)", (double)p);
    const uint32_t thresh = (uint32_t)(p * 65536.0f + 0.5f);      // keep iff u16 >= p * 2^16
    const long nblk = (n + 7) / 8;
    int blocks = (int)((nblk + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_keep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, n, thresh, seed, offset);
    MTTS_CHECK_LAUNCH("dropout_keep_kernel");
    return 0;
}


// ---- stop rule of batched free-running synthesis, on the device ---------------------------------------------------------------------
// Reference Decoder._decode, modules/tacotron2.py:201-207 (batch 1): a frame whose stop probability reaches the threshold arms a
// counter (stop_frames) the first time and decrements it afterwards; the utterance ends at the frame where the counter reaches 0.
// Per sample b over the frames [t0, t1) of `out` ([T+1][B][Mo], stop logit in column M, slot t+1 = frame t):
//   state[b] = armed (-1 = not armed), state[B + b] = done (-1 = running, else the frame count); *running = samples still running.
// One thread per sample walks its frames in order (the rule is sequential in t); the host reads ONE int per chunk.
__global__ void stop_rule_kernel(const float* __restrict__ out, int t0, int t1, int B, int Mo, int M, float logit_threshold, int stop_frames,
                                 int* __restrict__ state, int* __restrict__ running_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int running = 0;
    if (b < B) {
        int armed = state[b], done = state[B + b];
        for (int t = t0; t < t1 && done < 0; ++t) {
            const bool f = out[((long)(t + 1) * B + b) * Mo + M] >= logit_threshold;      // sigmoid(x) >= p  <=>  x >= logit(p)
            if (f) {
                if (armed == -1) armed = stop_frames;
                else if (--armed == 0) done = t + 1;
            }
        }
        state[b] = armed; state[B + b] = done;
        running = done < 0;
    }
    // running count (block-level, then one atomic per block into a zeroed word)
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    if (running) atomicAdd(&cnt, 1);
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(running_out, cnt);
}

// state: int[2B] device array, filled with -1 before the FIRST chunk; *running (device int, cleared here on the stream) receives the
// number of utterances still running after this chunk - give every in-flight chunk its own word.  stop_threshold in (0, 1) is the
// probability threshold (>= 1 disables the rule).
MTTS_API int mtts_stop_rule_update(const float* out, int t0, int t1, int B, int Mo, int M, float stop_threshold, int stop_frames, int* state,
                                   int* running, void* stream) {
    MTTS_REQUIRE(B > 0 && t1 >= t0 && M < Mo && stop_frames >= 1, "mtts_stop_rule_update: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    MTTS_REQUIRE(state && running, "mtts_stop_rule_update: state / running are required");
    MTTS_CHECK_HIP(hipMemsetAsync(running, 0, sizeof(int), s));
    float thr;
    if (stop_threshold >= 1.f) thr = INFINITY;
    else if (stop_threshold <= 0.f) thr = -INFINITY;
    else thr = logf(stop_threshold / (1.f - stop_threshold));
    hipLaunchKernelGGL(stop_rule_kernel, dim3((B + 127) / 128), dim3(128), 0, s, out, t0, t1, B, Mo, M, thr, stop_frames, state, running);
    MTTS_CHECK_LAUNCH("stop_rule_kernel");
    return 0;
}
