// Data-movement kernels: embedding gather / scatter-add, strided 2-D copy, conv weight repack.
#include "common.h"

__global__ void embedding_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids, float* __restrict__ out,
                                     int rows, int D, int ldo, int col0) {
    const long total = (long)rows * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / D; const int d = (int)(i - r * D);
        out[r * ldo + col0 + d] = table[ids[r] * D + d];
    }
}

__global__ void embedding_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ ids, float* __restrict__ dtable,
                                     int rows, int D, int ldo, int col0, int padding_idx) {
    const long total = (long)rows * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / D; const int d = (int)(i - r * D);
        const int64_t id = ids[r];
        if (id == padding_idx) continue;
        atomicAdd(dtable + id * D + d, dout[r * ldo + col0 + d]);
    }
}

__global__ void copy2d_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int ldi, int ldo) {
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols; const int c = (int)(i - r * cols);
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
                                void* stream) {
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(nblocks((long)rows * D)), dim3(256), 0, (hipStream_t)stream, table, ids, out,
                       rows, D, ldo, col0);
    MTTS_CHECK_LAUNCH("embedding_fwd");
    return 0;
}

MTTS_API int mtts_embedding_bwd(const float* dout, const int64_t* ids, float* dtable, int rows, int D, int ldo, int col0,
                                int padding_idx, void* stream) {
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(nblocks((long)rows * D)), dim3(256), 0, (hipStream_t)stream, dout, ids,
                       dtable, rows, D, ldo, col0, padding_idx);
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
