// Ablation micro-benchmark for the skinny LSTM step kernel (run on the GPU box: hipcc + execute).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NW, int VAR, int DEPTH>
__global__ __launch_bounds__(NW * 64) void k(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ out,
                                              int B, int K, int N, float* __restrict__ qacc, const float* __restrict__ Wq) {
    __shared__ float red[NW][64][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int cb = blockIdx.x;
    const int wrow = cb * 16 + li;
    int rows[4];
    for (int m = 0; m < 4; ++m) rows[m] = m * 16 + li;
    const int total = K / 16;
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 fw[DEPTH], fx[DEPTH][4];
    auto load = [&](int c, int slot) {
        const bool live = c < total;
        const int k = (live ? c : 0) * 16 + lq * 4;
        if (VAR != 4 && VAR != 2) { float4 v = *reinterpret_cast<const float4*>(W + (long)wrow * K + k); fw[slot] = live ? v : make_float4(0, 0, 0, 0); }
        if (VAR != 3 && VAR != 2)
            for (int m = 0; m < 4; ++m) { float4 v = *reinterpret_cast<const float4*>(X + (long)rows[m] * K + k); fx[slot][m] = live ? v : make_float4(0, 0, 0, 0); }
    };
    auto load_b = [&](int c, int slot) {     // uniform-branch variant
        if (c < total) {
            const int k = c * 16 + lq * 4;
            fw[slot] = *reinterpret_cast<const float4*>(W + (long)wrow * K + k);
            for (int m = 0; m < 4; ++m) fx[slot][m] = *reinterpret_cast<const float4*>(X + (long)rows[m] * K + k);
        } else { fw[slot] = make_float4(0, 0, 0, 0); for (int m = 0; m < 4; ++m) fx[slot][m] = make_float4(0, 0, 0, 0); }
    };
    // VAR 6: quad-coalesced loads (lane = 4*row + kq) + in-register transpose with ds_bpermute at use
    const int r4 = lane >> 2, kq4 = lane & 3;
    const int src_lane = 4 * (lane & 15) + (lane >> 4);
    auto load_c = [&](int c, int slot) {
        if (c < total) {
            const int k = c * 16 + kq4 * 4;
            fw[slot] = *reinterpret_cast<const float4*>(W + (long)(cb * 16 + r4) * K + k);
            for (int m = 0; m < 4; ++m) fx[slot][m] = *reinterpret_cast<const float4*>(X + (long)(m * 16 + r4) * K + k);
        } else { fw[slot] = make_float4(0, 0, 0, 0); for (int m = 0; m < 4; ++m) fx[slot][m] = make_float4(0, 0, 0, 0); }
    };
    auto perm4 = [&](float4 v) { return make_float4(__shfl(v.x, src_lane, 64), __shfl(v.y, src_lane, 64), __shfl(v.z, src_lane, 64), __shfl(v.w, src_lane, 64)); };
    auto mma_c = [&](int slot) {
        const float4 w4 = perm4(fw[slot]);
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
        for (int m = 0; m < 4; ++m) {
            const float4 x4 = perm4(fx[slot][m]);
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
            for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], wv[s], acc[m], 0, 0, 0);
        }
    };
    // VAR 7: operands pre-packed in MFMA tile order: [tile][chunk][64 lanes][4]
    auto load_p = [&](int c, int slot) {
        if (c < total) {
            fw[slot] = *reinterpret_cast<const float4*>(W + (((long)cb * total + c) * 64 + lane) * 4);
            for (int m = 0; m < 4; ++m) fx[slot][m] = *reinterpret_cast<const float4*>(X + (((long)m * total + c) * 64 + lane) * 4);
        } else { fw[slot] = make_float4(0, 0, 0, 0); for (int m = 0; m < 4; ++m) fx[slot][m] = make_float4(0, 0, 0, 0); }
    };
    auto mma = [&](int slot) {
        const float wv[4] = {fw[slot].x, fw[slot].y, fw[slot].z, fw[slot].w};
        for (int m = 0; m < 4; ++m) {
            const float xv[4] = {fx[slot][m].x, fx[slot][m].y, fx[slot][m].z, fx[slot][m].w};
            if (VAR == 1) { for (int s = 0; s < 4; ++s) acc[m][s] += xv[s] * wv[s]; }
            else for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s], wv[s], acc[m], 0, 0, 0);
        }
    };
    for (int d = 0; d < DEPTH; ++d) { fw[d] = make_float4(1, 1, 1, 1); for (int m = 0; m < 4; ++m) fx[d][m] = make_float4(1, 1, 1, 1); }
    int c = wave;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { if (VAR == 7 || VAR == 8) load_p(c + d * NW, d); else if (VAR == 6) load_c(c + d * NW, d); else if (VAR == 5) load_b(c + d * NW, d); else load(c + d * NW, d); }
    for (; c < total; c += DEPTH * NW) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (VAR == 6) mma_c(d); else mma(d);
            if (VAR == 7 || VAR == 8) load_p(c + (DEPTH + d) * NW, d); else if (VAR == 6) load_c(c + (DEPTH + d) * NW, d); else if (VAR == 5) load_b(c + (DEPTH + d) * NW, d); else load(c + (DEPTH + d) * NW, d);
        }
    }
    for (int m = 0; m < 4; ++m)
        for (int r = 0; r < 4; ++r) red[wave][m * 16 + lq * 4 + r][li] = acc[m][r];
    __syncthreads();
    for (int e = tid; e < 1024; e += NW * 64) {
        const int rr = e >> 4, cc = e & 15;
        float v = 0.f;
        for (int w = 0; w < NW; ++w) v += red[w][rr][cc];
        out[(long)rr * N + cb * 16 + cc] = v;
        if (VAR == 8) red[0][rr][cc] = v;
    }
    if (VAR == 8) {     // q[b, a] += sum over this block's 4 'units' of Wq[a][u] * h[b][u]  (8192 atomics per workgroup)
        __syncthreads();
        for (int e = tid; e < 64 * 128; e += NW * 64) {
            const int b = e >> 7, a = e & 127;
            const float4 wq = *reinterpret_cast<const float4*>(Wq + (long)a * 1024 + cb * 4);
            const float v = wq.x * red[0][b][0] + wq.y * red[0][b][1] + wq.z * red[0][b][2] + wq.w * red[0][b][3];
            atomicAdd(qacc + b * 128 + a, v);
        }
    }
}

template <int NW, int VAR, int DEPTH>
float run(const float* X, const float* W, float* out, int B, int K, int N, int iters) {
    static float* qacc = nullptr; static float* Wq = nullptr;
    if (!qacc) { hipMalloc(&qacc, 64 * 128 * 4); hipMalloc(&Wq, 128 * 1024 * 4); hipMemset(qacc, 0, 64 * 128 * 4); hipMemset(Wq, 0, 128 * 1024 * 4); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<NW, VAR, DEPTH>), dim3(N / 16), dim3(NW * 64), 0, 0, X, W, out, B, K, N, qacc, Wq);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<NW, VAR, DEPTH>), dim3(N / 16), dim3(NW * 64), 0, 0, X, W, out, B, K, N, qacc, Wq);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / iters;
}

int main() {
    const int B = 64, K = 1568, N = 4096, iters = 200;
    float *X, *W, *out;
    CK(hipMalloc(&X, (size_t)B * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&out, (size_t)B * N * 4));
    std::vector<float> h((size_t)N * K, 0.01f);
    CK(hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, h.data(), (size_t)B * K * 4, hipMemcpyHostToDevice));
    printf("B=%d K=%d N=%d  (us per launch incl. launch gap, back-to-back)\n", B, K, N);
    printf("NW8 full d4      %.2f\n", run<8, 0, 4>(X, W, out, B, K, N, iters));
    printf("NW8 loads only   %.2f\n", run<8, 1, 4>(X, W, out, B, K, N, iters));
    printf("NW8 mfma only    %.2f\n", run<8, 2, 4>(X, W, out, B, K, N, iters));
    printf("NW8 W+mfma       %.2f\n", run<8, 3, 4>(X, W, out, B, K, N, iters));
    printf("NW8 X+mfma       %.2f\n", run<8, 4, 4>(X, W, out, B, K, N, iters));
    printf("NW8 branch d4    %.2f\n", run<8, 5, 4>(X, W, out, B, K, N, iters));
    printf("NW8 branch d2    %.2f\n", run<8, 5, 2>(X, W, out, B, K, N, iters));
    printf("NW4 full d4      %.2f\n", run<4, 0, 4>(X, W, out, B, K, N, iters));
    printf("NW4 branch d4    %.2f\n", run<4, 5, 4>(X, W, out, B, K, N, iters));
    printf("NW4 branch d8    %.2f\n", run<4, 5, 8>(X, W, out, B, K, N, iters));
    printf("NW4 mfma only    %.2f\n", run<4, 2, 4>(X, W, out, B, K, N, iters));
    printf("NW4 loads only   %.2f\n", run<4, 1, 4>(X, W, out, B, K, N, iters));
    printf("NW16 branch d2   %.2f\n", run<16, 5, 2>(X, W, out, B, K, N, iters));
    printf("NW8 packed d4 %.2f\n", run<8, 7, 4>(X, W, out, B, K, N, iters));
    printf("NW8 packed d2 %.2f\n", run<8, 7, 2>(X, W, out, B, K, N, iters));
    printf("NW8 packed d2 + q atomics %.2f\n", run<8, 8, 2>(X, W, out, B, K, N, iters));
    printf("NW8 packed d6 %.2f\n", run<8, 7, 6>(X, W, out, B, K, N, iters));
    printf("NW4 packed d4 %.2f\n", run<4, 7, 4>(X, W, out, B, K, N, iters));
    printf("NW16 packed d2 %.2f\n", run<16, 7, 2>(X, W, out, B, K, N, iters));
    printf("NW8 coalesced+perm d4 %.2f\n", run<8, 6, 4>(X, W, out, B, K, N, iters));
    printf("NW4 coalesced+perm d4 %.2f\n", run<4, 6, 4>(X, W, out, B, K, N, iters));
    printf("NW8 coalesced+perm d2 %.2f\n", run<8, 6, 2>(X, W, out, B, K, N, iters));
    printf("NW4 coalesced+perm d8 %.2f\n", run<4, 6, 8>(X, W, out, B, K, N, iters));
    return 0;
}
