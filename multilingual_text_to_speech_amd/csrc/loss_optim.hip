// the training loss and the optimizer step as fused kernels.
//   mtts_tacotron_loss  - TacotronLoss.forward (reference modules/tacotron2.py:443-485): 2*MSE(pre) + MSE(post) + weighted BCE(stop)
//                         + guided attention (no per-sample Python loop), values AND gradients in one pass.
//   mtts_clip_adam_step - clip_grad_norm_ + Adam with L2-coupled weight decay (reference train.py:84-85,260) over a table of tensors.
// Reductions go through per-block partial slots (deterministic), then one small finishing kernel.
#include "common.h"

constexpr int LT = 256;

__device__ __forceinline__ float block_reduce_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// part[blk*4 + {0,1,2,3}] = partial sums of (pre-tgt)^2, (post-tgt)^2, bce, guided
__global__ __launch_bounds__(LT) void loss_kernel(TacoLossArgs p, float* __restrict__ part) {
    __shared__ float sh[4];
    const long nm = (long)p.B * p.M * p.T, ns = (long)p.B * p.T, na = (long)p.B * p.T * p.L;
    const long stride = (long)gridDim.x * LT, i0 = (long)blockIdx.x * LT + threadIdx.x;
    float s_pre = 0.f, s_post = 0.f, s_bce = 0.f, s_ga = 0.f;
    const float inv_nm = 1.f / (float)nm;
    for (long i = i0; i < nm; i += stride) {
        const float t = p.target[i];
        const float d1 = p.pre[i] - t, d2 = p.post[i] - (p.post_target ? p.post_target[i] : t);
        s_pre += d1 * d1; s_post += d2 * d2;
        p.d_pre[i] = 4.f * d1 * inv_nm * p.gscale;            // d(2*mean)
        p.d_post[i] = 2.f * d2 * inv_nm * p.gscale;
    }
    const float inv_ns = 1.f / ((float)ns * (float)(p.M + 2));
    for (long i = i0; i < ns; i += stride) {
        const float x = p.stop[i], y = p.stop_target[i];
        const float w = 1.f + (p.pos_weight - 1.f) * y;
        const float sp = log1pf(__expf(-fabsf(x))) + fmaxf(-x, 0.f);          // softplus(-x)
        s_bce += (1.f - y) * x + w * sp;
        const float sig = 1.f / (1.f + __expf(-x));
        p.d_stop[i] = ((1.f - y) - w * (1.f - sig)) * inv_ns * p.gscale;
    }
    if (p.align) {
        const float inv2g2 = 1.f / (2.f * p.g * p.g);
        for (long i = i0; i < na; i += stride) {
            const int l = (int)(i % p.L); const long bt = i / p.L; const int t = (int)(bt % p.T), b = (int)(bt / p.T);
            const int fl = p.target_len[b], ll = p.text_len[b];
            float gw = 0.f;
            if (p.ga_on && t < fl && l < ll) {
                const float d = (float)l / (float)ll - (float)t / (float)fl;
                gw = (1.f - __expf(-d * d * inv2g2)) / ((float)fl * (float)p.B);
            }
            s_ga += gw * p.align[i];
            p.d_align[i] = gw * p.gscale;
        }
    }
    s_pre = block_reduce_256(s_pre, sh); s_post = block_reduce_256(s_post, sh);
    s_bce = block_reduce_256(s_bce, sh); s_ga = block_reduce_256(s_ga, sh);
    if (threadIdx.x == 0) { float* o = part + (long)blockIdx.x * 4; o[0] = s_pre; o[1] = s_post; o[2] = s_bce; o[3] = s_ga; }
}

// out[0..3] = mel_pre, mel_pos, stop_token, guided_att ; out[4] = their sum
__global__ void loss_finish_kernel(const float* __restrict__ part, int nblk, TacoLossArgs p, float* __restrict__ out) {
    __shared__ float sh[4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nblk; i += LT)
        for (int k = 0; k < 4; ++k) s[k] += part[(long)i * 4 + k];
    for (int k = 0; k < 4; ++k) s[k] = block_reduce_256(s[k], sh);
    if (threadIdx.x == 0) {
        const float nm = (float)p.B * p.M * p.T, ns = (float)p.B * p.T;
        out[0] = 2.f * s[0] / nm; out[1] = s[1] / nm; out[2] = s[2] / ns / (float)(p.M + 2); out[3] = s[3];
        out[4] = out[0] + out[1] + out[2] + out[3];
    }
}

MTTS_API int mtts_tacotron_loss(const TacoLossArgs* args, void* stream) {
    const TacoLossArgs& p = *args;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = p.nblk;
    MTTS_REQUIRE(nblk > 0 && p.partials && p.out, "mtts_tacotron_loss: workspace missing");
    hipLaunchKernelGGL(loss_kernel, dim3(nblk), dim3(LT), 0, s, p, p.partials);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LT), 0, s, p.partials, nblk, p, p.out);
    MTTS_CHECK_LAUNCH("tacotron_loss");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam.  Tensor table on the device: ptrs[4*n] = {param, grad, exp_avg, exp_avg_sq} per tensor,
// chunk table: chunk c covers elements [chunk_off[c], chunk_off[c] + chunk_len[c]) of tensor chunk_tensor[c].
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LT) void adam_sumsq_kernel(AdamArgs p) {
    __shared__ float sh[4];
    const int c = blockIdx.x;
    const float* g = (const float*)p.ptrs[4 * p.chunk_tensor[c] + 1] + p.chunk_off[c];
    const int n = p.chunk_len[c];
    float s = 0.f;
    // eight elements requested per round trip, squares added in index order (one per iteration left the 114 MB of gradients at 1 TB/s)
    for (int i = threadIdx.x; i < n; i += 8 * LT) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = g[min(i + u * LT, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i + u * LT < n) s += v[u] * v[u];
    }
    s = block_reduce_256(s, sh);
    if (threadIdx.x == 0) p.norm_partials[c] = s;
}

__global__ __launch_bounds__(LT) void adam_norm_kernel(AdamArgs p) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < p.nchunks; i += LT) s += p.norm_partials[i];
    s = block_reduce_256(s, sh);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        p.norm_out[0] = norm;
        float coef = p.max_norm > 0.f ? fminf(1.f, p.max_norm / (norm + 1e-6f)) : 1.f;      // clip coefficient (torch semantics)
        // guarded step: an invalid decode (persistent kernel gave up: error word set, NaN sentinels in its outputs) or a non-finite
        // gradient must not reach the weights - the update kernels return at once on a negative coefficient
        if (p.guard && (p.guard[1] != 0 || !(fabsf(norm) <= 3.0e38f))) coef = -1.f;
        p.norm_out[1] = coef;
    }
}

__global__ __launch_bounds__(LT) void adam_apply_kernel(AdamArgs p) {
    const int c = blockIdx.x;
    const int t = p.chunk_tensor[c];
    const long off = p.chunk_off[c];
    float* w = (float*)p.ptrs[4 * t] + off;
    float* g = (float*)p.ptrs[4 * t + 1] + off;
    float* m = (float*)p.ptrs[4 * t + 2] + off;
    float* v = (float*)p.ptrs[4 * t + 3] + off;
    const int n = p.chunk_len[c];
    const float coef = p.norm_out[1];
    if (coef < 0.f) return;                                   // guarded step skipped (adam_norm_kernel)
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * LT) {       // four elements (16 requests) per round trip
        float w4[4], g4[4], m4[4], v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * LT, n - 1);
            w4[u] = w[i]; g4[u] = g[i]; m4[u] = m[i]; v4[u] = v[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * LT;
            if (i >= n) break;
            const float wi = w4[u];
            const float gc = g4[u] * coef;
            g[i] = gc;                                            // clip_grad_norm_ leaves the clipped gradient behind
            const float gi = gc + p.weight_decay * wi;            // L2-coupled decay (torch.optim.Adam)
            const float mi = p.beta1 * m4[u] + (1.f - p.beta1) * gi;
            const float vi = p.beta2 * v4[u] + (1.f - p.beta2) * gi * gi;
            m[i] = mi; v[i] = vi;
            w[i] = wi - p.step_size * mi / (sqrtf(vi) * p.inv_sqrt_bc2 + p.eps);
        }
    }
}

MTTS_API int mtts_clip_adam_step(const AdamArgs* args, void* stream) {
    const AdamArgs& p = *args;
    hipStream_t s = (hipStream_t)stream;
    MTTS_REQUIRE(p.nchunks > 0, "mtts_clip_adam_step: empty chunk table");
    MTTS_REQUIRE(p.phase >= 0 && p.phase <= 2, "mtts_clip_adam_step: phase must be 0, 1 or 2");
    if (p.phase != 2) {
        hipLaunchKernelGGL(adam_sumsq_kernel, dim3(p.nchunks), dim3(LT), 0, s, p);
        hipLaunchKernelGGL(adam_norm_kernel, dim3(1), dim3(LT), 0, s, p);
    }
    if (p.phase != 1) hipLaunchKernelGGL(adam_apply_kernel, dim3(p.nchunks), dim3(LT), 0, s, p);
    MTTS_CHECK_LAUNCH("clip_adam_step");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Adversarial speaker classifier loss (reference ReversalClassifier.loss, modules/classifier.py:62-69): cross entropy of
// pred [B, L, S] against the utterance's speaker at every VALID character (padding = ignore_index), mean over the valid
// characters, times `scale`.  One wave per (b, l) row: log-sum-exp over S, per-row loss (already divided by the count) and
// the gradient (softmax - onehot) * scale / count in the same pass; the count is the sum of the lengths, recomputed per
// workgroup (B values).
// ---------------------------------------------------------------------------------------------------------------------
struct CeArgs { const float* pred; const int64_t* speakers; const int* lengths; float* row_loss; float* dpred; int B, L, S; float scale; };

__global__ __launch_bounds__(256) void masked_ce_kernel(CeArgs p) {
    __shared__ float cnt_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {
        float c = 0.f;
        for (int b = lane; b < p.B; b += 64) c += (float)min(p.lengths[b], p.L);
        c = wave_sum(c);
        if (lane == 0) cnt_s = c;
    }
    __syncthreads();
    const float inv = cnt_s > 0.f ? p.scale / cnt_s : 0.f;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)p.B * p.L) return;
    const int b = (int)(row / p.L), l = (int)(row - (long)b * p.L);
    const bool valid = l < p.lengths[b];
    const float* x = p.pred + row * p.S;
    float* dx = p.dpred + row * p.S;
    if (!valid) {
        for (int s = lane; s < p.S; s += 64) dx[s] = 0.f;
        if (lane == 0) p.row_loss[row] = 0.f;
        return;
    }
    float mx = -INFINITY;
    for (int s = lane; s < p.S; s += 64) mx = fmaxf(mx, x[s]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < p.S; s += 64) sum += __expf(x[s] - mx);
    sum = wave_sum(sum);
    const float lse = mx + __logf(sum);
    const int tgt = (int)p.speakers[b];
    for (int s = lane; s < p.S; s += 64) dx[s] = (__expf(x[s] - lse) - (s == tgt ? 1.f : 0.f)) * inv;
    if (lane == 0) p.row_loss[row] = (tgt >= 0 && tgt < p.S) ? (lse - x[tgt]) * inv : 0.f;
}

MTTS_API int mtts_masked_cross_entropy(const float* pred, const int64_t* speakers, const int* lengths, float* row_loss, float* dpred,
                                       int B, int L, int S, float scale, void* stream) {
    MTTS_REQUIRE(B > 0 && L > 0 && S > 0, "mtts_masked_cross_entropy: empty input");
    CeArgs p{pred, speakers, lengths, row_loss, dpred, B, L, S, scale};
    hipLaunchKernelGGL(masked_ce_kernel, dim3((unsigned)(((long)B * L + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    MTTS_CHECK_LAUNCH("masked_ce_kernel");
    return 0;
}
