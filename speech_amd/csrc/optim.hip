// optim.hip -- the optimiser step of the reference's training loop on ONE flat fp32 buffer
// (/root/reference/train.py:32 clip_grad_norm(params, 200); train.py:35,95-97 SGD(lr, momentum)).
// The reference issues a norm kernel and an update kernel per parameter tensor; here all parameters and gradients
// are views into two flat buffers (which is also the single RCCL all-reduce message), so the step is
//   pass 1: per-block sums of squares (fixed-order tree: deterministic)
//   pass 2: every block re-reduces the <= 1024 partials, derives the clip coefficient, updates its slice.
#include "common.h"

namespace {

constexpr int kMaxPartials = 1024;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = sa_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        red[0] = t;
    }
    __syncthreads();
    t = red[0];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float scale,
                                                    float* __restrict__ partials) {
    __shared__ float red[4];
    // a block's share is a whole number of 1024-element trips (256 threads x 16 bytes), so that an aligned buffer is read
    // with 16-byte loads, four trips in flight: one 4-byte load per trip with the add behind it was a chain of 32 memory
    // round trips (17 us for 27 MB; round 5)
    const size_t per = ((n + gridDim.x - 1) / gridDim.x + 1023) / 1024 * 1024;
    const size_t beg = (size_t)blockIdx.x * per;
    const size_t end = beg + per < n ? beg + per : n;
    float acc = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        size_t i = beg + 4 * threadIdx.x;
        for (; i + 3 * 1024 + 3 < end; i += 4 * 1024) {
            const float4 v0 = *reinterpret_cast<const float4*>(g + i), v1 = *reinterpret_cast<const float4*>(g + i + 1024);
            const float4 v2 = *reinterpret_cast<const float4*>(g + i + 2048), v3 = *reinterpret_cast<const float4*>(g + i + 3072);
            a0 += (v0.x * scale) * (v0.x * scale) + (v0.y * scale) * (v0.y * scale) + (v0.z * scale) * (v0.z * scale) + (v0.w * scale) * (v0.w * scale);
            a1 += (v1.x * scale) * (v1.x * scale) + (v1.y * scale) * (v1.y * scale) + (v1.z * scale) * (v1.z * scale) + (v1.w * scale) * (v1.w * scale);
            a2 += (v2.x * scale) * (v2.x * scale) + (v2.y * scale) * (v2.y * scale) + (v2.z * scale) * (v2.z * scale) + (v2.w * scale) * (v2.w * scale);
            a3 += (v3.x * scale) * (v3.x * scale) + (v3.y * scale) * (v3.y * scale) + (v3.z * scale) * (v3.z * scale) + (v3.w * scale) * (v3.w * scale);
        }
        for (; i < end; i += 1024)  // the last trips, element by element at the ragged end
            for (int j = 0; j < 4; ++j)
                if (i + j < end) { const float v = g[i + j] * scale; a0 += v * v; }
        acc = (a0 + a1) + (a2 + a3);
    } else {
        for (size_t i = beg + threadIdx.x; i < end; i += 256) {
            const float v = g[i] * scale;
            acc += v * v;
        }
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void clip_sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ mom, size_t n, float lr, float momentum,
                                                       float max_norm, float scale, const float* __restrict__ partials,
                                                       int nparts, float* __restrict__ norm_out,
                                                       const float* __restrict__ skip_flag) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[i];
    const float total = sqrtf(block_sum(acc, red));
    // the health gate (see sa_gru_health_flag): a non-zero flag -- this rank's, or any rank's after the gradient
    // all-reduce summed them -- means some gradient of this step is garbage: leave the parameters alone and say so
    const bool skip = skip_flag && *skip_flag != 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = skip ? -fmaxf(total, 1e-30f) : total;  // < 0 also for a zero or NaN norm
    if (skip) return;
    float coef = max_norm / (total + 1e-6f);   // torch.nn.utils.clip_grad_norm_
    coef = coef < 1.0f ? coef : 1.0f;
    const float gs = scale * coef;
    if (!mom && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(p)) & 15) == 0) {
        // plain SGD on aligned buffers: 16-byte accesses, two trips in flight (the same arithmetic per element)
        const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + stride < n4; i += 2 * stride) {
            const float4 g0 = reinterpret_cast<const float4*>(g)[i], g1 = reinterpret_cast<const float4*>(g)[i + stride];
            float4 p0 = reinterpret_cast<float4*>(p)[i], p1 = reinterpret_cast<float4*>(p)[i + stride];
            p0.x -= lr * (g0.x * gs); p0.y -= lr * (g0.y * gs); p0.z -= lr * (g0.z * gs); p0.w -= lr * (g0.w * gs);
            p1.x -= lr * (g1.x * gs); p1.y -= lr * (g1.y * gs); p1.z -= lr * (g1.z * gs); p1.w -= lr * (g1.w * gs);
            reinterpret_cast<float4*>(p)[i] = p0;
            reinterpret_cast<float4*>(p)[i + stride] = p1;
        }
        for (; i < n4; i += stride) {
            const float4 g0 = reinterpret_cast<const float4*>(g)[i];
            float4 p0 = reinterpret_cast<float4*>(p)[i];
            p0.x -= lr * (g0.x * gs); p0.y -= lr * (g0.y * gs); p0.z -= lr * (g0.z * gs); p0.w -= lr * (g0.w * gs);
            reinterpret_cast<float4*>(p)[i] = p0;
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const size_t e = 4 * n4 + threadIdx.x; p[e] -= lr * (g[e] * gs); }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float d = g[i] * gs;
        if (mom) {
            d = momentum * mom[i] + d;
            mom[i] = d;
        }
        p[i] -= lr * d;
    }
}

}  // namespace

extern "C" size_t sa_sgd_workspace_bytes(size_t n) { (void)n; return kMaxPartials * sizeof(float); }

extern "C" ctcStatus_t sa_clip_sgd_step(float* params, float* grads, float* momentum_buf, size_t n, float lr,
                                        float momentum, float max_norm, float grad_scale, float* d_norm_out,
                                        const float* d_skip_flag, void* workspace, size_t workspace_bytes,
                                        void* stream_) {
    SA_CLEAR_ERR();
    if (!params || !grads || !workspace || workspace_bytes < sa_sgd_workspace_bytes(n)) return CTC_STATUS_INVALID_VALUE;
    if (n == 0) return CTC_STATUS_SUCCESS;
    hipStream_t stream = (hipStream_t)stream_;
    int nparts = (int)((n + 8191) / 8192);
    if (nparts > kMaxPartials) nparts = kMaxPartials;
    float* partials = (float*)workspace;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nparts), dim3(256), 0, stream, grads, n, grad_scale, partials);
    SA_CHECK_LAUNCH();
    int grid = (int)((n + 1023) / 1024);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(clip_sgd_kernel, dim3(grid), dim3(256), 0, stream, params, grads,
                       momentum != 0.f ? momentum_buf : nullptr, n, lr, momentum, max_norm, grad_scale, partials,
                       nparts, d_norm_out, d_skip_flag);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
