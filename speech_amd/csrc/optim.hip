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
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t beg = (size_t)blockIdx.x * per;
    const size_t end = beg + per < n ? beg + per : n;
    float acc = 0.f;
    for (size_t i = beg + threadIdx.x; i < end; i += 256) {
        const float v = g[i] * scale;
        acc += v * v;
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
