// joint.hip -- the element-wise / row-wise pieces of the Transducer's prediction and joint networks
// (/root/reference/speech/models/transducer_model.py:54-78, Transducer.decode):
//   embedding lookup (:59) and its gradient, the broadcast joint  relu(fc1(x)[b,t,:] + fc1(y)[b,u,:])  (:72-74) and its
//   two gradient reductions, log_softmax over the class axis (:76) and its gradient.
// The matrix products around them (fc1, fc2 and their gradients) are sa_gemm_f32; the prediction GRU is
// sa_gru_stack_*.  All tensors fp32, contiguous, DEVICE.
#include "common.h"

namespace {

// out[i, :] = table[idx[i], :]
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const float* __restrict__ table,
                                                            const long long* __restrict__ idx,
                                                            float* __restrict__ out, int E) {
    const int i = blockIdx.x;
    const float* src = table + (long)idx[i] * E;
    float* dst = out + (long)i * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) dst[e] = src[e];
}

// dtable[v, :] = sum over i with idx[i] == v of dout[i, :]   (ascending i: deterministic)
// One block per table row.  The n indices are scanned 256 at a time: a wave ballot + a 4-entry prefix compacts the
// matching positions, in order, into LDS, and only those rows are read (n / V of them on average, not n).
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dout,
                                                            const long long* __restrict__ idx,
                                                            float* __restrict__ dtable, int n, int E) {
    __shared__ int list[256];
    __shared__ int wcount[4];
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kMaxE = 4;  // columns per thread held in registers (E <= 1024); wider tables loop below
    float acc[kMaxE];
#pragma unroll
    for (int c = 0; c < kMaxE; ++c) acc[c] = 0.f;
    for (int e0 = 0; e0 < E; e0 += 256 * kMaxE) {
        for (int base = 0; base < n; base += 256) {
            const int i = base + tid;
            const bool hit = i < n && idx[i] == v;
            const unsigned long long mask = __ballot(hit);
            if (lane == 0) wcount[wave] = __popcll(mask);
            __syncthreads();
            int off = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w < wave) off += wcount[w];
                total += wcount[w];
            }
            if (hit) list[off + __popcll(mask & ((1ull << lane) - 1ull))] = i;
            __syncthreads();
            for (int k = 0; k < total; ++k) {
                const float* row = dout + (long)list[k] * E + e0 + tid;
#pragma unroll
                for (int c = 0; c < kMaxE; ++c)
                    if (e0 + tid + 256 * c < E) acc[c] += row[256 * c];
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < kMaxE; ++c) {
            if (e0 + tid + 256 * c < E) dtable[(long)v * E + e0 + tid + 256 * c] = acc[c];
            acc[c] = 0.f;
        }
    }
}

// z[b,t,u,:] = relu(xa[b,t,:] + ya[b,u,:]);  grid (U1, T, B)
__global__ __launch_bounds__(128) void joint_relu_fwd_kernel(const float* __restrict__ xa, const float* __restrict__ ya,
                                                             float* __restrict__ z, int T, int U1, int H) {
    const int u = blockIdx.x, t = blockIdx.y, b = blockIdx.z;
    const float* x = xa + ((long)b * T + t) * H;
    const float* y = ya + ((long)b * U1 + u) * H;
    float* o = z + (((long)b * T + t) * U1 + u) * H;
    if ((H & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* y4 = reinterpret_cast<const float4*>(y);
        float4* o4 = reinterpret_cast<float4*>(o);
        for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
            const float4 a = x4[i], c = y4[i];
            o4[i] = make_float4(fmaxf(a.x + c.x, 0.f), fmaxf(a.y + c.y, 0.f), fmaxf(a.z + c.z, 0.f),
                                fmaxf(a.w + c.w, 0.f));
        }
    } else {
        for (int i = threadIdx.x; i < H; i += blockDim.x) o[i] = fmaxf(x[i] + y[i], 0.f);
    }
}

// Gradient of the joint: with m = [xa + ya > 0] (recomputed from two L2-resident rows instead of re-reading z),
//   AXIS 0: dxa[b,t,:] = sum_u dz[b,t,u,:] m      grid (T, B)
//   AXIS 1: dya[b,u,:] = sum_t dz[b,t,u,:] m      grid (U1, B)
// one thread per hidden unit, serial over the reduced axis (fixed order), four loads in flight.
template <int AXIS>
__global__ __launch_bounds__(256) void joint_relu_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ xa,
                                                             const float* __restrict__ ya, float* __restrict__ out,
                                                             int T, int U1, int H) {
    const int j = blockIdx.x, b = blockIdx.y;  // j = t (AXIS 0) or u (AXIS 1)
    const int n = AXIS == 0 ? U1 : T;
    const float* own_row = AXIS == 0 ? xa + ((long)b * T + j) * H : ya + ((long)b * U1 + j) * H;
    const float* oth = AXIS == 0 ? ya + (long)b * U1 * H : xa + (long)b * T * H;  // row i of the other operand
    const long dz_step = AXIS == 0 ? (long)H : (long)U1 * H;
    const float* dzp = dz + (AXIS == 0 ? ((long)b * T + j) * U1 * H : ((long)b * T * U1 + j) * H);
    float* o = out + (AXIS == 0 ? ((long)b * T + j) : ((long)b * U1 + j)) * H;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        const float own = own_row[h];
        float acc = 0.f;
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            float g[4], p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g[k] = dzp[(long)(i + k) * dz_step + h];
                p[k] = oth[(long)(i + k) * H + h];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += own + p[k] > 0.f ? g[k] : 0.f;
        }
        for (; i < n; ++i) acc += own + oth[(long)i * H + h] > 0.f ? dzp[(long)i * dz_step + h] : 0.f;
        o[h] = acc;
    }
}

// Row-wise log-softmax over K classes, one wave per row (K <= 64: one class per lane; larger K: strided).
//   FWD:  y = x - max - log sum exp(x - max)
//   BWD:  dx = dy - exp(y) * sum(dy)
template <bool BWD>
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ a, const float* __restrict__ y_in,
                                                          float* __restrict__ out, long rows, int K) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* ar = a + row * K;
    float* orow = out + row * K;
    if (!BWD) {
        float m = -3.0e38f;
        for (int k = lane; k < K; k += 64) m = fmaxf(m, ar[k]);
        m = sa_wave_max_dpp(m);
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += sa_exp2((ar[k] - m) * SA_LOG2E);
        s = sa_wave_sum_dpp(s);
        const float lz = m + sa_log2(s) * SA_LN2;
        for (int k = lane; k < K; k += 64) orow[k] = ar[k] - lz;
    } else {
        const float* yr = y_in + row * K;
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += ar[k];
        s = sa_wave_sum_dpp(s);
        for (int k = lane; k < K; k += 64) orow[k] = ar[k] - sa_exp2(yr[k] * SA_LOG2E) * s;
    }
}

}  // namespace

extern "C" ctcStatus_t sa_embedding_fwd(const float* table, const long long* idx, float* out, int n, int E,
                                        void* stream) {
    SA_CLEAR_ERR();
    if (!table || !idx || !out || n <= 0 || E <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table, idx, out, E);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_embedding_bwd(const float* dout, const long long* idx, float* dtable, int n, int E, int V,
                                        void* stream) {
    SA_CLEAR_ERR();
    if (!dout || !idx || !dtable || n <= 0 || E <= 0 || V <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, dout, idx, dtable, n, E);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_joint_relu_fwd(const float* xa, const float* ya, float* z, int B, int T, int U1, int H,
                                         void* stream) {
    SA_CLEAR_ERR();
    if (!xa || !ya || !z || B <= 0 || T <= 0 || U1 <= 0 || H <= 0 || T > 65535 || B > 65535)
        return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(joint_relu_fwd_kernel, dim3(U1, T, B), dim3(128), 0, (hipStream_t)stream, xa, ya, z, T, U1, H);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_joint_relu_bwd(const float* dz, const float* xa, const float* ya, float* dxa, float* dya,
                                         int B, int T, int U1, int H, void* stream) {
    SA_CLEAR_ERR();
    if (!dz || !xa || !ya || !dxa || !dya || B <= 0 || T <= 0 || U1 <= 0 || H <= 0 || B > 65535)
        return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(joint_relu_bwd_kernel<0>, dim3(T, B), dim3(256), 0, (hipStream_t)stream, dz, xa, ya, dxa, T, U1,
                       H);
    hipLaunchKernelGGL(joint_relu_bwd_kernel<1>, dim3(U1, B), dim3(256), 0, (hipStream_t)stream, dz, xa, ya, dya, T,
                       U1, H);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_log_softmax_fwd(const float* x, float* y, long rows, int K, void* stream) {
    SA_CLEAR_ERR();
    if (!x || !y || rows <= 0 || K <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(log_softmax_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x, nullptr, y, rows, K);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_log_softmax_bwd(const float* dy, const float* y, float* dx, long rows, int K, void* stream) {
    SA_CLEAR_ERR();
    if (!dy || !y || !dx || rows <= 0 || K <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(log_softmax_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       dy, y, dx, rows, K);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
