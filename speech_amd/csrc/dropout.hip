// dropout.hip -- stand-alone forms of the dropout mask (dropout.h): what the kernels with a fused mask (conv.hip's
// epilogue, gru.hip's fused recurrence kernels) compute in place, as plain element-wise launches for the paths that
// have no fused form (the generic im2col convolution, the chunked / step-kernel GRU paths, bidirectional stacks) and
// for the tests (sa_dropout_mask_f32 materialises a mask so that the CPU restatement can be run on the SAME mask).
// Reference: nn.Dropout / nn.GRU(dropout=p), /root/reference/speech/models/model.py:25-27,35-39.
#include "common.h"
#include "dropout.h"
#include "internal.h"

namespace {

// out[i] = (in ? in[i] : 1) * factor(idx0 + i), i < n.  A thread owns one aligned group of four mask words.
__global__ __launch_bounds__(256) void dropout_flat_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           unsigned long long n, unsigned long long idx0, SaDrop d,
                                                           unsigned stream) {
    const unsigned long long q0 = idx0 >> 2, nq = ((idx0 + n + 3) >> 2) - q0;
    for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < nq;
         q += (unsigned long long)gridDim.x * 256) {
        const unsigned long long base = (q0 + q) << 2;
        uint32_t w[4];
        sa_drop_words(d, stream, base, w);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long idx = base + e;
            if (idx < idx0 || idx >= idx0 + n) continue;
            const float f = w[e] >= d.thresh ? d.scale : 0.f;
            out[idx - idx0] = in ? in[idx - idx0] * f : f;
        }
    }
}

// In place on a conv output stored with the caller's strides: y[b, c, t, f] *= factor(((b O + c) T' + t) F' + f).
__global__ __launch_bounds__(256) void dropout_nchw_strided_kernel(float* __restrict__ y, int B, int O, int To, int Fo,
                                                                   long ys_b, long ys_c, long ys_t, SaDrop d,
                                                                   unsigned stream) {
    const long total = (long)B * O * To * Fo;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int fo = (int)(idx % Fo);
        const long r = idx / Fo;
        const int t = (int)(r % To);
        const long bc = r / To;
        const int c = (int)(bc % O), b = (int)(bc / O);
        float* p = y + (long)b * ys_b + (long)c * ys_c + (long)t * ys_t + fo;
        *p *= sa_drop_factor(d, stream, (uint64_t)idx);
    }
}

int grid_for_q(unsigned long long nq) {
    unsigned long long g = (nq + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

// internal (conv.hip's generic path)
ctcStatus_t sa_dropout_nchw_strided_impl(float* y, int B, int O, int To, int Fo, long ys_b, long ys_c, long ys_t,
                                         const SaDrop& d, unsigned stream_id, hipStream_t stream) {
    if (!d.on()) return CTC_STATUS_SUCCESS;
    hipLaunchKernelGGL(dropout_nchw_strided_kernel, dim3(grid_for_q((unsigned long long)B * O * To * Fo)), dim3(256), 0,
                       stream, y, B, O, To, Fo, ys_b, ys_c, ys_t, d, stream_id);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

ctcStatus_t sa_dropout_apply_impl(const float* in, float* out, size_t n, size_t idx0, const SaDrop& d,
                                  unsigned stream_id, hipStream_t stream) {
    if (n == 0) return CTC_STATUS_SUCCESS;
    hipLaunchKernelGGL(dropout_flat_kernel, dim3(grid_for_q((n + 3) / 4 + 1)), dim3(256), 0, stream, in, out,
                       (unsigned long long)n, (unsigned long long)idx0, d, stream_id);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_dropout_mask_f32(float* d_out, size_t n, size_t idx0, float p, unsigned long long seed,
                                           unsigned int mask_stream, void* stream_) {
    SA_CLEAR_ERR();
    if (!d_out || !sa_drop_valid(p)) return CTC_STATUS_INVALID_VALUE;
    SaDrop d = sa_drop_make(p, seed);
    return sa_dropout_apply_impl(nullptr, d_out, n, idx0, d, mask_stream, (hipStream_t)stream_);
}

extern "C" ctcStatus_t sa_dropout_apply_f32(const float* d_in, float* d_out, size_t n, size_t idx0, float p,
                                            unsigned long long seed, unsigned int mask_stream, void* stream_) {
    SA_CLEAR_ERR();
    if (!d_in || !d_out || !sa_drop_valid(p)) return CTC_STATUS_INVALID_VALUE;
    SaDrop d = sa_drop_make(p, seed);
    return sa_dropout_apply_impl(d_in, d_out, n, idx0, d, mask_stream, (hipStream_t)stream_);
}
