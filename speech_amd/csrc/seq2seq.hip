// seq2seq.hip -- the per-token pieces of the attention decoder of /root/reference/speech/models/seq2seq.py:
//   GRUCell gate math (:20-21, :97; the two projections around it are sa_gemm_f32),
//   NNAttention (:331-360): score_t = w . relu(eh_t + ox + conv1d(ax_prev)_t) + b, optional log(T) temperature (:351-353),
//   softmax over time, context sx = sum_t ax_t eh_t -- the score network is parallel over (utterance, 16-step time
//   chunk) in both directions; only the softmax / context and the final fold run one workgroup per utterance,
//   softmax cross-entropy over the output classes (:59-63) with the gradient produced in the same pass, row argmax
//   (scheduled sampling :93 and greedy decode :157).
// All tensors fp32, contiguous, DEVICE.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// stash (B, 4H): r | z | n | gh_n (the hidden-side pre-activation of the candidate gate, needed by the backward)
__global__ __launch_bounds__(256) void grucell_gates_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                                const float* __restrict__ h_prev, float* __restrict__ h_out,
                                                                float* __restrict__ stash, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* a = gi + (long)b * 3 * H;
    const float* c = gh + (long)b * 3 * H;
    const float r = sigmoidf_(a[j] + c[j]);
    const float z = sigmoidf_(a[H + j] + c[H + j]);
    const float ghn = c[2 * H + j];
    const float n = tanhf(a[2 * H + j] + r * ghn);
    h_out[i] = (1.0f - z) * n + z * h_prev[i];
    if (stash) {
        float* s = stash + (long)b * 4 * H;
        s[j] = r; s[H + j] = z; s[2 * H + j] = n; s[3 * H + j] = ghn;
    }
}

__global__ __launch_bounds__(256) void grucell_gates_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ stash,
                                                                const float* __restrict__ h_prev, float* __restrict__ dgi,
                                                                float* __restrict__ dgh, float* __restrict__ dh_prev,
                                                                int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* s = stash + (long)b * 4 * H;
    const float r = s[j], z = s[H + j], n = s[2 * H + j], ghn = s[3 * H + j];
    const float g = dh[i];
    const float dn = g * (1.0f - z) * (1.0f - n * n);       // through tanh
    const float dz = g * (h_prev[i] - n) * z * (1.0f - z);  // through sigmoid
    const float dr = dn * ghn * r * (1.0f - r);
    float* a = dgi + (long)b * 3 * H;
    float* c = dgh + (long)b * 3 * H;
    a[j] = dr; a[H + j] = dz; a[2 * H + j] = dn;
    c[j] = dr; c[H + j] = dz; c[2 * H + j] = dn * r;
    dh_prev[i] = g * z;
}

struct AttArgs {
    const float* eh;       // (B, T, H)
    const float* ox;       // (B, H)
    const float* ax_prev;  // (B, T) or NULL (first token: no location term at all, seq2seq.py:343)
    const float* conv_w;   // (H, KS)
    const float* conv_b;   // (H)
    const float* nn_w;     // (H)
    const float* nn_b;     // (1)
    float scale;           // log(T) if log_t else 1
    int B, T, H, KS;
};

constexpr int kAttTB = 16;  // time steps per workgroup: grid (ceil(T / 16), B) fills the chip where one workgroup per
                            // utterance (B = 16) would use 16 of 256 CUs

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {  // 256 threads, red[4]
    v = is_max ? sa_wave_max_dpp(v) : sa_wave_sum_dpp(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float a = red[0], b = red[1], c = red[2], d = red[3];
    return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

// LDS staging of a time chunk [t0, t0 + kAttTB): axp[kAttTB + KS - 1] = ax_prev[t0 - pad ..] zero-padded, cw[H * KS]
__device__ __forceinline__ void att_stage(const AttArgs& A, int b, int t0, float* axp, float* cw) {
    const int pad = (A.KS - 1) / 2;
    for (int i = threadIdx.x; i < kAttTB + A.KS - 1; i += blockDim.x) {
        const int t = t0 + i - pad;
        axp[i] = (A.ax_prev && t >= 0 && t < A.T) ? A.ax_prev[(long)b * A.T + t] : 0.f;
    }
    for (int i = threadIdx.x; i < A.H * A.KS; i += blockDim.x) cw[i] = A.conv_w[i];
}

// pre-activation of the score network at (t0 + tl, h)
__device__ __forceinline__ float att_pre(const AttArgs& A, const float* ehb, const float* oxb, const float* axp,
                                         const float* cw, int t0, int tl, int h) {
    float v = ehb[(long)(t0 + tl) * A.H + h] + oxb[h];
    if (A.ax_prev) {
        float c = A.conv_b[h];
        for (int k = 0; k < A.KS; ++k) c += cw[h * A.KS + k] * axp[tl + k];
        v += c;
    }
    return v;
}

// ---- forward, stage 1: scores.  grid (ceil(T / 16), B), 256 threads: a wave per time step (4 per wave).
// dynamic LDS: axp[kAttTB + KS - 1] | cw[H * KS]
__global__ __launch_bounds__(256) void attention_score_kernel(AttArgs A, float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* axp = reinterpret_cast<float*>(smem_raw);
    float* cw = axp + kAttTB + A.KS - 1;
    const int b = blockIdx.y, t0 = blockIdx.x * kAttTB, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ehb = A.eh + (long)b * A.T * A.H;
    const float* oxb = A.ox + (long)b * A.H;
    att_stage(A, b, t0, axp, cw);
    __syncthreads();
    const float nb = A.nn_b[0];
    for (int tl = wave; tl < kAttTB && t0 + tl < A.T; tl += 4) {
        float v = 0.f;
        for (int h = lane; h < A.H; h += 64) v += fmaxf(att_pre(A, ehb, oxb, axp, cw, t0, tl, h), 0.f) * A.nn_w[h];
        v = sa_wave_sum_dpp(v);
        if (lane == 0) score[(long)b * A.T + t0 + tl] = (v + nb) * A.scale;
    }
}

// ---- forward, stage 2: softmax over time and the context.  grid B, 256 threads.  dynamic LDS: a[T] | red[4] | part[4][H]
__global__ __launch_bounds__(256) void attention_context_kernel(AttArgs A, const float* __restrict__ score,
                                                                float* __restrict__ ax, float* __restrict__ sx) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* a = reinterpret_cast<float*>(smem_raw);
    float* red = a + A.T;
    float* part = red + 4;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ehb = A.eh + (long)b * A.T * A.H;
    float m = -3.0e38f;
    for (int t = threadIdx.x; t < A.T; t += 256) {
        a[t] = score[(long)b * A.T + t];
        m = fmaxf(m, a[t]);
    }
    m = block_reduce(m, red, true);
    float s = 0.f;
    for (int t = threadIdx.x; t < A.T; t += 256) {
        const float e = __expf(a[t] - m);
        a[t] = e;
        s += e;
    }
    s = block_reduce(s, red, false);
    const float inv = 1.0f / s;
    for (int t = threadIdx.x; t < A.T; t += 256) {
        const float v = a[t] * inv;
        a[t] = v;
        ax[(long)b * A.T + t] = v;
    }
    __syncthreads();
    // context: each wave sums a quarter of the time axis for every hidden unit, then the quarters are combined
    const int tq = (A.T + 3) / 4, tb = wave * tq, te = min(A.T, tb + tq);
    for (int h = lane; h < A.H; h += 64) {
        float acc = 0.f;
        for (int t = tb; t < te; ++t) acc += a[t] * ehb[(long)t * A.H + h];
        part[wave * A.H + h] = acc;
    }
    __syncthreads();
    for (int h = threadIdx.x; h < A.H; h += 256)
        sx[(long)b * A.H + h] = (part[h] + part[A.H + h]) + (part[2 * A.H + h] + part[3 * A.H + h]);
}

struct AttBwd {
    const float* ax;         // (B, T) this step's alignment
    const float* d_sx;       // (B, H)
    const float* d_ax_next;  // (B, T) or NULL: gradient arriving through the next token's location term
    float* d_eh;             // (B, T, H)  +=
    float* d_ox;             // (B, H)     =
    float* d_ax_prev;        // (B, T)     =   (unused when ax_prev is NULL)
    float* g_conv_w;         // (B, H, KS) +=  per-utterance partials, reduced over B by the caller
    float* g_conv_b;         // (B, H)     +=
    float* g_nn_w;           // (B, H)     +=
    float* g_nn_b;           // (B)        +=
    float* dpax;             // workspace (B, T): d ax, then d score
    float* part;             // workspace (B, nchunk, H, 2 + KS): per-chunk sums of d_pre, d_pre-weighted terms
    float* q;                // workspace (B, T, KS): sum_h d_pre[t,h] conv_w[h,k]
    int nchunk;
};

// ---- backward, stage 1: d ax[t] = (from the next token) + d_sx . eh[t].  grid (ceil(T / 16), B), a wave per time step
__global__ __launch_bounds__(256) void attention_bwd_dax_kernel(AttArgs A, AttBwd G) {
    const int b = blockIdx.y, t0 = blockIdx.x * kAttTB, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ehb = A.eh + (long)b * A.T * A.H;
    const float* dsx = G.d_sx + (long)b * A.H;
    for (int tl = wave; tl < kAttTB && t0 + tl < A.T; tl += 4) {
        const int t = t0 + tl;
        float v = 0.f;
        for (int h = lane; h < A.H; h += 64) v += dsx[h] * ehb[(long)t * A.H + h];
        v = sa_wave_sum_dpp(v);
        if (lane == 0) G.dpax[(long)b * A.T + t] = v + (G.d_ax_next ? G.d_ax_next[(long)b * A.T + t] : 0.f);
    }
}

// ---- backward, stage 2: through the softmax and the temperature (in place: dpax becomes d score).  grid B
__global__ __launch_bounds__(256) void attention_bwd_softmax_kernel(AttArgs A, AttBwd G) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* ax = G.ax + (long)b * A.T;
    float* d = G.dpax + (long)b * A.T;
    float s = 0.f;
    for (int t = threadIdx.x; t < A.T; t += 256) s += ax[t] * d[t];
    s = block_reduce(s, red, false);
    float sb = 0.f;
    for (int t = threadIdx.x; t < A.T; t += 256) {
        const float v = ax[t] * (d[t] - s) * A.scale;
        d[t] = v;
        sb += v;
    }
    sb = block_reduce(sb, red, false);
    if (threadIdx.x == 0) G.g_nn_b[b] += sb;
}

// ---- backward, stage 3: the score network over one time chunk.  grid (nchunk, B), 256 threads.
// phase 1, a thread per hidden unit, serial over the chunk's 16 steps: d_pre, d_eh, the chunk's per-unit partial sums;
// phase 2, a thread per (step, tap): q[t][k] = sum_h d_pre[t,h] cw[h,k] out of an LDS tile of d_pre.
// dynamic LDS: axp | cw | dps[kAttTB] | axs[kAttTB] | dp_tile[kAttTB][H]
__global__ __launch_bounds__(256) void attention_bwd_main_kernel(AttArgs A, AttBwd G) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* axp = reinterpret_cast<float*>(smem_raw);
    float* cw = axp + kAttTB + A.KS - 1;
    float* dps = cw + A.H * A.KS;
    float* axs = dps + kAttTB;
    float* dp_tile = axs + kAttTB;
    const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * kAttTB;
    const int nt = min(kAttTB, A.T - t0);
    const float* ehb = A.eh + (long)b * A.T * A.H;
    const float* oxb = A.ox + (long)b * A.H;
    const float* dsx = G.d_sx + (long)b * A.H;
    float* dehb = G.d_eh + (long)b * A.T * A.H;
    att_stage(A, b, t0, axp, cw);
    if ((int)threadIdx.x < kAttTB) {
        const bool ok = (int)threadIdx.x < nt;
        dps[threadIdx.x] = ok ? G.dpax[(long)b * A.T + t0 + threadIdx.x] : 0.f;
        axs[threadIdx.x] = ok ? G.ax[(long)b * A.T + t0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
    const int W = 2 + A.KS;
    for (int h = threadIdx.x; h < A.H; h += 256) {
        const float w = A.nn_w[h], dsxh = dsx[h];
        float a_ox = 0.f, a_nw = 0.f;
        float a_cw[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a_cw[k] = 0.f;
        for (int tl = 0; tl < nt; ++tl) {
            const float pre = att_pre(A, ehb, oxb, axp, cw, t0, tl, h);
            const float dp = pre > 0.f ? dps[tl] * w : 0.f;
            a_ox += dp;
            a_nw += dps[tl] * fmaxf(pre, 0.f);
            if (A.ax_prev) {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < A.KS) a_cw[k] += dp * axp[tl + k];
            }
            dehb[(long)(t0 + tl) * A.H + h] += axs[tl] * dsxh + dp;
            dp_tile[tl * A.H + h] = dp;
        }
        float* p = G.part + (((long)b * G.nchunk + chunk) * A.H + h) * W;
        p[0] = a_ox;
        p[1] = a_nw;
        for (int k = 0; k < A.KS; ++k) p[2 + k] = a_cw[k];
    }
    if (!A.ax_prev) return;
    __syncthreads();
    for (int i = threadIdx.x; i < nt * A.KS; i += 256) {
        const int tl = i / A.KS, k = i - tl * A.KS;
        float acc = 0.f;
        for (int h = 0; h < A.H; ++h) acc += dp_tile[tl * A.H + h] * cw[h * A.KS + k];
        G.q[((long)b * A.T + t0 + tl) * A.KS + k] = acc;
    }
}

// ---- backward, stage 4: fold the chunks.  grid B, 256 threads
__global__ __launch_bounds__(256) void attention_bwd_fold_kernel(AttArgs A, AttBwd G) {
    const int b = blockIdx.x;
    const int W = 2 + A.KS;
    for (int h = threadIdx.x; h < A.H; h += 256) {
        float acc[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) acc[k] = 0.f;
        for (int c = 0; c < G.nchunk; ++c) {
            const float* p = G.part + (((long)b * G.nchunk + c) * A.H + h) * W;
#pragma unroll
            for (int k = 0; k < 18; ++k)
                if (k < W) acc[k] += p[k];
        }
        G.d_ox[(long)b * A.H + h] = acc[0];
        G.g_nn_w[(long)b * A.H + h] += acc[1];
        if (A.ax_prev) {
            G.g_conv_b[(long)b * A.H + h] += acc[0];
            for (int k = 0; k < A.KS; ++k) G.g_conv_w[((long)b * A.H + h) * A.KS + k] += acc[2 + k];
        }
    }
    if (!A.ax_prev) return;
    const int pad = (A.KS - 1) / 2;
    for (int tp = threadIdx.x; tp < A.T; tp += 256) {
        float acc = 0.f;
        for (int k = 0; k < A.KS; ++k) {
            const int t = tp + pad - k;
            if (t >= 0 && t < A.T) acc += G.q[((long)b * A.T + t) * A.KS + k];
        }
        G.d_ax_prev[(long)b * A.T + tp] = acc;
    }
}

// softmax cross-entropy, one wave per row: loss_rows[i] = lse(x_i) - x_i[target];  dlogits = (softmax - onehot) * scale
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ x, const long long* __restrict__ tgt,
                                                           float scale, float* __restrict__ loss_rows,
                                                           float* __restrict__ dx, long rows, int K) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * K;
    float m = -3.0e38f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, xr[k]);
    m = sa_wave_max_dpp(m);
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += __expf(xr[k] - m);
    s = sa_wave_sum_dpp(s);
    const int t = (int)tgt[row];
    if (lane == 0) loss_rows[row] = m + __logf(s) - xr[t];
    if (dx) {
        const float inv = 1.0f / s;
        for (int k = lane; k < K; k += 64) dx[row * K + k] = (__expf(xr[k] - m) * inv - (k == t ? 1.f : 0.f)) * scale;
    }
}

// first index of the row maximum, one wave per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long long* __restrict__ out,
                                                          long rows, int K) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * K;
    float best = -3.0e38f;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += 64)
        if (xr[k] > best) { best = xr[k]; bi = k; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[row] = bi;
}

}  // namespace

extern "C" ctcStatus_t sa_grucell_gates_fwd(const float* gi, const float* gh, const float* h_prev, float* h_out,
                                            float* stash, int B, int H, void* stream) {
    SA_CLEAR_ERR();
    if (!gi || !gh || !h_prev || !h_out || B <= 0 || H <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(grucell_gates_fwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, gi, gh,
                       h_prev, h_out, stash, B, H);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_grucell_gates_bwd(const float* dh, const float* stash, const float* h_prev, float* dgi,
                                            float* dgh, float* dh_prev, int B, int H, void* stream) {
    SA_CLEAR_ERR();
    if (!dh || !stash || !h_prev || !dgi || !dgh || !dh_prev || B <= 0 || H <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(grucell_gates_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, dh,
                       stash, h_prev, dgi, dgh, dh_prev, B, H);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

static bool att_ok(int B, int T, int H, int KS) {
    return B > 0 && B <= 65535 && T > 0 && H > 0 && KS >= 1 && KS <= 15 && (KS & 1);
}

static size_t att_layout(int B, int T, int H, int KS, size_t o[4]) {
    const size_t nchunk = (T + kAttTB - 1) / kAttTB;
    size_t p = 0;
    o[0] = p; p += sa_align_up((size_t)B * T * sizeof(float), 256);                          // scores / dpax
    o[1] = p; p += sa_align_up((size_t)B * nchunk * H * (2 + KS) * sizeof(float), 256);      // part
    o[2] = p; p += sa_align_up((size_t)B * T * KS * sizeof(float), 256);                     // q
    return p;
}

extern "C" size_t sa_attention_workspace_bytes(int B, int T, int H, int KS) {
    if (!att_ok(B, T, H, KS)) return 0;
    size_t o[4];
    return att_layout(B, T, H, KS, o);
}

static bool att_smem(const void* fn, size_t smem) {
    if (smem > 150 * 1024) return false;
    return smem <= 48 * 1024 ||
           hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
}

extern "C" ctcStatus_t sa_attention_fwd(const float* eh, const float* ox, const float* ax_prev, const float* conv_w,
                                        const float* conv_b, const float* nn_w, const float* nn_b, float scale,
                                        float* ax, float* sx, int B, int T, int H, int KS, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    if (!eh || !ox || !conv_w || !conv_b || !nn_w || !nn_b || !ax || !sx || !workspace || !att_ok(B, T, H, KS))
        return CTC_STATUS_INVALID_VALUE;
    if (workspace_bytes < sa_attention_workspace_bytes(B, T, H, KS)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    AttArgs A{eh, ox, ax_prev, conv_w, conv_b, nn_w, nn_b, scale, B, T, H, KS};
    float* score = (float*)workspace;
    const size_t smem1 = ((size_t)(kAttTB + KS - 1) + (size_t)H * KS) * sizeof(float);
    const size_t smem2 = ((size_t)T + 4 + 4 * (size_t)H) * sizeof(float);
    if (!att_smem((const void*)attention_score_kernel, smem1) || !att_smem((const void*)attention_context_kernel, smem2))
        return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(attention_score_kernel, dim3((T + kAttTB - 1) / kAttTB, B), dim3(256), smem1, stream, A, score);
    hipLaunchKernelGGL(attention_context_kernel, dim3(B), dim3(256), smem2, stream, A, score, ax, sx);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_attention_bwd(const float* eh, const float* ox, const float* ax_prev, const float* conv_w,
                                        const float* conv_b, const float* nn_w, const float* nn_b, float scale,
                                        const float* ax, const float* d_sx, const float* d_ax_next, float* d_eh,
                                        float* d_ox, float* d_ax_prev, float* g_conv_w, float* g_conv_b, float* g_nn_w,
                                        float* g_nn_b, int B, int T, int H, int KS, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    if (!eh || !ox || !conv_w || !conv_b || !nn_w || !nn_b || !ax || !d_sx || !d_eh || !d_ox || !g_conv_w ||
        !g_conv_b || !g_nn_w || !g_nn_b || !workspace || !att_ok(B, T, H, KS) || (ax_prev && !d_ax_prev))
        return CTC_STATUS_INVALID_VALUE;
    if (workspace_bytes < sa_attention_workspace_bytes(B, T, H, KS)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    size_t o[4];
    att_layout(B, T, H, KS, o);
    char* ws = (char*)workspace;
    const int nchunk = (T + kAttTB - 1) / kAttTB;
    AttArgs A{eh, ox, ax_prev, conv_w, conv_b, nn_w, nn_b, scale, B, T, H, KS};
    AttBwd G{ax, d_sx, d_ax_next, d_eh, d_ox, d_ax_prev, g_conv_w, g_conv_b, g_nn_w, g_nn_b,
             (float*)(ws + o[0]), (float*)(ws + o[1]), (float*)(ws + o[2]), nchunk};
    const size_t smem = ((size_t)(kAttTB + KS - 1) + (size_t)H * KS + 2 * kAttTB + (size_t)kAttTB * H) * sizeof(float);
    if (!att_smem((const void*)attention_bwd_main_kernel, smem)) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(attention_bwd_dax_kernel, dim3(nchunk, B), dim3(256), 0, stream, A, G);
    hipLaunchKernelGGL(attention_bwd_softmax_kernel, dim3(B), dim3(256), 0, stream, A, G);
    hipLaunchKernelGGL(attention_bwd_main_kernel, dim3(nchunk, B), dim3(256), smem, stream, A, G);
    hipLaunchKernelGGL(attention_bwd_fold_kernel, dim3(B), dim3(256), 0, stream, A, G);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_softmax_xent(const float* logits, const long long* targets, float scale, float* loss_rows,
                                       float* dlogits, long rows, int K, void* stream) {
    SA_CLEAR_ERR();
    if (!logits || !targets || !loss_rows || rows <= 0 || K <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(softmax_xent_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits,
                       targets, scale, loss_rows, dlogits, rows, K);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_argmax_rows(const float* x, long long* out, long rows, int K, void* stream) {
    SA_CLEAR_ERR();
    if (!x || !out || rows <= 0 || K <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out,
                       rows, K);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
