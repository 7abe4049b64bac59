// seq2seq.hip -- the per-token pieces of the attention decoder of /root/reference/speech/models/seq2seq.py:
//   GRUCell gate math (:20-21, :97; the two projections around it are sa_gemm_f32),
//   NNAttention (:331-360): score_t = w . relu(eh_t + ox + conv1d(ax_prev)_t) + b, optional log(T) temperature (:351-353),
//   softmax over time, context sx = sum_t ax_t eh_t -- the score network is parallel over (utterance, 16-step time
//   chunk) in both directions; only the softmax / context and the final fold run one workgroup per utterance,
//   softmax cross-entropy over the output classes (:59-63) with the gradient produced in the same pass, row argmax
//   (scheduled sampling :93 and greedy decode :157).
// All tensors fp32, contiguous, DEVICE.
#include "common.h"
#include "internal.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

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

// dh = dh + dh2 + dh3 (the two extra addends may be NULL; dh3 may alias dh_prev: it is read before the write);
// h_prev NULL = zeros (first token)
__global__ __launch_bounds__(256) void grucell_gates_bwd_kernel(const float* __restrict__ dh, const float* dh2,
                                                                const float* dh3, const float* __restrict__ stash,
                                                                const float* h_prev, float* __restrict__ dgi,
                                                                float* __restrict__ dgh, float* dh_prev, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* s = stash + (long)b * 4 * H;
    const float r = s[j], z = s[H + j], n = s[2 * H + j], ghn = s[3 * H + j];
    const float g = dh[i] + (dh2 ? dh2[i] : 0.f) + (dh3 ? dh3[i] : 0.f);
    const float hp = h_prev ? h_prev[i] : 0.f;
    const float dn = g * (1.0f - z) * (1.0f - n * n);       // through tanh
    const float dz = g * (hp - n) * z * (1.0f - z);  // through sigmoid
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
    int eh_shared = 0;     // != 0: every batch row reads the SAME (T, H) encoder states (the hypotheses of a beam search)
    // (r4) the training loop's fusions -- all null outside sa_s2s_decoder_fwd:
    // attention_score_kernel forms ox = GRUCell gates(gi, gh, h_prev) itself (every block for its utterance; the block of
    // the first time chunk also writes it to `ox_out` and the gate stash), saving the gate kernel's launch ...
    const float* gi = nullptr;      // (B, 3H)
    const float* gh = nullptr;      // (B, 3H)
    const float* h_prev = nullptr;  // (B, H)
    float* ox_out = nullptr;        // (B, H)  == ox
    float* gate_stash = nullptr;    // (B, 4H) r | z | n | gh_n
    // ... and attention_context_kernel forms the NEXT token's GRU input emb[y_next] + sx (teacher forcing), saving the index
    // and embedding launches
    const long long* y_next = nullptr;  // y + (t + 1): element b * y_stride
    long y_stride = 0;
    const float* emb = nullptr;         // (V, E), E == H
    float* ix_next = nullptr;           // (B, E)
    long long* idx_next = nullptr;      // (B)
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
    // sixteen loads in flight per thread: one load -> LDS store per trip is a serial chain of round trips (H * KS / 256 =
    // 15 of them at the shipped shapes, and it was most of the score / backward kernels' time; eight in flight (r4) still
    // made it two, on an L2 that every launch finds cold)
    const int n = A.H * A.KS, stride = blockDim.x;
    for (int base = 0; base < n; base += stride * 16) {
        float r[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = base + j * stride + (int)threadIdx.x;
            r[j] = i < n ? A.conv_w[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = base + j * stride + (int)threadIdx.x;
            if (i < n) cw[i] = r[j];
        }
    }
}

// The score network's pre-activation at (t, h) is, in this order in every kernel that forms it (forward and backward
// must agree on the ReLU mask):  v = eh[t,h] + ox[h];  c = conv_b[h];  c += conv_w[h,k] * ax_prev[t + k - pad] (k
// ascending);  v += c   -- the location term only when ax_prev is given.

// ---- forward, stage 1: scores.  grid (ceil(T / 16), B), 256 threads: four consecutive time steps per wave.
// dynamic LDS: axp[kAttTB + KS - 1] | cw[H * KS]
__global__ __launch_bounds__(256) void attention_score_kernel(AttArgs A, float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* axp = reinterpret_cast<float*>(smem_raw);
    float* cw = axp + kAttTB + A.KS - 1;
    const int b = blockIdx.y, t0 = blockIdx.x * kAttTB, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ehb = A.eh + (A.eh_shared ? 0l : (long)b * A.T * A.H);
    const float* oxb = A.ox + (long)b * A.H;
    att_stage(A, b, t0, axp, cw);
    if (A.gi) {  // fused GRUCell gates (grucell_gates_fwd_kernel's arithmetic): h of this utterance into LDS
        float* oxs = cw + A.H * A.KS;
        for (int j = threadIdx.x; j < A.H; j += blockDim.x) {
            const float* a = A.gi + (long)b * 3 * A.H;
            const float* c = A.gh + (long)b * 3 * A.H;
            const float r = sigmoidf_(a[j] + c[j]);
            const float z = sigmoidf_(a[A.H + j] + c[A.H + j]);
            const float ghn = c[2 * A.H + j];
            const float n = tanhf(a[2 * A.H + j] + r * ghn);
            const float h = (1.0f - z) * n + z * A.h_prev[(long)b * A.H + j];
            oxs[j] = h;
            if (blockIdx.x == 0) {
                A.ox_out[(long)b * A.H + j] = h;
                if (A.gate_stash) {
                    float* st = A.gate_stash + (long)b * 4 * A.H;
                    st[j] = r; st[A.H + j] = z; st[2 * A.H + j] = n; st[3 * A.H + j] = ghn;
                }
            }
        }
        oxb = oxs;
    }
    __syncthreads();
    const float nb = A.nn_b[0];
    // hidden unit outer, the wave's four time steps inner: the unit's conv taps are read from LDS once, the four eh
    // loads are independent.  Each step still sums its units in ascending order (same result as a step-outer loop).
    float vs[4] = {0.f, 0.f, 0.f, 0.f};
    // the alignment window is the same for every lane: keep its 16 + KS - 1 values in registers instead of issuing a
    // broadcast LDS read per multiply (the reads, not the arithmetic, were most of this kernel)
    // (a wave takes four consecutive steps, so its window is 4 + KS - 1 values at compile-time offsets)
    float axr[4 + 15];
#pragma unroll
    for (int i = 0; i < 4 + 15; ++i) axr[i] = i < 4 + A.KS - 1 ? axp[4 * wave + i] : 0.f;
    for (int h = lane; h < A.H; h += 64) {
        const float oxh = oxb[h], nw = A.nn_w[h];
        float cwr[16], cb = 0.f;
        if (A.ax_prev) {
            cb = A.conv_b[h];
#pragma unroll
            for (int k = 0; k < 16; ++k) cwr[k] = k < A.KS ? cw[h * A.KS + k] : 0.f;
        }
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tl = 4 * wave + q;
            e[q] = ehb[(long)min(t0 + tl, A.T - 1) * A.H + h];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = e[q] + oxh;
            if (A.ax_prev) {
                float c = cb;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < A.KS) c += cwr[k] * axr[q + k];
                v += c;
            }
            vs[q] += fmaxf(v, 0.f) * nw;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tl = 4 * wave + q;
        const float v = sa_wave_sum_dpp(vs[q]);
        if (lane == 0 && t0 + tl < A.T) score[(long)b * A.T + t0 + tl] = (v + nb) * A.scale;
    }
}

// ---- forward, stage 1 in the MFMA accumulator layout (r6; H <= 256).  Same grid.  The round-4 kernel above gives a wave four
// steps and walks the units in the lanes: 4 x 15 tap FMAs per unit and step on the VALU, the eh loads inside the unit loop
// (four dependent trips).  Here, as in attention_bwd_main2_kernel, a wave owns up to four tiles of 16 units, lane (kq, m) holds
// steps 4 kq .. 4 kq + 3 of unit 16 tile + m, the location term is four v_mfma_f32_16x16x4_f32 per tile (the same FMA chain in
// k order: the pre-activation's bits, hence the ReLU mask, are the backward kernels'), every load goes out before the gates,
// and the sum over the units is a DPP row sum + the four waves through LDS.
// dynamic LDS: axp[32] | oxs[H] | red[4][16]
__global__ __launch_bounds__(256) void attention_score2_kernel(AttArgs A, float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* axp = reinterpret_cast<float*>(smem_raw);
    float* oxs = axp + 32;
    float* red = oxs + A.H;
    const int b = blockIdx.y, t0 = blockIdx.x * kAttTB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int ntile = (A.H + 15) >> 4;
    const float* ehb = A.eh + (A.eh_shared ? 0l : (long)b * A.T * A.H);
    const bool loc = A.ax_prev != nullptr;
    float ev[4][4], cwB[4][4], wv[4], cbv[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int h = (wave + 4 * jj) * 16 + m;
        const bool hv = h < A.H;
        const int hc = hv ? h : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[jj][r] = ehb[(long)min(t0 + 4 * kq + r, A.T - 1) * A.H + hc];
        wv[jj] = hv ? A.nn_w[hc] : 0.f;
        cbv[jj] = loc ? A.conv_b[hc] : 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = 4 * s4 + kq;
            cwB[jj][s4] = (loc && hv && k < A.KS) ? A.conv_w[hc * A.KS + k] : 0.f;
        }
    }
    if (tid < 32) {
        const int t = t0 + tid - (A.KS - 1) / 2;
        axp[tid] = (loc && tid < kAttTB + A.KS - 1 && t >= 0 && t < A.T) ? A.ax_prev[(long)b * A.T + t] : 0.f;
    }
    if (A.gi) {  // fused GRUCell gates (grucell_gates_fwd_kernel's arithmetic): h of this utterance into LDS
        for (int j = tid; j < A.H; j += 256) {
            const float* a = A.gi + (long)b * 3 * A.H;
            const float* c = A.gh + (long)b * 3 * A.H;
            const float r = sigmoidf_(a[j] + c[j]);
            const float z = sigmoidf_(a[A.H + j] + c[A.H + j]);
            const float ghn = c[2 * A.H + j];
            const float n = tanhf(a[2 * A.H + j] + r * ghn);
            const float h = (1.0f - z) * n + z * A.h_prev[(long)b * A.H + j];
            oxs[j] = h;
            if (blockIdx.x == 0) {
                A.ox_out[(long)b * A.H + j] = h;
                if (A.gate_stash) {
                    float* st = A.gate_stash + (long)b * 4 * A.H;
                    st[j] = r; st[A.H + j] = z; st[2 * A.H + j] = n; st[3 * A.H + j] = ghn;
                }
            }
        }
    } else {
        for (int j = tid; j < A.H; j += 256) oxs[j] = A.ox[(long)b * A.H + j];
    }
    __syncthreads();
    float aA[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) aA[s4] = axp[m + 4 * s4 + kq];
    float vs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int tile = wave + 4 * jj;
        if (tile < ntile) {
            const int h = tile * 16 + m;
            const float oxh = oxs[h < A.H ? h : 0];
            f32x4_t c1 = {cbv[jj], cbv[jj], cbv[jj], cbv[jj]};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aA[s4], cwB[jj][s4], c1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) vs[r] += fmaxf((ev[jj][r] + oxh) + c1[r], 0.f) * wv[jj];   // wv = 0 beyond H
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = vs[r];
        v += SA_DPP_F(0.f, v, 0x111, 0xf);
        v += SA_DPP_F(0.f, v, 0x112, 0xf);
        v += SA_DPP_F(0.f, v, 0x114, 0xf);
        v += SA_DPP_F(0.f, v, 0x118, 0xf);
        if (m == 15) red[wave * 16 + 4 * kq + r] = v;
    }
    __syncthreads();
    if (tid < kAttTB && t0 + tid < A.T) {
        const float v = (red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]);
        score[(long)b * A.T + t0 + tid] = (v + A.nn_b[0]) * A.scale;
    }
}

// the score launch: the MFMA-layout kernel for H <= 256 (option s2s.kernels bit 1)
static void score_launch(const AttArgs& A, float* score, int grid_y, size_t smem_round4, hipStream_t stream) {
    const dim3 grid((A.T + kAttTB - 1) / kAttTB, grid_y);
    if (A.H <= 256 && (sa_opt(SA_OPT_S2S_KERNELS) & 2) != 0)
        hipLaunchKernelGGL(attention_score2_kernel, grid, dim3(256), (size_t)(32 + A.H + 64) * sizeof(float), stream, A, score);
    else
        hipLaunchKernelGGL(attention_score_kernel, grid, dim3(256), smem_round4, stream, A, score);
}

// ---- forward, stage 2: softmax over time and the context.  grid B, 256 threads.  dynamic LDS: a[T] | red[4] | pad | part[4][H]
__global__ __launch_bounds__(256) void attention_context_kernel(AttArgs A, const float* __restrict__ score,
                                                                float* __restrict__ ax, float* __restrict__ sx,
                                                                float* __restrict__ oin /* ox + sx, or NULL */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* a = reinterpret_cast<float*>(smem_raw);
    float* red = a + A.T;
    float* part = a + ((A.T + 4 + 3) & ~3);  // 16-byte aligned: the context partials are written as float4
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ehb = A.eh + (A.eh_shared ? 0l : (long)b * A.T * A.H);
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
    if ((A.H & 3) == 0) {
        // a lane owns 4 consecutive hidden units (one 16-byte load per row), 4 rows in flight; the sum over t keeps its
        // ascending order, so the result is the same as the scalar loop's
        const int H4 = A.H >> 2;
        const float4* e4 = reinterpret_cast<const float4*>(ehb);
        for (int c = lane; c < H4; c += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int t = tb;
            for (; t + 4 <= te; t += 4) {
                const float4 v0 = e4[(long)t * H4 + c], v1 = e4[(long)(t + 1) * H4 + c];
                const float4 v2 = e4[(long)(t + 2) * H4 + c], v3 = e4[(long)(t + 3) * H4 + c];
                const float a0 = a[t], a1 = a[t + 1], a2 = a[t + 2], a3 = a[t + 3];
                acc.x += a0 * v0.x; acc.y += a0 * v0.y; acc.z += a0 * v0.z; acc.w += a0 * v0.w;
                acc.x += a1 * v1.x; acc.y += a1 * v1.y; acc.z += a1 * v1.z; acc.w += a1 * v1.w;
                acc.x += a2 * v2.x; acc.y += a2 * v2.y; acc.z += a2 * v2.z; acc.w += a2 * v2.w;
                acc.x += a3 * v3.x; acc.y += a3 * v3.y; acc.z += a3 * v3.z; acc.w += a3 * v3.w;
            }
            for (; t < te; ++t) {
                const float4 v = e4[(long)t * H4 + c];
                const float at = a[t];
                acc.x += at * v.x; acc.y += at * v.y; acc.z += at * v.z; acc.w += at * v.w;
            }
            *reinterpret_cast<float4*>(part + wave * A.H + 4 * c) = acc;
        }
    } else {
        for (int h = lane; h < A.H; h += 64) {
            float acc = 0.f;
            for (int t = tb; t < te; ++t) acc += a[t] * ehb[(long)t * A.H + h];
            part[wave * A.H + h] = acc;
        }
    }
    __syncthreads();
    const long long yn = A.ix_next ? A.y_next[(long)b * A.y_stride] : 0;
    for (int h = threadIdx.x; h < A.H; h += 256) {
        const float v = (part[h] + part[A.H + h]) + (part[2 * A.H + h] + part[3 * A.H + h]);
        sx[(long)b * A.H + h] = v;
        if (oin) oin[(long)b * A.H + h] = A.ox[(long)b * A.H + h] + v;
        if (A.ix_next) A.ix_next[(long)b * A.H + h] = A.emb[yn * A.H + h] + v;   // s2s_embed_add_kernel's sum
    }
    if (A.ix_next && threadIdx.x == 0) A.idx_next[b] = yn;
}

// ---- forward, stage 2 on four workgroups per utterance (r6; H = 16, 32, 64, 128 or 256).  One workgroup walking its
// utterance's T x H encoder states alone (above) is as slow as one CU's address path: 200 KB in 6 us at the shipped shapes.
// Here workgroup (b, q) redoes the softmax (T values) and forms the context of a QUARTER of the hidden units: its 256 threads
// are H / 16 lanes of four units x 4096 / H time groups, a thread adds rows g, g + G, ... (eight 16-byte loads in flight),
// the groups are added in group order through LDS.  Quarter 0 writes the alignment.
// dynamic LDS: a[T] | red[4] | pad | part[G][H / 4]
__global__ __launch_bounds__(256) void attention_context2_kernel(AttArgs A, const float* __restrict__ score,
                                                                 float* __restrict__ ax, float* __restrict__ sx,
                                                                 float* __restrict__ oin /* ox + sx, or NULL */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* a = reinterpret_cast<float*>(smem_raw);
    float* red = a + A.T;
    float* part = a + ((A.T + 4 + 3) & ~3);
    const int b = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
    const float* ehb = A.eh + (A.eh_shared ? 0l : (long)b * A.T * A.H);
    const int HB = A.H >> 2, L4 = HB >> 2, G = 256 / L4;   // units per workgroup, 16-byte lanes per row, time groups
    const int g = tid / L4, c = tid - g * L4;
    const float4* e4 = reinterpret_cast<const float4*>(ehb + q * HB) + c;   // row t: e4[t * (H / 4)]
    const int H4 = A.H >> 2;
    // the first eight rows of the thread go out before the softmax (they do not depend on it)
    float4 v0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v0[u] = e4[(long)min(g + u * G, A.T - 1) * H4];
    float mx = -3.0e38f;
    for (int t = tid; t < A.T; t += 256) {
        a[t] = score[(long)b * A.T + t];
        mx = fmaxf(mx, a[t]);
    }
    mx = block_reduce(mx, red, true);
    float s = 0.f;
    for (int t = tid; t < A.T; t += 256) {
        const float e = __expf(a[t] - mx);
        a[t] = e;
        s += e;
    }
    s = block_reduce(s, red, false);
    const float inv = 1.0f / s;
    for (int t = tid; t < A.T; t += 256) {
        const float v = a[t] * inv;
        a[t] = v;
        if (q == 0) ax[(long)b * A.T + t] = v;
    }
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int t = g + u * G;
        if (t < A.T) {
            const float at = a[t];
            acc.x += at * v0[u].x; acc.y += at * v0[u].y; acc.z += at * v0[u].z; acc.w += at * v0[u].w;
        }
    }
    for (int t0 = g + 8 * G; t0 < A.T; t0 += 8 * G) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = e4[(long)min(t0 + u * G, A.T - 1) * H4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + u * G;
            if (t < A.T) {
                const float at = a[t];
                acc.x += at * v[u].x; acc.y += at * v[u].y; acc.z += at * v[u].z; acc.w += at * v[u].w;
            }
        }
    }
    *reinterpret_cast<float4*>(part + g * HB + 4 * c) = acc;
    __syncthreads();
    if (tid < HB) {
        float v = 0.f;
        for (int gg = 0; gg < G; ++gg) v += part[gg * HB + tid];
        const int h = q * HB + tid;
        sx[(long)b * A.H + h] = v;
        if (oin) oin[(long)b * A.H + h] = A.ox[(long)b * A.H + h] + v;
        if (A.ix_next) {
            const long long yn = A.y_next[(long)b * A.y_stride];
            A.ix_next[(long)b * A.H + h] = A.emb[yn * A.H + h] + v;   // s2s_embed_add_kernel's sum
            if (q == 0 && tid == 0) A.idx_next[b] = yn;
        }
    }
}

static bool context2_ok(int H) { return H == 16 || H == 32 || H == 64 || H == 128 || H == 256; }
static size_t context2_smem(int T, int H) { return ((size_t)((T + 4 + 3) & ~3) + (size_t)(4096 / H) * (H / 4)) * sizeof(float); }

// the context launch: four workgroups per utterance where the width allows (option s2s.kernels bit 1)
static void context_launch(const AttArgs& A, const float* score, float* ax, float* sx, float* oin, int rows,
                           size_t smem_round4, hipStream_t stream) {
    if (context2_ok(A.H) && (sa_opt(SA_OPT_S2S_KERNELS) & 2) != 0 && context2_smem(A.T, A.H) <= 48 * 1024)
        hipLaunchKernelGGL(attention_context2_kernel, dim3(rows, 4), dim3(256), context2_smem(A.T, A.H), stream, A, score, ax, sx,
                           oin);
    else
        hipLaunchKernelGGL(attention_context_kernel, dim3(rows), dim3(256), smem_round4, stream, A, score, ax, sx, oin);
}

struct AttBwd {
    const float* ax;         // (B, T) this step's alignment
    const float* d_sx;       // (B, H)
    const float* d_sx2;      // (B, H) or NULL: added to d_sx (the next token's input gradient)
    const float* d_ax_next;  // (B, T) or NULL: gradient arriving through the next token's location term
    float* d_eh;             // (B, T, H)  +=
    float* d_ox;             // (B, H)     =
    float* d_ax_prev;        // (B, T)     =   (unused when ax_prev is NULL)
    float* g_conv_w;         // (B, H, KS) +=  per-utterance partials, reduced over B by the caller
    float* g_conv_b;         // (B, H)     +=
    float* g_nn_w;           // (B, H)     +=
    float* g_nn_b;           // (B)        +=
    float* dpax;             // workspace (B, T): d ax, then d score
    float* part;             // workspace (B, nchunk, 2 + KS, H): per-chunk sums of d_pre, d_pre-weighted terms
    float* q;                // workspace (B, T, KS): sum_h d_pre[t,h] conv_w[h,k]
    int nchunk;
    // (r4) fusions of the training loop (zero / null outside sa_s2s_decoder_bwd):
    int fused_softmax = 0;   // attention_bwd_main_kernel goes through the softmax itself: dpax holds d ax (stage 1's output) and
                             // every block recomputes the utterance's sum_t ax d ax (T values); no attention_bwd_softmax launch
    // attention_bwd_fold_kernel's d_ox block continues with the GRUCell gate gradients of its utterance (grucell_gates_bwd_kernel's
    // arithmetic): dh = dh_a + d_ox + dh_c
    const float* gb_dh_a = nullptr;     // (B, H) gradient of the state from the fc (dOIN[t])
    const float* gb_stash = nullptr;    // (B, 4H)
    const float* gb_h_prev = nullptr;   // (B, H) or NULL (first token)
    float* gb_dgi = nullptr;            // (B, 3H)
    float* gb_dgh = nullptr;            // (B, 3H)
    float* gb_dh_prev = nullptr;        // (B, H): read (dh_c, the gradient arriving from the next token's GRU) then written
    // (r6) attention_bwd_main2_kernel: stages 1 - 3 in ONE launch.  The softmax backward needs s = sum_t ax[t] d ax[t] over the
    // whole utterance, and d ax[t] = d_sx . eh[t] + d_ax_next[t], so s = d_sx . (sum_t ax[t] eh[t]) + sum_t ax[t] d_ax_next[t]
    // = d_sx . sx + ...: the forward's context vector (sx = oin - ox, both stashed) replaces the pass over the utterance that
    // made stage 1 a launch of its own; a block forms d ax for its own 16 steps from the eh rows it loads anyway.
    const float* oin = nullptr;         // (B, H) ox + sx of this token (the fc's input)
    int sb_chunks = 0;                  // != 0: dpax holds (B, nchunk) per-chunk sums of d score (main2), folded into g_nn_b
                                        // by attention_bwd_fold_kernel's last block row
};

// ---- backward, stage 1: d ax[t] = (from the next token) + d_sx . eh[t].  grid (ceil(T / 16), B), a wave per time step
__global__ __launch_bounds__(256) void attention_bwd_dax_kernel(AttArgs A, AttBwd G) {
    const int b = blockIdx.y, t0 = blockIdx.x * kAttTB, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ehb = A.eh + (long)b * A.T * A.H;
    const float* dsx = G.d_sx + (long)b * A.H;
    const float* dsx2 = G.d_sx2 ? G.d_sx2 + (long)b * A.H : nullptr;
    for (int tl = wave; tl < kAttTB && t0 + tl < A.T; tl += 4) {
        const int t = t0 + tl;
        float v = 0.f;
        for (int h = lane; h < A.H; h += 64) v += (dsx[h] + (dsx2 ? dsx2[h] : 0.f)) * ehb[(long)t * A.H + h];
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
// dynamic LDS: axp | cw | dps[kAttTB] | axs[kAttTB] | dp_tile[kAttTB][H + 1]  (odd pitch: phase 2's lanes read different
// frames of one hidden unit -- a pitch of H put all of them on one bank)
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
    float ssum = 0.f;
    if (G.fused_softmax) {   // s = sum_t ax[t] d ax[t] over the WHOLE utterance (attention_bwd_softmax_kernel's first pass)
        __shared__ float red[4];
        const float* axb = G.ax + (long)b * A.T;
        const float* dab = G.dpax + (long)b * A.T;
        float sp = 0.f;
        for (int t = threadIdx.x; t < A.T; t += 256) sp += axb[t] * dab[t];
        ssum = block_reduce(sp, red, false);
        if (chunk == 0) {   // the score bias' gradient: sum_t d score[t] (zero up to rounding: the softmax is shift invariant)
            float sb = 0.f;
            for (int t = threadIdx.x; t < A.T; t += 256) sb += axb[t] * (dab[t] - ssum) * A.scale;
            sb = block_reduce(sb, red, false);
            if (threadIdx.x == 0) G.g_nn_b[b] += sb;
        }
    }
    if ((int)threadIdx.x < kAttTB) {
        const bool ok = (int)threadIdx.x < nt;
        const float axv = ok ? G.ax[(long)b * A.T + t0 + threadIdx.x] : 0.f;
        const float dv = ok ? G.dpax[(long)b * A.T + t0 + threadIdx.x] : 0.f;
        dps[threadIdx.x] = G.fused_softmax ? axv * (dv - ssum) * A.scale : dv;
        axs[threadIdx.x] = axv;
    }
    __syncthreads();
    const int W = 2 + A.KS;
    for (int h = threadIdx.x; h < A.H; h += 256) {
        const float w = A.nn_w[h], dsxh = dsx[h] + (G.d_sx2 ? G.d_sx2[(long)b * A.H + h] : 0.f);
        float a_ox = 0.f, a_nw = 0.f;
        float a_cw[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a_cw[k] = 0.f;
        // the unit's conv taps from LDS once; the chunk's 16 eh / d_eh values are fetched up front (16 serial
        // load -> compute -> store round trips were the whole cost of this kernel)
        const float oxh = oxb[h];
        float cwr[16], cb = 0.f;
        if (A.ax_prev) {
            cb = A.conv_b[h];
#pragma unroll
            for (int k = 0; k < 16; ++k) cwr[k] = k < A.KS ? cw[h * A.KS + k] : 0.f;
        }
        // the alignment window and the per-step scalars are the same for every lane: registers, not an LDS read per use
        float axr[kAttTB + 15], dpr[kAttTB], axsr[kAttTB];
#pragma unroll
        for (int i = 0; i < kAttTB + 15; ++i) axr[i] = i < kAttTB + A.KS - 1 ? axp[i] : 0.f;
#pragma unroll
        for (int i = 0; i < kAttTB; ++i) { dpr[i] = dps[i]; axsr[i] = axs[i]; }
        float ev[kAttTB], dv[kAttTB];
#pragma unroll
        for (int tl = 0; tl < kAttTB; ++tl) {
            const long o = (long)(t0 + (tl < nt ? tl : 0)) * A.H + h;
            ev[tl] = ehb[o];
            dv[tl] = dehb[o];
        }
#pragma unroll
        for (int tl = 0; tl < kAttTB; ++tl) {
            if (tl < nt) {
                float pre = ev[tl] + oxh;
                if (A.ax_prev) {
                    float c = cb;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < A.KS) c += cwr[k] * axr[tl + k];
                    pre += c;
                }
                const float dp = pre > 0.f ? dpr[tl] * w : 0.f;
                a_ox += dp;
                a_nw += dpr[tl] * fmaxf(pre, 0.f);
                if (A.ax_prev) {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < A.KS) a_cw[k] += dp * axr[tl + k];
                }
                dehb[(long)(t0 + tl) * A.H + h] = dv[tl] + (axsr[tl] * dsxh + dp);
                dp_tile[tl * (A.H + 1) + h] = dp;
            }
        }
        float* p = G.part + ((long)b * G.nchunk + chunk) * W * A.H + h;   // (B, nchunk, 2 + KS, H): quantity-major
        p[0] = a_ox;
        p[A.H] = a_nw;
        for (int k = 0; k < A.KS; ++k) p[(long)(2 + k) * A.H] = a_cw[k];
    }
    if (!A.ax_prev) return;
    __syncthreads();
    for (int i = threadIdx.x; i < nt * A.KS; i += 256) {
        const int tl = i / A.KS, k = i - tl * A.KS;
        float acc = 0.f;
        for (int h = 0; h < A.H; ++h) acc += dp_tile[tl * (A.H + 1) + h] * cw[h * A.KS + k];
        G.q[((long)b * A.T + t0 + tl) * A.KS + k] = acc;
    }
}

// ---- backward, stages 1 - 3 in one launch (H <= 256, H % 4 == 0).  grid (nchunk, B), 256 threads.
// One wave per SIMD runs this kernel's instruction stream alone, so the stream's LENGTH is the kernel's time: a thread per
// hidden unit walking its 16 steps (the round-4 layout) was ~2900 instructions, 480 of them the two 16 x 15 tap sums.  Here
// a wave owns up to four tiles of 16 hidden units in the MFMA accumulator layout -- lane (kq = lane / 16, m = lane % 16)
// holds steps 4 kq .. 4 kq + 3 of unit 16 tile + m -- and the tap sums are f32 MFMAs (v_mfma_f32_16x16x4_f32, exact fp32 FMA
// chains in k order):
//   location term  c[t][h] = conv_b[h] + sum_k ax_prev[t + k - pad] cw[h][k]   A = the alignment window (Toeplitz), B = cw^T
//   tap gradients  a_cw[h][k] = sum_t d_pre[t][h] ax_prev[t + k - pad]         A = d_pre as it sits in the accumulator
//                  (column 15 of B is ones: the same product delivers a_ox[h] = sum_t d_pre[t][h])
//   a_nw[h] = sum_t d score[t] relu(pre[t][h])                                 A = the products, B = ones
//   q[t][k] = sum_h d_pre[t][h] cw[h][k]   (phase 2, out of an LDS tile of d_pre; was a 256-iteration LDS loop per (t, k))
// Every global load of the block is issued up front; ONE barrier later every lane holds s and its four d score values (the
// 16 sums d_sx . eh[t] by DPP row sums, combined across the waves from LDS in wave order).
// dynamic LDS: axp[32] | cw[H * KS] | red[4][17] axs[16] daxn[16] (100) | dp_tile[kAttTB][H + 1] | qp[4][256]
constexpr int kRed2 = 100;
constexpr int kAxp2 = 32;
__global__ __launch_bounds__(256) void attention_bwd_main2_kernel(AttArgs A, AttBwd G) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* axp = reinterpret_cast<float*>(smem_raw);
    float* cw = axp + kAxp2;
    float* red = cw + A.H * A.KS;
    float* axs = red + 68;
    float* daxn = axs + kAttTB;
    float* dp_tile = red + kRed2;
    float* qp = dp_tile + kAttTB * (A.H + 1);
    const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * kAttTB;
    const int nt = min(kAttTB, A.T - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int ntile = (A.H + 15) >> 4;
    const float* ehb = A.eh + (long)b * A.T * A.H;
    float* dehb = G.d_eh + (long)b * A.T * A.H;
    const float* axb = G.ax + (long)b * A.T;
    const float* dnb = G.d_ax_next ? G.d_ax_next + (long)b * A.T : nullptr;
    const bool loc = A.ax_prev != nullptr;
    // ---- every global load of the block, before anything waits
    float ev[4][4], dv[4][4], cwB[4][4], oxh[4], dsxh[4], wv[4], cbv[4];
    float sp = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int h = (wave + 4 * jj) * 16 + m;
        const bool hv = h < A.H;   // false for the whole wave beyond the last tile, for some lanes inside a ragged last tile
        const int hc = hv ? h : 0;
        const long bh = (long)b * A.H + hc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tl = 4 * kq + r;
            const long o = (long)(t0 + (tl < nt ? tl : 0)) * A.H + hc;
            ev[jj][r] = ehb[o];
            dv[jj][r] = dehb[o];
        }
        oxh[jj] = A.ox[bh];
        dsxh[jj] = hv ? G.d_sx[bh] + (G.d_sx2 ? G.d_sx2[bh] : 0.f) : 0.f;
        wv[jj] = hv ? A.nn_w[hc] : 0.f;
        cbv[jj] = loc ? A.conv_b[hc] : 0.f;
        if (kq == 0) sp += dsxh[jj] * (G.oin[bh] - oxh[jj]);   // a unit sits in four lanes (one per step group): counted once
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = 4 * s4 + kq;
            cwB[jj][s4] = (loc && hv && k < A.KS) ? A.conv_w[hc * A.KS + k] : 0.f;
        }
    }
    float axw = 0.f;   // the alignment window of the location term: ax_prev[t0 - pad + tid], zero beyond the ends
    if (tid < kAxp2) {
        const int t = t0 + tid - (A.KS - 1) / 2;
        axw = (loc && tid < kAttTB + A.KS - 1 && t >= 0 && t < A.T) ? A.ax_prev[(long)b * A.T + t] : 0.f;
    }
    if (dnb)
        for (int t = tid; t < A.T; t += 256) sp += axb[t] * dnb[t];
    if (tid < kAttTB) {
        const bool ok = tid < nt;
        axs[tid] = ok ? axb[t0 + tid] : 0.f;
        daxn[tid] = (ok && dnb) ? dnb[t0 + tid] : 0.f;
    }
    if (tid < kAxp2) axp[tid] = axw;
    if (loc) {   // the taps into LDS for phase 2's B operand (every element of cw is held by exactly one lane)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int h = (wave + 4 * jj) * 16 + m;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int k = 4 * s4 + kq;
                if (h < A.H && k < A.KS) cw[h * A.KS + k] = cwB[jj][s4];
            }
        }
    }
    // ---- the block's 17 sums: s, and d_sx . eh[t] for the chunk's steps (a lane: its four steps over its four units, then the
    // row of 16 lanes by DPP, then the four waves through LDS)
    {
        const float v = sa_wave_sum_dpp(sp);
        if (lane == 0) red[wave * 17 + 16] = v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = (dsxh[0] * ev[0][r] + dsxh[1] * ev[1][r]) + (dsxh[2] * ev[2][r] + dsxh[3] * ev[3][r]);
        v += SA_DPP_F(0.f, v, 0x111, 0xf);
        v += SA_DPP_F(0.f, v, 0x112, 0xf);
        v += SA_DPP_F(0.f, v, 0x114, 0xf);
        v += SA_DPP_F(0.f, v, 0x118, 0xf);
        if (m == 15) red[wave * 17 + 4 * kq + r] = v;
    }
    __syncthreads();
    const float ssum = (red[16] + red[17 + 16]) + (red[34 + 16] + red[51 + 16]);
    float dpr[4], axsr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tl = 4 * kq + r;
        const float dax = ((red[tl] + red[17 + tl]) + (red[34 + tl] + red[51 + tl])) + daxn[tl];
        axsr[r] = axs[tl];
        dpr[r] = axsr[r] * (dax - ssum) * A.scale;   // 0 beyond the utterance's end (axs = 0)
    }
    {   // the score bias' gradient, per chunk (folded by stage 4)
        const int tl = lane & 15;
        const float dax = ((red[tl] + red[17 + tl]) + (red[34 + tl] + red[51 + tl])) + daxn[tl];
        const float v = sa_wave_sum_dpp(lane < 16 ? axs[tl] * (dax - ssum) * A.scale : 0.f);
        if (tid == 0) G.dpax[(long)b * G.nchunk + chunk] = v;
    }
    float aA[4], bX[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        aA[s4] = axp[m + 4 * s4 + kq];                         // A[step m][tap 4 s4 + kq] of the location term
        bX[s4] = m < 15 ? axp[4 * kq + s4 + m] : 1.0f;         // [tap m][step 4 kq + s4] of the window; tap 15: ones
    }
    const int W = 2 + A.KS;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int tile = wave + 4 * jj;
        if (tile < ntile) {
            const int h = tile * 16 + m;
            const bool hv = h < A.H;
            f32x4_t c1 = {cbv[jj], cbv[jj], cbv[jj], cbv[jj]};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aA[s4], cwB[jj][s4], c1, 0, 0, 0);
            float dp[4], nw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tl = 4 * kq + r;
                const float pre = (ev[jj][r] + oxh[jj]) + c1[r];
                dp[r] = pre > 0.f ? dpr[r] * wv[jj] : 0.f;                  // wv = 0 beyond H, dpr = 0 beyond nt
                nw[r] = hv ? dpr[r] * fmaxf(pre, 0.f) : 0.f;
                if (hv && tl < nt) {
                    dehb[(long)(t0 + tl) * A.H + h] = dv[jj][r] + (axsr[r] * dsxh[jj] + dp[r]);
                    dp_tile[tl * (A.H + 1) + h] = dp[r];
                }
            }
            // the products transposed (the window as A, d_pre as B): accumulator row 4 kq + r = tap (15: a_ox), column m = unit,
            // so a quantity's 16 units of the tile are one 64-byte run of the quantity-major partials (B, nchunk, 2 + KS, H)
            f32x4_t c2 = {0.f, 0.f, 0.f, 0.f}, c3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(bX[r], dp[r], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, nw[r], c3, 0, 0, 0);
            }
            if (hv) {
                float* p = G.part + ((long)b * G.nchunk + chunk) * W * A.H + h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 4 * kq + r;
                    if (n == 15) p[0] = c2[r];
                    else if (loc && n < A.KS) p[(long)(2 + n) * A.H] = c2[r];
                }
                if (kq == 0) p[A.H] = c3[0];
            }
        }
    }
    if (!loc) return;
    __syncthreads();
    {   // A[m = step][k = unit] from dp_tile, B[k = unit][n = tap] from cw (column 15 and the rows beyond nt are never stored)
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const int nk = A.H >> 2;
        int ks = wave;
        for (; ks + 4 < nk; ks += 8) {
            const int h0 = ks * 4 + kq, h1 = h0 + 16;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dp_tile[m * (A.H + 1) + h0], cw[h0 * A.KS + m], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dp_tile[m * (A.H + 1) + h1], cw[h1 * A.KS + m], acc1, 0, 0, 0);
        }
        if (ks < nk) {
            const int h0 = ks * 4 + kq;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dp_tile[m * (A.H + 1) + h0], cw[h0 * A.KS + m], acc0, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) qp[wave * 256 + (4 * kq + r) * 16 + m] = acc0[r] + acc1[r];
    }
    __syncthreads();
    {
        const int tl = tid >> 4, k = tid & 15;
        const float v = (qp[tid] + qp[256 + tid]) + (qp[512 + tid] + qp[768 + tid]);
        if (tl < nt && k < A.KS) G.q[((long)b * A.T + t0 + tl) * A.KS + k] = v;
    }
}

// ---- backward, stage 4: fold the chunks.  grid (2 + KS + 1, B), 256 threads: block (j, b) folds quantity j of the
// per-chunk partials for every hidden unit (j = 0: d_pre sums -> d_ox and conv bias, 1: nn weight, 2 + k: conv tap k);
// the last block row gathers d ax_prev from q.  (One block per utterance doing all 17 quantities serially was 25 us of
// dependent L2 round trips on 16 CUs.)  Sums stay in chunk order.
__global__ __launch_bounds__(256) void attention_bwd_fold_kernel(AttArgs A, AttBwd G) {
    const int b = blockIdx.y, j = blockIdx.x;
    const int W = 2 + A.KS;
    if (j < W) {
        if (j >= 2 && !A.ax_prev) return;
        for (int h = threadIdx.x; h < A.H; h += 256) {
            const float* p = G.part + ((long)b * G.nchunk * W + j) * A.H + h;
            const long cs = (long)A.H * W;
            // what the sum is added to / continued with is fetched with the partials: one round trip, not two or three
            const long i = (long)b * A.H + h;
            float r = 0.f, z = 0.f, n = 0.f, ghn = 0.f, dha = 0.f, dhc = 0.f, hp = 0.f, old = 0.f;
            if (j == 0) {
                if (A.ax_prev) old = G.g_conv_b[i];
                if (G.gb_dgi) {
                    const float* sg = G.gb_stash + (long)b * 4 * A.H;
                    r = sg[h]; z = sg[A.H + h]; n = sg[2 * A.H + h]; ghn = sg[3 * A.H + h];
                    dha = G.gb_dh_a[i]; dhc = G.gb_dh_prev[i];
                    hp = G.gb_h_prev ? G.gb_h_prev[i] : 0.f;
                }
            } else if (j == 1) {
                old = G.g_nn_w[i];
            } else {
                old = G.g_conv_w[i * A.KS + j - 2];
            }
            // sixteen chunks' partials in flight (four (r4) made a T' = 197 utterance's 13 chunks four dependent trips)
            float acc = 0.f;
            for (int c0 = 0; c0 < G.nchunk; c0 += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = c0 + u < G.nchunk ? p[(c0 + u) * cs] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (c0 + u < G.nchunk) acc += v[u];
            }
            if (j == 0) {
                G.d_ox[i] = acc;
                if (A.ax_prev) G.g_conv_b[i] = old + acc;
                if (G.gb_dgi) {   // grucell_gates_bwd_kernel for (b, h): dh = dOIN + d_ox + (from the next token's GRU)
                    const float g = dha + acc + dhc;
                    const float dn = g * (1.0f - z) * (1.0f - n * n);
                    const float dz = g * (hp - n) * z * (1.0f - z);
                    const float dr = dn * ghn * r * (1.0f - r);
                    float* a = G.gb_dgi + (long)b * 3 * A.H;
                    float* c = G.gb_dgh + (long)b * 3 * A.H;
                    a[h] = dr; a[A.H + h] = dz; a[2 * A.H + h] = dn;
                    c[h] = dr; c[A.H + h] = dz; c[2 * A.H + h] = dn * r;
                    G.gb_dh_prev[i] = g * z;
                }
            } else if (j == 1) {
                G.g_nn_w[i] = old + acc;
            } else {
                G.g_conv_w[i * A.KS + j - 2] = old + acc;
            }
        }
        return;
    }
    if (G.sb_chunks && threadIdx.x < 64) {   // main2's per-chunk sums of d score (a lane per chunk: one round trip)
        float acc = 0.f;
        for (int c = threadIdx.x; c < G.nchunk; c += 64) acc += G.dpax[(long)b * G.nchunk + c];
        acc = sa_wave_sum_dpp(acc);
        if (threadIdx.x == 0) G.g_nn_b[b] += acc;
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
                       nullptr, nullptr, stash, h_prev, dgi, dgh, dh_prev, B, H);
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
    const size_t smem2 = ((size_t)T + 8 + 4 * (size_t)H) * sizeof(float);
    if (!att_smem((const void*)attention_score_kernel, smem1) || !att_smem((const void*)attention_context_kernel, smem2))
        return CTC_STATUS_INVALID_VALUE;
    score_launch(A, score, B, smem1, stream);
    context_launch(A, score, ax, sx, (float*)nullptr, B, smem2, stream);
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
    AttBwd G{ax, d_sx, nullptr, d_ax_next, d_eh, d_ox, d_ax_prev, g_conv_w, g_conv_b, g_nn_w, g_nn_b,
             (float*)(ws + o[0]), (float*)(ws + o[1]), (float*)(ws + o[2]), nchunk};
    const size_t smem = ((size_t)(kAttTB + KS - 1) + (size_t)H * KS + 2 * kAttTB + (size_t)kAttTB * (H + 1)) * sizeof(float);
    if (!att_smem((const void*)attention_bwd_main_kernel, smem)) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(attention_bwd_dax_kernel, dim3(nchunk, B), dim3(256), 0, stream, A, G);
    hipLaunchKernelGGL(attention_bwd_softmax_kernel, dim3(B), dim3(256), 0, stream, A, G);
    hipLaunchKernelGGL(attention_bwd_main_kernel, dim3(nchunk, B), dim3(256), smem, stream, A, G);
    hipLaunchKernelGGL(attention_bwd_fold_kernel, dim3(2 + KS + 1, B), dim3(256), 0, stream, A, G);
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

// =====================================================================================================================
// The decoder loop itself (seq2seq.py:77-112 and its BPTT), host side in C++: one library call enqueues every token's
// kernels back to back (13 launches per token forward + backward), so the Python interpreter is off the launch path
// -- the same move as sa_gru_stack_* for the encoder.  Tokens are strictly sequential (token t's input contains token
// t-1's context), batch rows are few (B = 16 in the shipped config), so the projections are "skinny" products.
// =====================================================================================================================
namespace {


// C[m, n] = sum_k A[m, k] B[n, k] (+ bias[n]) (+ C[m, n])  for few rows m: a workgroup owns a 16 x 16 output tile,
// its four waves split K (v_mfma_f32_16x16x4_f32, 16-byte k-contiguous operand loads on both sides, LDS reduce).
// Up to two problems per launch (blockIdx.z).  K % 4 == 0, lda / ldb % 4 == 0.
struct SkinnyProb {
    const float* A;
    const float* B;
    const float* bias;   // or NULL
    float* C;
    int N, K, accumulate;
    long lda, ldb, ldc;
};
struct SkinnyArgs {
    SkinnyProb p[2];
    int M;
};

__global__ __launch_bounds__(256) void skinny_gemm_nt_kernel(SkinnyArgs S) {
    __shared__ float red[4][256];
    const SkinnyProb& P = S.p[blockIdx.z];
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    if (n0 >= P.N) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int kq = ((P.K / 4 + 3) / 4) * 4;  // per-wave K range, a multiple of 4
    const int kbeg = wave * kq, kend = min(P.K, kbeg + kq);
    const int row = min(m0 + i, S.M - 1), col = min(n0 + i, P.N - 1);
    const float* a_row = P.A + (long)row * P.lda;
    const float* b_row = P.B + (long)col * P.ldb;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 64) {  // 4 x 16 k per trip: four independent 16-byte loads per side in flight
        float4 a[4], b[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int k = k0 + 16 * it + 4 * g;
            const bool ok = k < kend;
            a[it] = ok ? *reinterpret_cast<const float4*>(a_row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[it] = ok ? *reinterpret_cast<const float4*>(b_row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, b[it].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, b[it].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, b[it].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, b[it].w, acc, 0, 0, 0);
        }
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(g * 4 + r) * 16 + i] = acc[r];
    __syncthreads();
    const int rr = threadIdx.x >> 4, cc = threadIdx.x & 15;
    const int m = m0 + rr, n = n0 + cc;
    if (m < S.M && n < P.N) {
        float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (P.bias) v += P.bias[n];
        float* c = P.C + (long)m * P.ldc + n;
        if (P.accumulate) v += *c;
        *c = v;
    }
}

void skinny_launch(const SkinnyProb* probs, int nprob, int M, hipStream_t stream) {
    SkinnyArgs S;
    int maxN = 0;
    for (int i = 0; i < nprob; ++i) { S.p[i] = probs[i]; maxN = probs[i].N > maxN ? probs[i].N : maxN; }
    if (nprob == 1) S.p[1] = S.p[0];
    S.M = M;
    hipLaunchKernelGGL(skinny_gemm_nt_kernel, dim3((maxN + 15) / 16, (M + 15) / 16, nprob), dim3(256), 0, stream, S);
}

// IDX[b] = y[b * U + t]
__global__ void s2s_idx_kernel(const long long* __restrict__ y, long long* __restrict__ idx, int B, int U, int t) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) idx[b] = y[(long)b * U + t];
}

// ix[b, :] = emb[idx[b], :] (+ sx[b, :])
__global__ __launch_bounds__(256) void s2s_embed_add_kernel(const float* __restrict__ emb, const long long* __restrict__ idx,
                                                            const float* sx, float* __restrict__ ix, int E) {
    const int b = blockIdx.x;
    const float* src = emb + (long)idx[b] * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x)
        ix[(long)b * E + e] = src[e] + (sx ? sx[(long)b * E + e] : 0.f);
}

// out (C, R) = in (R, C)^T
__global__ void s2s_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < C) out[(long)c * R + r] = tile[threadIdx.x][j];
    }
}

struct S2SDims { int B, T, U1, H, E, KS, K; };

struct S2SLayout {
    size_t gi, gh, sx, logits, att, gemm, colsum;                                 // forward (+ shared)
    size_t dOIN, DGI, DGH, DIX, d_ox, d_hprev, d_ax0, d_ax1, g_cw, g_cb, g_nw, g_nb, wihT, whhT;  // backward
    size_t total;
};

S2SLayout s2s_layout(const S2SDims& d) {
    S2SLayout L;
    size_t p = 0;
    auto take = [&](size_t nfloat) { size_t o = p; p += sa_align_up(nfloat * sizeof(float), 256); return o; };
    const size_t BH = (size_t)d.B * d.H, UB = (size_t)d.U1 * d.B;
    L.gi = take(3 * BH); L.gh = take(3 * BH); L.sx = take(BH); L.logits = take((size_t)d.B * d.K);
    L.att = p; p += sa_align_up(sa_attention_workspace_bytes(d.B, d.T, d.H, d.KS), 256);
    size_t gw = sa_gemm_workspace_bytes((int)UB, d.K, d.H);
    const size_t c1 = sa_gemm_workspace_bytes(d.K, d.H, (int)UB), c2 = sa_gemm_workspace_bytes((int)UB, d.H, d.K);
    const size_t c3 = sa_gemm_workspace_bytes(3 * d.H, d.E, (int)UB), c4 = sa_gemm_workspace_bytes(3 * d.H, d.H, (int)UB);
    gw = gw > c1 ? gw : c1; gw = gw > c2 ? gw : c2; gw = gw > c3 ? gw : c3; gw = gw > c4 ? gw : c4;
    L.gemm = p; p += sa_align_up(gw, 256);
    size_t cs = sa_colsum_workspace_bytes((int)UB, 3 * d.H);
    const size_t cs2 = sa_colsum_workspace_bytes((int)UB, d.K), cs3 = sa_colsum_workspace_bytes(d.B, d.H * d.KS);
    cs = cs > cs2 ? cs : cs2; cs = cs > cs3 ? cs : cs3;
    L.colsum = p; p += sa_align_up(cs, 256);
    L.dOIN = take(UB * d.H); L.DGI = take(UB * 3 * d.H); L.DGH = take(UB * 3 * d.H); L.DIX = take(UB * d.E);
    L.d_ox = take(BH); L.d_hprev = take(BH); L.d_ax0 = take((size_t)d.B * d.T); L.d_ax1 = take((size_t)d.B * d.T);
    L.g_cw = take(BH * d.KS); L.g_cb = take(BH); L.g_nw = take(BH); L.g_nb = take(d.B);
    L.wihT = take((size_t)3 * d.H * d.E); L.whhT = take((size_t)3 * d.H * d.H);
    L.total = p;
    return L;
}

bool s2s_ok(const S2SDims& d) {
    return att_ok(d.B, d.T, d.H, d.KS) && d.U1 > 0 && d.E > 0 && d.K > 0 && (d.H & 3) == 0 && (d.E & 3) == 0 && d.E == d.H;
}

enum { P_EMB, P_WIH, P_WHH, P_BIH, P_BHH, P_CW, P_CB, P_NW, P_NB, P_FCW, P_FCB, P_COUNT };

}  // namespace

extern "C" size_t sa_s2s_decoder_workspace_bytes(int B, int T, int U1, int H, int E, int KS, int K) {
    S2SDims d{B, T, U1, H, E, KS, K};
    if (!s2s_ok(d)) return 0;
    return s2s_layout(d).total;
}

#define S2S_CHECK(call)                                          \
    do {                                                         \
        ctcStatus_t st__ = (call);                               \
        if (st__ != CTC_STATUS_SUCCESS) return st__;             \
    } while (0)

extern "C" ctcStatus_t sa_s2s_decoder_fwd(const float* eh, const long long* y, const unsigned char* sample,
                                          const float* const* params, int B, int T, int U, int H, int E, int KS, int K,
                                          float scale, float* out, long long* IDX, float* IX, float* ST, float* HX,
                                          float* AX, float* OIN, void* workspace, size_t workspace_bytes,
                                          void* stream_) {
    SA_CLEAR_ERR();
    S2SDims d{B, T, U - 1, H, E, KS, K};
    if (!eh || !y || !params || !out || !IDX || !IX || !ST || !HX || !AX || !OIN || !workspace || !s2s_ok(d))
        return CTC_STATUS_INVALID_VALUE;
    const S2SLayout L = s2s_layout(d);
    if (workspace_bytes < L.total) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* gi = (float*)(ws + L.gi);
    float* gh = (float*)(ws + L.gh);
    float* sx = (float*)(ws + L.sx);
    float* logits = (float*)(ws + L.logits);
    float* hzero = (float*)(ws + L.d_hprev);  // a zero state for the first token (the backward region is free here)
    if (hipMemsetAsync(hzero, 0, (size_t)B * H * sizeof(float), stream) != hipSuccess) return CTC_STATUS_MEMOPS_FAILED;
    const float* const* P = params;
    const size_t att_bytes = sa_attention_workspace_bytes(B, T, H, KS);
    const int U1 = U - 1;
    const size_t smem1 = ((size_t)(kAttTB + KS - 1) + (size_t)H * KS) * sizeof(float);
    const size_t smem2 = ((size_t)T + 8 + 4 * (size_t)H) * sizeof(float);
    if (!att_smem((const void*)attention_score_kernel, smem1) || !att_smem((const void*)attention_context_kernel, smem2))
        return CTC_STATUS_INVALID_VALUE;
    (void)att_bytes;
    // (r4) three launches per token instead of six: the projections; the score network, which forms the GRUCell gates itself;
    // softmax + context, which also forms the NEXT token's index and embedding + context row when that token is teacher
    // forced.  SA_S2S_FUSE=0: the six-launch loop.
    const bool fuse = true;  // (the GRUCell gates inside the score kernel; the unfused launches were SA_S2S_FUSE=0, retired in round 5)
    const size_t smem1f = smem1 + (fuse ? (size_t)H * sizeof(float) : 0);
    if (fuse && !att_smem((const void*)attention_score_kernel, smem1f)) return CTC_STATUS_INVALID_VALUE;
    bool have_ix = false;   // token t's idx / ix were produced by token t-1's context kernel
    for (int t = 0; t < U1; ++t) {
        long long* idx = IDX + (long)t * B;
        float* ix = IX + (long)t * B * E;
        float* hx = HX + (long)t * B * H;
        const float* hprev = t > 0 ? HX + (long)(t - 1) * B * H : hzero;
        if (t > 0 && sample && sample[t]) {  // scheduled sampling: feed back the argmax of the previous token's logits
            SkinnyProb q{OIN + (long)(t - 1) * B * H, P[P_FCW], P[P_FCB], logits, K, H, 0, H, H, K};
            skinny_launch(&q, 1, B, stream);
            hipLaunchKernelGGL(argmax_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, logits, idx, (long)B, K);
        } else if (!have_ix) {
            hipLaunchKernelGGL(s2s_idx_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, y, idx, B, U, t);
        }
        if (!have_ix)
            hipLaunchKernelGGL(s2s_embed_add_kernel, dim3(B), dim3(256), 0, stream, P[P_EMB], idx,
                               t > 0 ? (const float*)sx : (const float*)nullptr, ix, E);
        SkinnyProb pr[2] = {{ix, P[P_WIH], P[P_BIH], gi, 3 * H, E, 0, E, E, 3 * H},
                            {hprev, P[P_WHH], P[P_BHH], gh, 3 * H, H, 0, H, H, 3 * H}};
        skinny_launch(pr, 2, B, stream);
        if (!fuse)
            hipLaunchKernelGGL(grucell_gates_fwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, stream, gi, gh, hprev, hx,
                               ST + (long)t * B * 4 * H, B, H);
        AttArgs A{eh, hx, t > 0 ? AX + (long)(t - 1) * B * T : nullptr, P[P_CW], P[P_CB], P[P_NW], P[P_NB], scale,
                  B, T, H, KS};
        have_ix = false;
        if (fuse) {
            A.gi = gi; A.gh = gh; A.h_prev = hprev; A.ox_out = hx; A.gate_stash = ST + (long)t * B * 4 * H;
            if (t + 1 < U1 && !(sample && sample[t + 1])) {
                A.y_next = y + (t + 1); A.y_stride = U; A.emb = P[P_EMB];
                A.ix_next = IX + (long)(t + 1) * B * E; A.idx_next = IDX + (long)(t + 1) * B;
                have_ix = true;
            }
        }
        float* score = (float*)(ws + L.att);
        score_launch(A, score, B, fuse ? smem1f : smem1, stream);
        context_launch(A, score, AX + (long)t * B * T, sx, OIN + (long)t * B * H, B, smem2, stream);
    }
    SA_CHECK_LAUNCH();
    return sa_gemm_f32_impl(0, 1, U1 * B, K, H, 1.0f, OIN, H, P[P_FCW], H, 0.0f, out, K, P[P_FCB], nullptr,
                            ws + L.gemm, L.colsum - L.gemm, stream);
}

// One decoder token without the training stashes (Seq2Seq.decode_step, seq2seq.py:114-138): the inference / beam-search
// step.  Same kernels and the same arithmetic as one iteration of sa_s2s_decoder_fwd: embedding (+ previous context),
// both GRU projections in ONE skinny launch, the gate kernel, attention score + context, fc as a skinny product.
//   idx (B) int64 tokens; hprev / ax_prev / sx_prev null for the first token (zero state, no location features).
//   out (B,K) logits; hx (B,H), ax (B,T), sx (B,H) the new state.  workspace: sa_s2s_decoder_workspace_bytes(.., U1=1, ..).
extern "C" ctcStatus_t sa_s2s_decoder_step(const float* eh, const long long* idx, const float* hprev,
                                           const float* ax_prev, const float* sx_prev, const float* const* params,
                                           int B, int T, int H, int E, int KS, int K, float scale, float* hx, float* ax,
                                           float* sx, float* out, void* workspace, size_t workspace_bytes,
                                           void* stream_) {
    SA_CLEAR_ERR();
    S2SDims d{B, T, 1, H, E, KS, K};
    if (!eh || !idx || !params || !hx || !ax || !sx || !out || !workspace || !s2s_ok(d)) return CTC_STATUS_INVALID_VALUE;
    if ((hprev == nullptr) != (ax_prev == nullptr) || (hprev == nullptr) != (sx_prev == nullptr))
        return CTC_STATUS_INVALID_VALUE;
    const S2SLayout L = s2s_layout(d);
    if (workspace_bytes < L.total) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* gi = (float*)(ws + L.gi);
    float* gh = (float*)(ws + L.gh);
    float* ix = (float*)(ws + L.DIX);    // the backward regions are free in a forward-only step
    float* oin = (float*)(ws + L.dOIN);
    float* score = (float*)(ws + L.att);
    const float* const* P = params;
    if (!hprev) {
        float* hzero = (float*)(ws + L.d_hprev);
        if (hipMemsetAsync(hzero, 0, (size_t)B * H * sizeof(float), stream) != hipSuccess) return CTC_STATUS_MEMOPS_FAILED;
        hprev = hzero;
    }
    const size_t smem1 = ((size_t)(kAttTB + KS - 1) + (size_t)H * KS) * sizeof(float);
    const size_t smem2 = ((size_t)T + 8 + 4 * (size_t)H) * sizeof(float);
    if (!att_smem((const void*)attention_score_kernel, smem1) || !att_smem((const void*)attention_context_kernel, smem2))
        return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(s2s_embed_add_kernel, dim3(B), dim3(256), 0, stream, P[P_EMB], idx, sx_prev, ix, E);
    SkinnyProb pr[2] = {{ix, P[P_WIH], P[P_BIH], gi, 3 * H, E, 0, E, E, 3 * H},
                        {hprev, P[P_WHH], P[P_BHH], gh, 3 * H, H, 0, H, H, 3 * H}};
    skinny_launch(pr, 2, B, stream);
    hipLaunchKernelGGL(grucell_gates_fwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, stream, gi, gh, hprev, hx,
                       (float*)nullptr, B, H);
    AttArgs A{eh, hx, ax_prev, P[P_CW], P[P_CB], P[P_NW], P[P_NB], scale, B, T, H, KS};
    score_launch(A, score, B, smem1, stream);
    context_launch(A, score, ax, sx, oin, B, smem2, stream);
    SkinnyProb q{oin, P[P_FCW], P[P_FCB], out, K, H, 0, H, H, K};
    skinny_launch(&q, 1, B, stream);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_s2s_decoder_bwd(const float* eh, const float* const* params, const float* d_out,
                                          const long long* IDX, const float* IX, const float* ST, const float* HX,
                                          const float* AX, const float* OIN, int B, int T, int U, int H, int E, int KS,
                                          int K, int V, float scale, float* d_eh, float* const* grads, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    S2SDims d{B, T, U - 1, H, E, KS, K};
    if (!eh || !params || !d_out || !IDX || !IX || !ST || !HX || !AX || !OIN || !d_eh || !grads || !workspace ||
        !s2s_ok(d) || V <= 0)
        return CTC_STATUS_INVALID_VALUE;
    const S2SLayout L = s2s_layout(d);
    if (workspace_bytes < L.total) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    const float* const* P = params;
    const int U1 = U - 1, UB = U1 * B;
    float* dOIN = (float*)(ws + L.dOIN);
    float* DGI = (float*)(ws + L.DGI);
    float* DGH = (float*)(ws + L.DGH);
    float* DIX = (float*)(ws + L.DIX);
    float* d_ox = (float*)(ws + L.d_ox);
    float* d_hprev = (float*)(ws + L.d_hprev);
    float* d_ax[2] = {(float*)(ws + L.d_ax0), (float*)(ws + L.d_ax1)};
    float* g_cw = (float*)(ws + L.g_cw);
    float* wihT = (float*)(ws + L.wihT);
    float* whhT = (float*)(ws + L.whhT);
    void* gws = ws + L.gemm;
    const size_t gbytes = L.colsum - L.gemm;
    void* cws = ws + L.colsum;
    const size_t cbytes = L.dOIN - L.colsum;

    // everything that does not depend on the token loop: fc gradients and the gradient entering every token
    S2S_CHECK(sa_gemm_f32_impl(1, 0, K, H, UB, 1.0f, d_out, K, OIN, H, 0.0f, grads[P_FCW], H, nullptr, nullptr, gws,
                               gbytes, stream));
    S2S_CHECK(sa_colsum_f32(d_out, K, UB, K, grads[P_FCB], 0, cws, cbytes, stream));
    S2S_CHECK(sa_gemm_f32_impl(0, 0, UB, H, K, 1.0f, d_out, K, P[P_FCW], H, 0.0f, dOIN, H, nullptr, nullptr, gws, gbytes,
                               stream));
    // zero: d_eh, the running state gradient, the per-utterance attention-parameter partials (g_cw .. g_nb contiguous)
    if (hipMemsetAsync(d_eh, 0, (size_t)B * T * H * sizeof(float), stream) != hipSuccess ||
        hipMemsetAsync(d_hprev, 0, (size_t)B * H * sizeof(float), stream) != hipSuccess ||
        hipMemsetAsync(g_cw, 0, L.wihT - L.g_cw, stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    // W^T once, so that the per-token input gradients are k-contiguous "NT" products too
    hipLaunchKernelGGL(s2s_transpose_kernel, dim3((E + 31) / 32, (3 * H + 31) / 32), dim3(32, 8), 0, stream, P[P_WIH],
                       wihT, 3 * H, E);
    hipLaunchKernelGGL(s2s_transpose_kernel, dim3((H + 31) / 32, (3 * H + 31) / 32), dim3(32, 8), 0, stream, P[P_WHH],
                       whhT, 3 * H, H);
    const int nchunk = (T + kAttTB - 1) / kAttTB;
    size_t ao[4];
    att_layout(B, T, H, KS, ao);
    char* aws = ws + L.att;
    const size_t smem = ((size_t)(kAttTB + KS - 1) + (size_t)H * KS + 2 * kAttTB + (size_t)kAttTB * (H + 1)) * sizeof(float);
    if (!att_smem((const void*)attention_bwd_main_kernel, smem)) return CTC_STATUS_INVALID_VALUE;
    const bool fuse_b = true;
    const size_t smem2 = ((size_t)kAxp2 + (size_t)H * KS + kRed2 + (size_t)kAttTB * (H + 1) + 4 * 256) * sizeof(float);
    const bool one_launch = H <= 256 && (sa_opt(SA_OPT_S2S_KERNELS) & 1) != 0 &&
                            att_smem((const void*)attention_bwd_main2_kernel, smem2);
    for (int t = U1 - 1; t >= 0; --t) {
        const bool has_next = t + 1 < U1;
        const float* hx = HX + (long)t * B * H;
        AttArgs A{eh, hx, t > 0 ? AX + (long)(t - 1) * B * T : nullptr, P[P_CW], P[P_CB], P[P_NW], P[P_NB], scale,
                  B, T, H, KS};
        // the context of token t fed the fc (dOIN[t]) and the next token's GRU input (DIX[t + 1])
        AttBwd G{AX + (long)t * B * T, dOIN + (long)t * B * H, has_next ? DIX + (long)(t + 1) * B * E : nullptr,
                 has_next ? d_ax[(t + 1) & 1] : nullptr, d_eh, d_ox, t > 0 ? d_ax[t & 1] : nullptr, g_cw,
                 (float*)(ws + L.g_cb), (float*)(ws + L.g_nw), (float*)(ws + L.g_nb), (float*)(aws + ao[0]),
                 (float*)(aws + ao[1]), (float*)(aws + ao[2]), nchunk};
        // the state of token t fed the fc, the attention and the next token's GRU
        float* dgi = DGI + (long)t * B * 3 * H;
        float* dgh = DGH + (long)t * B * 3 * H;
        if (fuse_b) {   // (r4) four launches per token instead of six: the softmax inside the score-network kernel, the gate
                        // gradients inside the fold kernel's d_ox block
            G.fused_softmax = 1;
            G.gb_dh_a = dOIN + (long)t * B * H; G.gb_stash = ST + (long)t * B * 4 * H;
            G.gb_h_prev = t > 0 ? HX + (long)(t - 1) * B * H : nullptr;
            G.gb_dgi = dgi; G.gb_dgh = dgh; G.gb_dh_prev = d_hprev;
        }
        if (one_launch) {   // (r6) stages 1 - 3 in one launch: three launches per token
            G.oin = OIN + (long)t * B * H;
            G.sb_chunks = 1;
            hipLaunchKernelGGL(attention_bwd_main2_kernel, dim3(nchunk, B), dim3(256), smem2, stream, A, G);
        } else {
            hipLaunchKernelGGL(attention_bwd_dax_kernel, dim3(nchunk, B), dim3(256), 0, stream, A, G);
            if (!fuse_b) hipLaunchKernelGGL(attention_bwd_softmax_kernel, dim3(B), dim3(256), 0, stream, A, G);
            hipLaunchKernelGGL(attention_bwd_main_kernel, dim3(nchunk, B), dim3(256), smem, stream, A, G);
        }
        hipLaunchKernelGGL(attention_bwd_fold_kernel, dim3(2 + KS + 1, B), dim3(256), 0, stream, A, G);
        if (!fuse_b)
            hipLaunchKernelGGL(grucell_gates_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, stream,
                               dOIN + (long)t * B * H, (const float*)d_ox, (const float*)d_hprev, ST + (long)t * B * 4 * H,
                               t > 0 ? HX + (long)(t - 1) * B * H : (const float*)nullptr, dgi, dgh, d_hprev, B, H);
        SkinnyProb pr[2] = {{dgh, whhT, nullptr, d_hprev, H, 3 * H, 1, 3 * H, 3 * H, H},
                            {dgi, wihT, nullptr, DIX + (long)t * B * E, E, 3 * H, 0, 3 * H, 3 * H, E}};
        skinny_launch(pr, 2, B, stream);
    }
    SA_CHECK_LAUNCH();
    // batched over all tokens: weight / bias / embedding gradients, attention parameters summed over utterances
    S2S_CHECK(sa_gemm_f32_impl(1, 0, 3 * H, E, UB, 1.0f, DGI, 3 * H, IX, E, 0.0f, grads[P_WIH], E, nullptr, nullptr, gws,
                               gbytes, stream));
    if (U1 > 1) {
        S2S_CHECK(sa_gemm_f32_impl(1, 0, 3 * H, H, UB - B, 1.0f, DGH + (long)B * 3 * H, 3 * H, HX, H, 0.0f, grads[P_WHH],
                                   H, nullptr, nullptr, gws, gbytes, stream));
    } else if (hipMemsetAsync(grads[P_WHH], 0, (size_t)3 * H * H * sizeof(float), stream) != hipSuccess) {
        return CTC_STATUS_MEMOPS_FAILED;
    }
    S2S_CHECK(sa_colsum_f32(DGI, 3 * H, UB, 3 * H, grads[P_BIH], 0, cws, cbytes, stream));
    S2S_CHECK(sa_colsum_f32(DGH, 3 * H, UB, 3 * H, grads[P_BHH], 0, cws, cbytes, stream));
    S2S_CHECK(sa_embedding_bwd(DIX, IDX, grads[P_EMB], UB, E, V, stream));
    S2S_CHECK(sa_colsum_f32(g_cw, (long)H * KS, B, H * KS, grads[P_CW], 0, cws, cbytes, stream));
    S2S_CHECK(sa_colsum_f32((float*)(ws + L.g_cb), H, B, H, grads[P_CB], 0, cws, cbytes, stream));
    S2S_CHECK(sa_colsum_f32((float*)(ws + L.g_nw), H, B, H, grads[P_NW], 0, cws, cbytes, stream));
    S2S_CHECK(sa_colsum_f32((float*)(ws + L.g_nb), 1, B, 1, grads[P_NB], 0, cws, cbytes, stream));
    return CTC_STATUS_SUCCESS;
}

// =====================================================================================================================
// Beam search of the attention decoder (Seq2Seq.beam_search, seq2seq.py:180-227) entirely on the device.
//
// The reference keeps python lists of (hypothesis tuple, score, state) and runs one decode_step per live hypothesis.
// Here the live hypotheses are the ROWS of one batched decoder step -- they all attend over the same encoder states
// (AttArgs::eh_shared), so nothing is replicated -- and what the reference does on the host between two steps is ONE
// kernel (`s2s_beam_select_kernel`): log-softmax of the fc rows, candidate scores, the two selections, the
// complete / live bookkeeping, the stopping rule, and the hand-over to the next token (the survivors' states gathered
// by parent, their next embedding + context row formed).  Hypotheses are back-pointers (token, parent slot) per
// search step; only the winner is ever materialised (`s2s_beam_final_kernel`).  No device -> host traffic per token:
// the host loop enqueues `check_every` tokens (6 launches each), then reads ONE word (the search's `done` flag);
// with check_every = 0 it never synchronises and enqueues all max_len tokens (a finished search ignores them).
//
// Reproduced from the reference, in its terms (the test suite holds a CPU restatement pinned to the live reference):
//   * candidates of a step are ordered (live rank r, class i) -- index r * K + i -- and sorted by score descending with
//     that order as the tie break (python's stable sort, :206);
//   * a score is a DOUBLE sum of float32 log-probabilities (python float + float32 -> float, :203);
//   * `complete` receives the end-token candidates among the first beam_size of the sorted list, in sorted order (:209-211);
//   * the next beam is the first beam_size non-end candidates of the WHOLE sorted list (:213-214);
//   * stop: empty beam, or beam_size complete hypotheses strictly better than the best live one (:216-223), or max_len;
//   * answer: the best complete hypothesis, the earliest appended among equals (stable sort, :225); the best live
//     hypothesis if nothing completed (:226-227).
// Only the beam_size best complete SCORES are kept for the stopping rule (the count of completes above a threshold
// reaches beam_size iff the beam_size-th best does), and only the best complete hypothesis is remembered.
// =====================================================================================================================
namespace {

constexpr int kBeamMaxW = 64;       // beam_size limit: the selection kernel gives every live hypothesis one lane of a
                                    // wave (the reference's default is 10; it takes any width)
constexpr int kBeamLdsCand = 8192;  // up to this many candidates (beam_size * classes) their scores sit in LDS as doubles
                                    // (64 KB); beyond it -- a word-piece vocabulary -- in the workspace

struct BeamState {
    int done, n_live, n_complete, steps;
    int best_step, best_parent, pad0, pad1;
    double best_score;
    double scores[kBeamMaxW];   // live hypotheses' scores, in rank order
    double cs[kBeamMaxW];       // the beam_size best complete scores, descending (-inf = none)
};

struct BeamBufs {
    BeamState* st;
    int* tok_hist;     // [max_len][W] token of the hypothesis that became slot j after step t
    int* par_hist;     // [max_len][W] its parent's slot in the beam that ENTERED step t
    const float* logits;  // (W, K) this step's fc rows
    const float* hx_cur;  // (W, H) state rows written by this step ...
    const float* ax_cur;  // (W, T)
    const float* sx_cur;  // (W, H)
    float* h_prev;        // (W, H) ... gathered by parent for the next step
    float* ax_prev;       // (W, T)
    float* ix;            // (W, E) next token's GRU input: embedding + context
    const float* emb;     // (V, E)
    int W, K, T, H, E, end_tok, step, max_len;
    double* cand_mem;     // (W, K) candidate scores when they do not fit the LDS budget, else null
};

__global__ void s2s_beam_init_kernel(BeamState* st, float* ix, float* h_prev, const float* __restrict__ emb, int start_tok,
                                     int W, int E, int H) {
    const int j = blockIdx.x;
    for (int e = threadIdx.x; e < E; e += blockDim.x) ix[(long)j * E + e] = emb[(long)start_tok * E + e];
    for (int h = threadIdx.x; h < H; h += blockDim.x) h_prev[(long)j * H + h] = 0.f;
    if (j == 0 && threadIdx.x == 0) {
        st->done = 0; st->n_live = 1; st->n_complete = 0; st->steps = 0;
        st->best_step = -1; st->best_parent = -1; st->best_score = -INFINITY;
        for (int k = 0; k < kBeamMaxW; ++k) { st->scores[k] = 0.0; st->cs[k] = -INFINITY; }
    }
}

// does candidate (s, i) come before candidate (e, ei) in the reference's sorted list?
__device__ __forceinline__ bool beam_precedes(double s, int i, double e, int ei) { return s > e || (s == e && i < ei); }

// one workgroup of 256 threads; dynamic LDS: cand[W * K] doubles (or none, and the scores live in Q.cand_mem: one
// workgroup reads back what it wrote before a barrier, so plain global memory does)
__global__ __launch_bounds__(256) void s2s_beam_select_kernel(BeamBufs Q) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* cand = Q.cand_mem ? Q.cand_mem : reinterpret_cast<double*>(smem_raw);
    __shared__ double ends[kBeamMaxW], sel_s[kBeamMaxW], comp_s[kBeamMaxW];
    __shared__ int sel_i[kBeamMaxW], comp_r[kBeamMaxW];
    __shared__ int n_new_s;
    BeamState* st = Q.st;
    if (st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = Q.W, K = Q.K, n_live = st->n_live, N = W * K;
    // ---- A: log-softmax rows (float32, as nn.functional.log_softmax) and candidate scores (double) ----
    for (int r = wave; r < W; r += 4) {
        if (r < n_live) {
            const float* x = Q.logits + (long)r * K;
            float m = -INFINITY;
            for (int i = lane; i < K; i += 64) m = fmaxf(m, x[i]);
            m = sa_wave_max(m);
            float s = 0.f;
            for (int i = lane; i < K; i += 64) s += expf(x[i] - m);
            s = sa_wave_sum(s);
            const float ls = logf(s);
            const double base = st->scores[r];
            for (int i = lane; i < K; i += 64) {
                const float lp = (x[i] - m) - ls;
                const double c = base + (double)lp;
                if (i == Q.end_tok) { ends[r] = c; cand[r * K + i] = -INFINITY; }
                else cand[r * K + i] = c;
            }
        } else {
            for (int i = lane; i < K; i += 64) cand[r * K + i] = -INFINITY;
            if (lane == 0) ends[r] = -INFINITY;
        }
    }
    __syncthreads();
    if (wave == 0) {
        // ---- B: the first W non-end candidates of the sorted list ----
        double bs; int bi;
        auto scan = [&]() {
            bs = -INFINITY; bi = 0x7fffffff;
            for (int i = lane; i < N; i += 64) { const double s = cand[i]; if (s > bs) { bs = s; bi = i; } }
        };
        scan();
        int n_new = 0;
        for (int j = 0; j < W; ++j) {
            double s = bs; int i = bi;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double so = __shfl_xor(s, o, 64);
                const int io = __shfl_xor(i, o, 64);
                if (so > s || (so == s && io < i)) { s = so; i = io; }
            }
            if (!(s > -INFINITY)) break;  // fewer live candidates than W (wave-uniform)
            if (lane == 0) { sel_s[j] = s; sel_i[j] = i; }
            if ((i & 63) == lane) { cand[i] = -INFINITY; scan(); }
            n_new = j + 1;
        }
        // ---- C: end-token candidates inside the first W of the WHOLE sorted list become complete, in sorted order ----
        // rank of an end candidate = (non-end candidates before it: only the selected ones can matter) + (end candidates
        // before it)
        const int r = lane;
        const bool mine = r < n_live;
        const double e = mine ? ends[r] : -INFINITY;
        const int ei = r * K + Q.end_tok;
        int before = 0;
        for (int j = 0; j < n_new; ++j) before += beam_precedes(sel_s[j], sel_i[j], e, ei) ? 1 : 0;
        for (int q = 0; q < n_live; ++q)
            if (q != r) before += beam_precedes(ends[q], q * K + Q.end_tok, e, ei) ? 1 : 0;
        const bool is_c = mine && before < W;
        const unsigned long long cmask = __ballot(is_c);
        int pos = 0;
        for (int q = 0; q < n_live; ++q)
            if (q != r && ((cmask >> q) & 1)) pos += beam_precedes(ends[q], q * K + Q.end_tok, e, ei) ? 1 : 0;
        if (is_c) { comp_s[pos] = e; comp_r[pos] = r; }
        const int ncomp = __popcll(cmask);
        if (lane == 0) {
            int nc = st->n_complete;
            double best = st->best_score;
            for (int c = 0; c < ncomp; ++c) {
                const double sc = comp_s[c];
                if (nc == 0 || sc > best) { best = sc; st->best_step = Q.step; st->best_parent = comp_r[c]; }
                ++nc;
                int p = W;  // insert into the W best complete scores (equal scores behind the earlier ones)
                for (int k = 0; k < W; ++k) if (sc > st->cs[k]) { p = k; break; }
                if (p < W) {
                    for (int k = W - 1; k > p; --k) st->cs[k] = st->cs[k - 1];
                    st->cs[p] = sc;
                }
            }
            st->n_complete = nc;
            st->best_score = best;
            // ---- D: stopping rule ----
            int done = 0;
            if (n_new == 0) done = 1;
            else if (st->cs[W - 1] > sel_s[0]) done = 1;
            if (Q.step + 1 >= Q.max_len) done = 1;
            st->steps = Q.step + 1;
            st->n_live = n_new;
            n_new_s = n_new;
            // the flag is published last: a host that polls it sees a finished record
            __threadfence();
            st->done = done;
        }
    }
    __syncthreads();
    // ---- E: hand-over to the next token ----
    const int n_new = n_new_s;
    if (n_new == 0) return;
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        const int jj = j < n_new ? j : 0;
        Q.tok_hist[(long)Q.step * W + j] = sel_i[jj] % K;
        Q.par_hist[(long)Q.step * W + j] = sel_i[jj] / K;
        if (j < n_new) st->scores[j] = sel_s[j];
    }
    for (int j = 0; j < W; ++j) {
        const int jj = j < n_new ? j : 0;   // unused slots mirror slot 0: finite numbers, ignored by the next selection
        const int par = sel_i[jj] / K, tok = sel_i[jj] % K;
        for (int h = threadIdx.x; h < Q.H; h += blockDim.x) Q.h_prev[(long)j * Q.H + h] = Q.hx_cur[(long)par * Q.H + h];
        for (int t = threadIdx.x; t < Q.T; t += blockDim.x) Q.ax_prev[(long)j * Q.T + t] = Q.ax_cur[(long)par * Q.T + t];
        for (int e = threadIdx.x; e < Q.E; e += blockDim.x)
            Q.ix[(long)j * Q.E + e] = Q.emb[(long)tok * Q.E + e] + Q.sx_cur[(long)par * Q.E + e];
    }
}

// one wave: write the winner.  hyp (max_len + 1) int64, len, score, info = {steps, n_complete}
__global__ void s2s_beam_final_kernel(const BeamState* st, const int* __restrict__ tok_hist, const int* __restrict__ par_hist,
                                      int W, int start_tok, int end_tok, long long* __restrict__ hyp, int* __restrict__ len,
                                      double* __restrict__ score, int* __restrict__ info) {
    if (threadIdx.x != 0) return;
    int t, j, n;
    double sc;
    if (st->n_complete > 0) {       // (live hypothesis (best_step - 1, best_parent)) + end token
        n = st->best_step + 2;
        hyp[n - 1] = end_tok;
        t = st->best_step - 1; j = st->best_parent; sc = st->best_score;
    } else {                        // the best live hypothesis of the last step
        t = st->steps - 1; j = 0; n = t + 2; sc = st->scores[0];
    }
    for (; t >= 0; --t) {
        hyp[t + 1] = tok_hist[(long)t * W + j];
        j = par_hist[(long)t * W + j];
    }
    hyp[0] = start_tok;
    *len = n;
    *score = sc;
    info[0] = st->steps;
    info[1] = st->n_complete;
}

struct BeamLayout { size_t st, tok, par, ix, hprev, axprev, hx, ax, sx, oin, gi, gh, logits, score, cand, done_host, total; };

BeamLayout beam_layout(int T, int H, int E, int K, int W, int max_len) {
    BeamLayout L;
    size_t p = 0;
    auto take = [&](size_t bytes) { size_t o = p; p += sa_align_up(bytes, 256); return o; };
    const size_t f = sizeof(float);
    L.st = take(sizeof(BeamState));
    L.tok = take((size_t)max_len * W * sizeof(int)); L.par = take((size_t)max_len * W * sizeof(int));
    L.ix = take((size_t)W * E * f); L.hprev = take((size_t)W * H * f); L.axprev = take((size_t)W * T * f);
    L.hx = take((size_t)W * H * f); L.ax = take((size_t)W * T * f); L.sx = take((size_t)W * H * f);
    L.oin = take((size_t)W * H * f); L.gi = take((size_t)W * 3 * H * f); L.gh = take((size_t)W * 3 * H * f);
    L.logits = take((size_t)W * K * f); L.score = take((size_t)W * T * f);
    L.cand = take((long)W * K > kBeamLdsCand ? (size_t)W * K * sizeof(double) : 0);
    L.total = p;
    return L;
}

bool beam_ok(int T, int H, int E, int KS, int K, int W, int max_len) {
    S2SDims d{W, T, 1, H, E, KS, K};
    return s2s_ok(d) && W >= 1 && W <= kBeamMaxW && (long)W * K < (1L << 30) && max_len >= 1;
}

}  // namespace

extern "C" size_t sa_s2s_beam_workspace_bytes(int T, int H, int E, int KS, int K, int beam_size, int max_len) {
    if (!beam_ok(T, H, E, KS, K, beam_size, max_len)) return 0;
    return beam_layout(T, H, E, K, beam_size, max_len).total;
}

extern "C" ctcStatus_t sa_s2s_beam_search(const float* eh, const float* const* params, int T, int H, int E, int KS, int K,
                                          float scale, int start_tok, int end_tok, int beam_size, int max_len,
                                          int check_every, long long* d_hyp, int* d_len, double* d_score, int* d_info,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    const int W = beam_size;
    if (!eh || !params || !d_hyp || !d_len || !d_score || !d_info || !workspace || !beam_ok(T, H, E, KS, K, W, max_len) ||
        end_tok < 0 || end_tok >= K || start_tok < 0 || check_every < 0)
        return CTC_STATUS_INVALID_VALUE;
    const BeamLayout L = beam_layout(T, H, E, K, W, max_len);
    if (workspace_bytes < L.total) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    BeamState* st = (BeamState*)(ws + L.st);
    float* ix = (float*)(ws + L.ix);
    float* hprev = (float*)(ws + L.hprev);
    float* axprev = (float*)(ws + L.axprev);
    float* hx = (float*)(ws + L.hx);
    float* ax = (float*)(ws + L.ax);
    float* sx = (float*)(ws + L.sx);
    float* oin = (float*)(ws + L.oin);
    float* gi = (float*)(ws + L.gi);
    float* gh = (float*)(ws + L.gh);
    float* logits = (float*)(ws + L.logits);
    float* score = (float*)(ws + L.score);
    const float* const* P = params;
    const size_t smem1 = ((size_t)(kAttTB + KS - 1) + (size_t)H * KS + (size_t)H) * sizeof(float);
    const size_t smem2 = ((size_t)T + 8 + 4 * (size_t)H) * sizeof(float);
    const bool cand_in_lds = (long)W * K <= kBeamLdsCand;
    const size_t smem3 = cand_in_lds ? (size_t)W * K * sizeof(double) : 0;
    double* cand_mem = cand_in_lds ? nullptr : (double*)(ws + L.cand);
    if (!att_smem((const void*)attention_score_kernel, smem1) || !att_smem((const void*)attention_context_kernel, smem2) ||
        !att_smem((const void*)s2s_beam_select_kernel, smem3))
        return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(s2s_beam_init_kernel, dim3(W), dim3(256), 0, stream, st, ix, hprev, P[P_EMB], start_tok, W, E, H);
    for (int t = 0; t < max_len; ++t) {
        SkinnyProb pr[2] = {{ix, P[P_WIH], P[P_BIH], gi, 3 * H, E, 0, E, E, 3 * H},
                            {hprev, P[P_WHH], P[P_BHH], gh, 3 * H, H, 0, H, H, 3 * H}};
        skinny_launch(pr, 2, W, stream);
        AttArgs A{eh, hx, t > 0 ? (const float*)axprev : (const float*)nullptr, P[P_CW], P[P_CB], P[P_NW], P[P_NB], scale,
                  W, T, H, KS, 1};
        A.gi = gi; A.gh = gh; A.h_prev = hprev; A.ox_out = hx;   // the GRUCell gates inside the score kernel
        score_launch(A, score, W, smem1, stream);
        context_launch(A, score, ax, sx, oin, W, smem2, stream);
        SkinnyProb q{oin, P[P_FCW], P[P_FCB], logits, K, H, 0, H, H, K};
        skinny_launch(&q, 1, W, stream);
        BeamBufs Q{st, (int*)(ws + L.tok), (int*)(ws + L.par), logits, hx, ax, sx, hprev, axprev, ix, P[P_EMB],
                   W, K, T, H, E, end_tok, t, max_len, cand_mem};
        hipLaunchKernelGGL(s2s_beam_select_kernel, dim3(1), dim3(256), smem3, stream, Q);
        if (check_every > 0 && (t + 1) % check_every == 0 && t + 1 < max_len) {
            int done = 0;
            if (hipMemcpyAsync(&done, &st->done, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess)
                return CTC_STATUS_MEMOPS_FAILED;
            if (done) break;
        }
    }
    hipLaunchKernelGGL(s2s_beam_final_kernel, dim3(1), dim3(64), 0, stream, (const BeamState*)st, (const int*)(ws + L.tok),
                       (const int*)(ws + L.par), W, start_tok, end_tok, d_hyp, d_len, d_score, d_info);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

// =====================================================================================================================
// Greedy decode of a batch (Seq2Seq.infer -> infer_decode, seq2seq.py:140-178) as ONE library call: per token the
// decoder step of sa_s2s_decoder_step on ping-pong state buffers, then one kernel that takes each row's arg-max (first
// maximum, as torch.max), appends it to the token matrix, makes it the next input, and raises `done` when EVERY row
// emitted the end token in this step (the reference's stopping rule, :155-156).  The host polls `done` every check_every
// tokens (0: never; all max_len tokens are enqueued and a finished decode ignores the rest).
//   eh (B,T,H); d_tokens (B, max_len + 1) int64, column 0 = the start tokens on entry (seq2seq.py:172-175);
//   d_steps[0] = decoder steps run: the reference returns the first d_steps + 1 columns.
// =====================================================================================================================
namespace {

struct GreedyState { int done, steps; };

__global__ __launch_bounds__(64) void s2s_greedy_pick_kernel(GreedyState* st, const float* __restrict__ logits,
                                                            long long* __restrict__ tokens, long long* __restrict__ idx,
                                                            unsigned* __restrict__ count, int B, int K, int ncol, int step,
                                                            int end_tok, int max_len, const float* __restrict__ emb,
                                                            const float* __restrict__ sx, float* __restrict__ ix, int E) {
    if (st->done) return;
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* xr = logits + (long)b * K;
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
    // a row without a finite maximum (all NaN / all -inf: nothing ever compared greater) picks class 0, a valid index
    // like the one torch.max returns there, rather than reading the embedding 2^31 rows out of bounds
    if ((unsigned)bi >= (unsigned)K) bi = 0;
    for (int e = lane; e < E; e += 64) ix[(long)b * E + e] = emb[(long)bi * E + e] + sx[(long)b * E + e];  // the next token's input
    if (lane == 0) {
        tokens[(long)b * ncol + step + 1] = bi;
        idx[b] = bi;
        // the last row to arrive closes the step: count[step] = rows done, high half = rows that emitted the end token
        const unsigned add = 1u + (bi == end_tok ? 0x10000u : 0u);
        const unsigned seen = atomicAdd(&count[step], add) + add;
        if ((int)(seen & 0xffffu) == B) {
            st->steps = step + 1;
            __threadfence();
            if ((int)(seen >> 16) == B || step + 1 >= max_len) st->done = 1;
        }
    }
}

__global__ void s2s_greedy_init_kernel(GreedyState* st, unsigned* count, int max_len, const long long* __restrict__ tokens,
                                       long long* __restrict__ idx, float* __restrict__ hzero, int B, int H, int ncol) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { st->done = 0; st->steps = 0; }
    if (i < max_len) count[i] = 0;
    if (i < B) idx[i] = tokens[(long)i * ncol];
    for (int k = i; k < B * H; k += gridDim.x * blockDim.x) hzero[k] = 0.f;
}

struct GreedyLayout { size_t st, count, idx, ix, h[2], ax[2], sx[2], oin, gi, gh, logits, score, total; };

GreedyLayout greedy_layout(int B, int T, int H, int E, int K, int max_len) {
    GreedyLayout L;
    size_t p = 0;
    auto take = [&](size_t bytes) { size_t o = p; p += sa_align_up(bytes, 256); return o; };
    const size_t f = sizeof(float);
    L.st = take(sizeof(GreedyState)); L.count = take((size_t)max_len * sizeof(unsigned));
    L.idx = take((size_t)B * sizeof(long long)); L.ix = take((size_t)B * E * f);
    for (int k = 0; k < 2; ++k) { L.h[k] = take((size_t)B * H * f); L.ax[k] = take((size_t)B * T * f); L.sx[k] = take((size_t)B * H * f); }
    L.oin = take((size_t)B * H * f); L.gi = take((size_t)B * 3 * H * f); L.gh = take((size_t)B * 3 * H * f);
    L.logits = take((size_t)B * K * f); L.score = take((size_t)B * T * f);
    L.total = p;
    return L;
}

}  // namespace

extern "C" size_t sa_s2s_greedy_workspace_bytes(int B, int T, int H, int E, int KS, int K, int max_len) {
    S2SDims d{B, T, 1, H, E, KS, K};
    if (!s2s_ok(d) || max_len < 1 || B > 65535) return 0;
    return greedy_layout(B, T, H, E, K, max_len).total;
}

extern "C" ctcStatus_t sa_s2s_greedy_decode(const float* eh, const float* const* params, int B, int T, int H, int E, int KS,
                                            int K, float scale, int end_tok, int max_len, int check_every,
                                            long long* d_tokens, int* d_steps, void* workspace, size_t workspace_bytes,
                                            void* stream_) {
    SA_CLEAR_ERR();
    S2SDims d{B, T, 1, H, E, KS, K};
    if (!eh || !params || !d_tokens || !d_steps || !workspace || !s2s_ok(d) || max_len < 1 || B > 65535 || check_every < 0)
        return CTC_STATUS_INVALID_VALUE;
    const GreedyLayout L = greedy_layout(B, T, H, E, K, max_len);
    if (workspace_bytes < L.total) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    GreedyState* st = (GreedyState*)(ws + L.st);
    unsigned* count = (unsigned*)(ws + L.count);
    long long* idx = (long long*)(ws + L.idx);
    float* ix = (float*)(ws + L.ix);
    float* gi = (float*)(ws + L.gi);
    float* gh = (float*)(ws + L.gh);
    float* oin = (float*)(ws + L.oin);
    float* logits = (float*)(ws + L.logits);
    float* score = (float*)(ws + L.score);
    const float* const* P = params;
    const int ncol = max_len + 1;
    const size_t smem1 = ((size_t)(kAttTB + KS - 1) + (size_t)H * KS + (size_t)H) * sizeof(float);
    const size_t smem2 = ((size_t)T + 8 + 4 * (size_t)H) * sizeof(float);
    if (!att_smem((const void*)attention_score_kernel, smem1) || !att_smem((const void*)attention_context_kernel, smem2))
        return CTC_STATUS_INVALID_VALUE;
    // (the zero state of the first token lives in slot 1's h)
    hipLaunchKernelGGL(s2s_greedy_init_kernel, dim3(64), dim3(256), 0, stream, st, count, max_len, (const long long*)d_tokens,
                       idx, (float*)(ws + L.h[1]), B, H, ncol);
    for (int t = 0; t < max_len; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        float* hx = (float*)(ws + L.h[cur]);
        float* ax = (float*)(ws + L.ax[cur]);
        float* sx = (float*)(ws + L.sx[cur]);
        const float* hprev = (const float*)(ws + L.h[prv]);
        const float* ax_prev = t > 0 ? (const float*)(ws + L.ax[prv]) : nullptr;
        const float* sx_prev = t > 0 ? (const float*)(ws + L.sx[prv]) : nullptr;
        if (t == 0)   // (later tokens: the pick kernel forms emb[arg-max] + context itself)
            hipLaunchKernelGGL(s2s_embed_add_kernel, dim3(B), dim3(256), 0, stream, P[P_EMB], (const long long*)idx, sx_prev, ix, E);
        SkinnyProb pr[2] = {{ix, P[P_WIH], P[P_BIH], gi, 3 * H, E, 0, E, E, 3 * H},
                            {hprev, P[P_WHH], P[P_BHH], gh, 3 * H, H, 0, H, H, 3 * H}};
        skinny_launch(pr, 2, B, stream);
        AttArgs A{eh, hx, ax_prev, P[P_CW], P[P_CB], P[P_NW], P[P_NB], scale, B, T, H, KS};
        A.gi = gi; A.gh = gh; A.h_prev = hprev; A.ox_out = hx;   // the GRUCell gates inside the score kernel
        score_launch(A, score, B, smem1, stream);
        context_launch(A, score, ax, sx, oin, B, smem2, stream);
        SkinnyProb q{oin, P[P_FCW], P[P_FCB], logits, K, H, 0, H, H, K};
        skinny_launch(&q, 1, B, stream);
        hipLaunchKernelGGL(s2s_greedy_pick_kernel, dim3(B), dim3(64), 0, stream, st, (const float*)logits, d_tokens, idx, count, B,
                           K, ncol, t, end_tok, max_len, P[P_EMB], (const float*)sx, ix, E);
        if (check_every > 0 && (t + 1) % check_every == 0 && t + 1 < max_len) {
            int done = 0;
            if (hipMemcpyAsync(&done, &st->done, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess)
                return CTC_STATUS_MEMOPS_FAILED;
            if (done) break;
        }
    }
    if (hipMemcpyAsync(d_steps, &st->steps, sizeof(int), hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
