// gru.hip -- the GRU recurrence (forward and backward through time) for gfx950.
//
// Reference: nn.GRU inside Model.encode, /root/reference/speech/models/model.py:35-39,73 (cuDNN persistent RNN on
// the reference's GPU path).  Equations: SURVEY.md Appendix B, gate row order [r; z; n], b_hn inside r * (.).
// The input projections x W_ih^T + b_ih are one big MFMA GEMM outside (gemm_f32.hip); this file owns what is
// sequential: per time step  h_t = cell(ai_t, h_{t-1} W_hh^T + b_hh)  and its reverse-time gradient.
//
// One launch per time step ("job list" kernel: blockIdx.z selects one of up to 8 independent (layer, direction, t)
// jobs, so both directions of a bidirectional layer -- or a diagonal of a layer stack -- share a launch).
// A block owns a 16 (batch) x 16 (hidden units) output tile: the small-M product is computed on
// v_mfma_f32_16x16x4_f32 with the K dimension split over the block's 4 waves (one per SIMD), operands loaded
// straight from L2 into MFMA fragments with 16-byte loads (k-contiguous rows on both sides, so no LDS staging),
// partial tiles reduced through LDS, and the gate math fused into the epilogue.
//   forward : 3 gate tiles (r, z, n rows of W_hh) x K = H
//   backward: 1 tile of dh_{t-1} = dah_t W_hh (rows of W_hh^T, transposed once per call) x K = 3H, then the gate
//             gradients of step t-1 in the epilogue (they become the next launch's A operand).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxJobs = 8;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// acc[n] += A_tile(16 x kslice) * B_tile_n(16 x kslice)^T over this wave's K slice.
// 16x16x4 f32 operand layout: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]; each lane loads
// 4 consecutive k (one float4) from its row and feeds 4 MFMAs, the four 16-lane groups cover 16 consecutive k.
template <int NT>
__device__ __forceinline__ void smallm_mfma(const float* __restrict__ a_row, const float* const (&b_row)[NT],
                                            int kbeg, int kend, int K, f32x4 (&acc)[NT], int g) {
    for (int kk = kbeg; kk < kend; kk += 16) {  // wave-uniform trip count (MFMA must not sit in divergent flow)
        const int k = kk + 4 * g;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 b[NT];
        if (k + 3 < K) {
            a = *reinterpret_cast<const float4*>(a_row + k);
#pragma unroll
            for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const float4*>(b_row[n] + k);
        } else {
#pragma unroll
            for (int n = 0; n < NT; ++n) b[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k + 0 < K) { a.x = a_row[k];
#pragma unroll
                for (int n = 0; n < NT; ++n) b[n].x = b_row[n][k]; }
            if (k + 1 < K) { a.y = a_row[k + 1];
#pragma unroll
                for (int n = 0; n < NT; ++n) b[n].y = b_row[n][k + 1]; }
            if (k + 2 < K) { a.z = a_row[k + 2];
#pragma unroll
                for (int n = 0; n < NT; ++n) b[n].z = b_row[n][k + 2]; }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[n].x, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[n].y, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[n].z, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[n].w, acc[n], 0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------------ forward step
struct FwdJob {
    const float* ai;     // (B, T, 3H)
    const float* w_hh;   // (3H, H)
    const float* b_hh;   // (3H)
    float* h_out;        // [b * hs_b + t * hs_t + j]
    float* stash;        // (B, T, 5H) or null: r, z, n, q, h_{t-1}
    long hs_b, hs_t;
    int t, t_prev;       // t_prev < 0: first step (h_{t-1} = 0)
};
struct FwdJobs {
    int n, B, T, H;
    FwdJob j[kMaxJobs];
};

__global__ __launch_bounds__(256) void gru_fwd_step_kernel(FwdJobs P) {
    __shared__ float red[4][3][256];
    const FwdJob& J = P.j[blockIdx.z];
    const int H = P.H, B = P.B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;

    if (J.t_prev >= 0) {
        const int kslice = ((H + 63) / 64) * 16;  // per-wave K slice, multiple of 16
        const int kbeg = wave * kslice, kend = min(H, kbeg + kslice);
        const int brow = min(b0 + i, B - 1);
        const int urow = min(u0 + i, H - 1);
        const float* a_row = J.h_out + (long)brow * J.hs_b + (long)J.t_prev * J.hs_t;
        const float* b_rows[3] = {J.w_hh + (long)urow * H, J.w_hh + (long)(H + urow) * H,
                                  J.w_hh + (long)(2 * H + urow) * H};
        f32x4 acc[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        smallm_mfma<3>(a_row, b_rows, kbeg, kend, H, acc, g);
        // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][n][(g * 4 + r) * 16 + i] = acc[n][r];
    }
    __syncthreads();

    const int bi = threadIdx.x >> 4, uj = threadIdx.x & 15;
    const int b = b0 + bi, u = u0 + uj;
    if (b >= B || u >= H) return;
    float sr = 0.f, sz = 0.f, sn = 0.f, hp = 0.f;
    if (J.t_prev >= 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sr += red[w][0][threadIdx.x];
            sz += red[w][1][threadIdx.x];
            sn += red[w][2][threadIdx.x];
        }
        hp = J.h_out[(long)b * J.hs_b + (long)J.t_prev * J.hs_t + u];
    }
    const float* ai = J.ai + ((long)b * P.T + J.t) * 3 * H;
    const float r = sigmoidf_(ai[u] + sr + J.b_hh[u]);
    const float z = sigmoidf_(ai[H + u] + sz + J.b_hh[H + u]);
    const float q = sn + J.b_hh[2 * H + u];
    const float n = tanhf(ai[2 * H + u] + r * q);
    const float h = (1.0f - z) * n + z * hp;
    J.h_out[(long)b * J.hs_b + (long)J.t * J.hs_t + u] = h;
    if (J.stash) {
        float* s = J.stash + ((long)b * P.T + J.t) * 5 * H;
        s[u] = r; s[H + u] = z; s[2 * H + u] = n; s[3 * H + u] = q; s[4 * H + u] = hp;
    }
}

// ----------------------------------------------------------------------------------------------------- backward step
struct BwdJob {
    const float* dh_out;  // gradient wrt h_out, same strides
    const float* h_out;
    const float* stash;   // (B, T, 5H)
    const float* w_hh_t;  // (H, 3H): W_hh transposed
    float* dai;           // (B, T, 3H)
    float* dah;           // (B, T, 3H)
    float* dh_ping;       // (B, H) running dh of the step being produced (write)
    const float* dh_pong; // (B, H) running dh of the previously produced step (read)
    long hs_b, hs_t;
    int t;                // step whose gate gradients this launch produces
    int t_next;           // step processed by the previous launch (t+1 forward-in-time layers), < 0: none
    int t_prev;           // step feeding h_{t-1} into step t (t-1), < 0: h_{t-1} = 0
};
struct BwdJobs {
    int n, B, T, H;
    BwdJob j[kMaxJobs];
};

__global__ __launch_bounds__(256) void gru_bwd_step_kernel(BwdJobs P) {
    __shared__ float red[4][256];
    const BwdJob& J = P.j[blockIdx.z];
    const int H = P.H, B = P.B, H3 = 3 * P.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;

    if (J.t_next >= 0) {  // dh_t += dah_{t_next} W_hh   (K = 3H)
        const int kslice = ((H3 + 63) / 64) * 16;
        const int kbeg = wave * kslice, kend = min(H3, kbeg + kslice);
        const int brow = min(b0 + i, B - 1);
        const int urow = min(u0 + i, H - 1);
        const float* a_row = J.dah + ((long)brow * P.T + J.t_next) * H3;
        const float* b_rows[1] = {J.w_hh_t + (long)urow * H3};
        f32x4 acc[1];
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        smallm_mfma<1>(a_row, b_rows, kbeg, kend, H3, acc, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(g * 4 + r) * 16 + i] = acc[0][r];
    }
    __syncthreads();

    const int bi = threadIdx.x >> 4, uj = threadIdx.x & 15;
    const int b = b0 + bi, u = u0 + uj;
    if (b >= B || u >= H) return;
    float dh = J.dh_out[(long)b * J.hs_b + (long)J.t * J.hs_t + u];
    if (J.t_next >= 0) {
        dh += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const float z_next = J.stash[((long)b * P.T + J.t_next) * 5 * H + H + u];
        dh += J.dh_pong[(long)b * H + u] * z_next;
    }
    J.dh_ping[(long)b * H + u] = dh;
    const float* s = J.stash + ((long)b * P.T + J.t) * 5 * H;
    const float r = s[u], z = s[H + u], n = s[2 * H + u], q = s[3 * H + u], hp = s[4 * H + u];
    const float dn = dh * (1.0f - z);
    const float dz = dh * (hp - n);
    const float dpn = dn * (1.0f - n * n);
    const float dpr = dpn * q * r * (1.0f - r);
    const float dpz = dz * z * (1.0f - z);
    float* di = J.dai + ((long)b * P.T + J.t) * H3;
    float* dhh = J.dah + ((long)b * P.T + J.t) * H3;
    di[u] = dpr; di[H + u] = dpz; di[2 * H + u] = dpn;
    dhh[u] = dpr; dhh[H + u] = dpz; dhh[2 * H + u] = dpn * r;
}

// ------------------------------------------------------------------------------------------------------- small helpers
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < C) tile[k][tx] = in[(long)(r0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && r0 + tx < R) out[(long)(c0 + k) * R + r0 + tx] = tile[tx][k];
}

// out[n] (+)= sum_m a[m * lda + n]: block = 32 columns x 8 row-lanes, fixed reduction order (deterministic)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, long lda, int M, int N,
                                                     float* __restrict__ out, int accumulate) {
    __shared__ float red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (n < N)
        for (int m = ty; m < M; m += 8) s += a[(long)m * lda + n];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][tx];
        out[n] = accumulate ? out[n] + t : t;
    }
}

__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ a, long lda,
                                                       const float* __restrict__ b, long ldb, float* __restrict__ y,
                                                       long ldy, int rows, int cols) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * cols) return;
    const int r = (int)(idx / cols), c = (int)(idx % cols);
    y[(long)r * ldy + c] = a[(long)r * lda + c] + b[(long)r * ldb + c];
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------ host
extern "C" ctcStatus_t sa_gru_fwd(const float* ai, const float* w_hh, const float* b_hh, float* h_out, long hs_b,
                                  long hs_t, float* stash, int B, int T, int H, int reverse, void* stream_) {
    if (!ai || !w_hh || !b_hh || !h_out || B <= 0 || T <= 0 || H <= 0) return CTC_STATUS_INVALID_VALUE;
    if ((H & 3) || (hs_b & 3) || (hs_t & 3) || ((uintptr_t)h_out & 15) || ((uintptr_t)w_hh & 15))
        return CTC_STATUS_INVALID_VALUE;  // 16-byte fragment loads need 4-float alignment
    hipStream_t stream = (hipStream_t)stream_;
    FwdJobs P;
    P.n = 1; P.B = B; P.T = T; P.H = H;
    FwdJob& J = P.j[0];
    J.ai = ai; J.w_hh = w_hh; J.b_hh = b_hh; J.h_out = h_out; J.stash = stash; J.hs_b = hs_b; J.hs_t = hs_t;
    dim3 grid((H + 15) / 16, (B + 15) / 16, 1);
    for (int s = 0; s < T; ++s) {
        J.t = reverse ? T - 1 - s : s;
        J.t_prev = s == 0 ? -1 : (reverse ? J.t + 1 : J.t - 1);
        hipLaunchKernelGGL(gru_fwd_step_kernel, grid, dim3(256), 0, stream, P);
    }
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" size_t sa_gru_bwd_workspace_bytes(int B, int T, int H) {
    (void)T;
    if (B <= 0 || H <= 0) return 0;
    return sa_align_up((size_t)2 * B * H * sizeof(float), 256) + sa_align_up((size_t)3 * H * H * sizeof(float), 256);
}

extern "C" ctcStatus_t sa_gru_bwd(const float* dh_out, long hs_b, long hs_t, const float* h_out, const float* stash,
                                  const float* w_hh, float* dai, float* dah, int B, int T, int H, int reverse,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    if (!dh_out || !h_out || !stash || !w_hh || !dai || !dah || !workspace || B <= 0 || T <= 0 || H <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if ((H & 3) || ((uintptr_t)dah & 15) || workspace_bytes < sa_gru_bwd_workspace_bytes(B, T, H))
        return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    float* dh0 = (float*)workspace;
    float* dh1 = dh0 + (size_t)B * H;
    float* w_t = (float*)((char*)workspace + sa_align_up((size_t)2 * B * H * sizeof(float), 256));
    hipLaunchKernelGGL(transpose_kernel, dim3((H + 31) / 32, (3 * H + 31) / 32), dim3(256), 0, stream, w_hh, w_t,
                       3 * H, H);
    BwdJobs P;
    P.n = 1; P.B = B; P.T = T; P.H = H;
    BwdJob& J = P.j[0];
    J.dh_out = dh_out; J.h_out = h_out; J.stash = stash; J.w_hh_t = w_t; J.dai = dai; J.dah = dah;
    J.hs_b = hs_b; J.hs_t = hs_t;
    dim3 grid((H + 15) / 16, (B + 15) / 16, 1);
    for (int s = 0; s < T; ++s) {
        // forward-in-time layers are unwound from t = T-1 down to 0; reverse layers from t = 0 up to T-1
        J.t = reverse ? s : T - 1 - s;
        J.t_next = s == 0 ? -1 : (reverse ? J.t - 1 : J.t + 1);
        J.t_prev = reverse ? (J.t + 1 < T ? J.t + 1 : -1) : (J.t - 1);
        J.dh_ping = (s & 1) ? dh1 : dh0;
        J.dh_pong = (s & 1) ? dh0 : dh1;
        hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(256), 0, stream, P);
    }
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_colsum_f32(const float* a, long lda, int M, int N, float* out, int accumulate,
                                     void* stream_) {
    if (!a || !out || M < 0 || N <= 0) return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream_, a, lda, M, N, out,
                       accumulate);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_add_rows_f32(const float* a, long lda, const float* b, long ldb, float* y, long ldy,
                                       int rows, int cols, void* stream_) {
    if (!a || !b || !y || rows < 0 || cols < 0) return CTC_STATUS_INVALID_VALUE;
    const long total = (long)rows * cols;
    if (total == 0) return CTC_STATUS_SUCCESS;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a,
                       lda, b, ldb, y, ldy, rows, cols);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
