// joint_fused.hip -- the Transducer's joint network as three MFMA kernels that never materialise the joint tensor
// (/root/reference/speech/models/transducer_model.py:72-76):
//
//     z[b,t,u,:]    = relu(xa[b,t,:] + ya[b,u,:])          (B, T, U1, H)   <- 3.3 GB at B=32, T'=500, U1=101, H=512
//     logp[b,t,u,:] = log_softmax(W2 z[b,t,u,:] + b2)      (B, T, U1, K)   K = vocab + 1 (29)
//
// The unfused path (sa_joint_relu_* + sa_gemm_f32 + sa_log_softmax_*) writes z once and reads it twice, and writes
// dz once and reads it twice: ~20 GB of HBM traffic per step around 160 GFLOP of matrix work whose narrow side (29
// classes) fills a quarter of a 128-wide GEMM tile.  Here the MFMA operand IS the joint: a lane builds its fragment
// of z from two L2/LDS-resident rows (one add, one max) right before the v_mfma_f32_16x16x4_f32 that consumes it.
// HBM traffic drops to the lattice itself (logp and its gradient, 187 MB each) plus small partial-sum buffers.
//
//   forward       : a wave owns 16 prediction rows u (their ya fragment stays in registers) and walks frames t two at
//                   a time: C[u, class] += relu(xa[t] + ya[u]) W2^T, K = H; bias + log-softmax in the epilogue
//                   (16-lane DPP reductions: a lattice row's 29 classes sit in one 16-lane row of two accumulators).
//   backward data : dl = glp - exp(logp) sum(glp) (log-softmax gradient, rebuilt per row), dz = dl W2 (K = 32 padded
//                   classes), masked by [xa + ya > 0]; dya accumulates in registers over t, dxa[t] is the sum over the
//                   tile's 16 rows.  Partial sums over u-tiles / t-chunks go to a workspace and are folded by a
//                   fixed-order reduction (deterministic, no atomics).
//   backward weight: dW2 = dl^T z, db2 = sum dl; persistent blocks, each wave owns a quarter of the hidden axis and
//                   keeps its 32 x H/4 accumulator in registers over all its work items.
// Shapes: H % 64 == 0, H <= 512, K <= 32 (sa_joint_fused_workspace_bytes returns 0 otherwise and the caller uses the
// unfused kernels).  fp32 throughout; products are exact fp32 MFMA with fp32 accumulation, as in sa_gemm_f32.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKP = 32;        // classes padded to two 16-wide MFMA tiles
constexpr int kFwdTc = 64;     // frames per forward block (8 waves x 4 pairs)
constexpr int kDataTc = 64;    // frames per backward-data block (4 frame lanes x 16, each lane two half-width waves)
constexpr int kWeightTc = 16;  // frames per backward-weight work item
constexpr int kWeightGrid = 768;  // persistent backward-weight blocks: 3 per CU (161 registers per lane)
constexpr int kLdT = 36;       // row stride of the transposed W2 tile in LDS (conflict-free 16-byte reads)

struct JArgs {
    const float* xa;    // (B, T, H)
    const float* ya;    // (B, U1, H)
    const float* w2;    // (K, H)
    const float* b2;    // (K)
    float* logp;        // (B, T, U1, K)   forward: out;  backward: in
    const float* glp;   // (B, T, U1, K)   gradient w.r.t. logp
    float* part_x;      // [nut][B][T][H]
    float* part_y;      // [ntc][B][U1][H]
    float* part_w;      // [grid][32][H]
    float* part_b;      // [grid][32]
    int B, T, U1, H, K, nut, ntc;
};

#define JF_ROR(v_, n_) SA_DPP_F((v_), (v_), 0x120 + (n_), 0xf)  // row_ror:n -- rotate inside each 16-lane row
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, JF_ROR(v, 8));
    v = fmaxf(v, JF_ROR(v, 4));
    v = fmaxf(v, JF_ROR(v, 2));
    v = fmaxf(v, JF_ROR(v, 1));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += JF_ROR(v, 8);
    v += JF_ROR(v, 4);
    v += JF_ROR(v, 2);
    v += JF_ROR(v, 1);
    return v;
}
// sum over the four 16-lane rows (lanes l, l^16, l^32, l^48), result in every lane
__device__ __forceinline__ float rows4_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- forward -----------------------------------------------------------------------------------------------------
// grid: ntc * nut * B blocks of 8 waves; block (b, ut, tcb) covers rows u0..u0+15 and frames [tcb*kFwdTc, +kFwdTc).
// 16x16x4 operand layout: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15]; a lane's float4
// covers k = 16 j + 4 g + {0..3} and feeds four MFMAs (the four 16-lane groups together cover 16 consecutive k).
template <int NJ, bool EXACT>
__global__ __launch_bounds__(512) void joint_fused_fwd_kernel(JArgs a) {
    extern __shared__ float smem[];
    const int H = a.H, K = a.K, T = a.T, U1 = a.U1;
    const int nj = EXACT ? NJ : H >> 4, LDW = H + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    float* w_s = smem;                                   // [32][LDW], rows >= K zero
    float* x_s = smem + kKP * LDW + wave * 2 * H;         // this wave's two xa rows
    int bid = blockIdx.x;
    const int tcb = bid % a.ntc; bid /= a.ntc;
    const int ut = bid % a.nut;
    const int b = bid / a.nut;
    const int u0 = ut * 16;

    const int h4 = H >> 2;
    for (int idx = tid; idx < kKP * h4; idx += 512) {
        const int n = idx / h4, c4 = idx - n * h4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < K) v = *reinterpret_cast<const float4*>(a.w2 + (long)n * H + 4 * c4);
        *reinterpret_cast<float4*>(w_s + n * LDW + 4 * c4) = v;
    }
    float4 yf[NJ];
    {
        const float* yrow = a.ya + ((long)b * U1 + min(u0 + i, U1 - 1)) * H + 4 * g;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (EXACT || j < nj) yf[j] = *reinterpret_cast<const float4*>(yrow + 16 * j);
    }
    const bool valid0 = i < K, valid1 = 16 + i < K;
    const float bias0 = valid0 ? a.b2[i] : 0.f, bias1 = valid1 ? a.b2[16 + i] : 0.f;
    __syncthreads();

    const int t_beg = tcb * kFwdTc, t_end = min(T, t_beg + kFwdTc);
    const float4* xa4 = reinterpret_cast<const float4*>(a.xa) + (long)b * T * h4;
    float4* xs4 = reinterpret_cast<float4*>(x_s);
    // a pair's xa rows: H/4 float4 each, at most two per lane per row (H <= 512); fetched one pair ahead
    const int col0 = min(lane, h4 - 1), col1 = min(lane + 64, h4 - 1);
    float4 xr00, xr01, xr10, xr11;
#define JF_FETCH_ROWS(t0_)                                          \
    do {                                                            \
        const long r0_ = (long)(t0_) * h4;                          \
        const long r1_ = (long)((t0_) + 1 < t_end ? (t0_) + 1 : (t0_)) * h4; \
        xr00 = xa4[r0_ + col0]; xr01 = xa4[r0_ + col1];             \
        xr10 = xa4[r1_ + col0]; xr11 = xa4[r1_ + col1];             \
    } while (0)
    {
        const int tf = t_beg + 2 * wave < t_end ? t_beg + 2 * wave : t_beg;
        JF_FETCH_ROWS(tf);
    }
    for (int t0 = t_beg + 2 * wave; t0 < t_end; t0 += 16) {
        const bool two = t0 + 1 < t_end;
        const int t1 = two ? t0 + 1 : t0;
        // stage this pair's xa rows in the wave's LDS slot (LDS is in-order per wave: no barrier between the write
        // and the fragment reads below, nor between those reads and the next pair's writes)
        if (lane < h4) { xs4[lane] = xr00; xs4[h4 + lane] = xr10; }
        if (lane + 64 < h4) { xs4[lane + 64] = xr01; xs4[h4 + lane + 64] = xr11; }
        __builtin_amdgcn_wave_barrier();
        {
            const int tf = t0 + 16 < t_end ? t0 + 16 : t0;  // travels under this pair's 512 MFMAs
            JF_FETCH_ROWS(tf);
        }
        f32x4 acc[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (EXACT || j < nj) {
                const int k = 16 * j + 4 * g;
                const float4 w0 = *reinterpret_cast<const float4*>(w_s + i * LDW + k);
                const float4 w1 = *reinterpret_cast<const float4*>(w_s + (16 + i) * LDW + k);
                const float4 x0 = *reinterpret_cast<const float4*>(x_s + k);
                const float4 x1 = *reinterpret_cast<const float4*>(x_s + H + k);
                const float4 y = yf[j];
                const float a0x = fmaxf(x0.x + y.x, 0.f), a0y = fmaxf(x0.y + y.y, 0.f);
                const float a0z = fmaxf(x0.z + y.z, 0.f), a0w = fmaxf(x0.w + y.w, 0.f);
                const float a1x = fmaxf(x1.x + y.x, 0.f), a1y = fmaxf(x1.y + y.y, 0.f);
                const float a1z = fmaxf(x1.z + y.z, 0.f), a1w = fmaxf(x1.w + y.w, 0.f);
                acc[0][0] = mfma4(a0x, w0.x, acc[0][0]); acc[0][1] = mfma4(a0x, w1.x, acc[0][1]);
                acc[1][0] = mfma4(a1x, w0.x, acc[1][0]); acc[1][1] = mfma4(a1x, w1.x, acc[1][1]);
                acc[0][0] = mfma4(a0y, w0.y, acc[0][0]); acc[0][1] = mfma4(a0y, w1.y, acc[0][1]);
                acc[1][0] = mfma4(a1y, w0.y, acc[1][0]); acc[1][1] = mfma4(a1y, w1.y, acc[1][1]);
                acc[0][0] = mfma4(a0z, w0.z, acc[0][0]); acc[0][1] = mfma4(a0z, w1.z, acc[0][1]);
                acc[1][0] = mfma4(a1z, w0.z, acc[1][0]); acc[1][1] = mfma4(a1z, w1.z, acc[1][1]);
                acc[0][0] = mfma4(a0w, w0.w, acc[0][0]); acc[0][1] = mfma4(a0w, w1.w, acc[0][1]);
                acc[1][0] = mfma4(a1w, w0.w, acc[1][0]); acc[1][1] = mfma4(a1w, w1.w, acc[1][1]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // epilogue: C/D layout col = lane & 15 (class), row = 4 g + r (prediction row)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p == 1 && !two) break;  // wave-uniform
            const int t = p == 0 ? t0 : t1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v0 = valid0 ? acc[p][0][r] + bias0 : -INFINITY;
                const float v1 = valid1 ? acc[p][1][r] + bias1 : -INFINITY;
                const float m = row16_max(fmaxf(v0, v1));
                const float s = row16_sum(sa_exp2((v0 - m) * SA_LOG2E) + sa_exp2((v1 - m) * SA_LOG2E));
                const float lz = m + sa_log2(s) * SA_LN2;
                const int uu = u0 + 4 * g + r;
                if (uu < U1) {
                    float* o = a.logp + (((long)b * T + t) * U1 + uu) * K;
                    if (valid0) o[i] = v0 - lz;
                    if (valid1) o[16 + i] = v1 - lz;
                }
            }
        }
    }
}

#undef JF_FETCH_ROWS

// ---- backward, data ------------------------------------------------------------------------------------------------
// grid: ntc * nut * B blocks of 8 waves = 4 frame lanes x 2 hidden halves, two waves per SIMD (one wave alone cannot
// overlap its own masking VALU work with its own back-to-back MFMAs; measured 1.4 ms with one 512-register wave per
// SIMD).  Wave (tw, hh) of block (b, ut, tcb) walks frames t_beg + tw, + 4, ... over hidden tiles [hh NTH, (hh+1) NTH)
// and keeps that half of ya and of the dya accumulator (2 x NTH x 4 registers) resident.
//   MFMA: A[row u = lane & 15][k = class slot g] = dl[u][8 g + s] at step s, B[k][col] = W2[8 g + s][nt*16 + col].
// EXACT: H == 16 NT, so the per-tile guards fold away (they are wave-uniform branches that fence the scheduler).
template <int NT, bool EXACT>
__global__ __launch_bounds__(512) void joint_fused_bwd_data_kernel(JArgs a) {
    extern __shared__ float smem[];
    const int H = a.H, K = a.K, T = a.T, U1 = a.U1, B = a.B;
    const int nt_n = EXACT ? NT : H >> 4;
    constexpr int NTH = NT / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tw = wave >> 1, hh = wave & 1;
    const int i = lane & 15, g = lane >> 4;
    float* wt_s = smem;                       // [H][kLdT]: wt_s[h][n] = W2[n][h], classes >= K zero
    float* red_s = smem + H * kLdT;           // [16][H]
    float* x_s = red_s + 16 * H + wave * (NTH * 16);  // this wave's part of the current xa row
    int bid = blockIdx.x;
    const int tcb = bid % a.ntc; bid /= a.ntc;
    const int ut = bid % a.nut;
    const int b = bid / a.nut;
    const int u0 = ut * 16;
    const int t0n = hh * NTH;                 // first tile of this wave's half
    const int hoff = t0n * 16;

    for (int idx = tid; idx < kKP * H; idx += 512) {
        const int n = idx / H, h = idx - n * H;
        wt_s[h * kLdT + n] = n < K ? a.w2[(long)n * H + h] : 0.f;
    }
    f32x4 yc[NTH], dya[NTH];
#pragma unroll
    for (int k = 0; k < NTH; ++k) {
        dya[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        yc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EXACT || t0n + k < nt_n) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                yc[k][r] = a.ya[((long)b * U1 + min(u0 + 4 * g + r, U1 - 1)) * H + hoff + k * 16 + i];
        }
    }
    __syncthreads();

    const int t_beg = tcb * kDataTc, t_end = min(T, t_beg + kDataTc);
    const bool valid_u = u0 + i < U1;
    const int urow = min(u0 + i, U1 - 1);
    // per-lane lattice offsets (frame-independent part) of its 8 class slots
    bool ok[8];
    long coff[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        ok[s] = valid_u && 8 * g + s < K;
        coff[s] = ok[s] ? ((long)b * T * U1 + urow) * K + 8 * g + s : 0;
    }
    const long tstride = (long)U1 * K;
    // this wave's part of an xa row: hidden units [hoff, hoff + 16 NTH) clipped to H -- at most one float4 per lane
    const int xq4 = max(0, min(H - hoff, NTH * 16)) >> 2;
    const float4* xrow4 = reinterpret_cast<const float4*>(a.xa + (long)b * T * H + (xq4 > 0 ? hoff : 0));
    const int xcol = max(0, min(lane, xq4 - 1));
    float gl[8], lp[8];
    float4 xnext;
#define JF_LOAD_T(t_)                                                      \
    do {                                                                   \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                    \
            const long o = ok[s] ? coff[s] + (long)(t_) * tstride : 0;     \
            gl[s] = a.glp[o];                                              \
            lp[s] = a.logp[o];                                             \
        }                                                                  \
        xnext = xrow4[(long)(t_) * (H >> 2) + xcol];                       \
    } while (0)
    int t = t_beg + tw;
    if (t < t_end) JF_LOAD_T(t);
    for (; t < t_end; t += 4) {
        // log-softmax gradient of this tile's 16 lattice rows
        float ssum = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) ssum += ok[s] ? gl[s] : 0.f;
        ssum = rows4_sum(ssum);
        float dl[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) dl[s] = ok[s] ? gl[s] - sa_exp2(lp[s] * SA_LOG2E) * ssum : 0.f;
        if (lane < xq4) reinterpret_cast<float4*>(x_s)[lane] = xnext;
        __builtin_amdgcn_wave_barrier();
        // the next frame's operands travel while this frame's MFMAs run
        {
            const int tn = t + 4 < t_end ? t + 4 : t;
            JF_LOAD_T(tn);
        }
        float* px = a.part_x + (((long)ut * B + b) * T + t) * H + hoff + lane;
        constexpr int NG = NTH / 4 > 0 ? NTH / 4 : 1;
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (EXACT || t0n + grp * 4 < nt_n) {
                f32x4 acc[4];
                float4 w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w[q] = *reinterpret_cast<const float4*>(wt_s + (hoff + (grp * 4 + q) * 16 + i) * kLdT + 8 * g);
                    acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma4(dl[0], w[q].x, acc[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma4(dl[1], w[q].y, acc[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma4(dl[2], w[q].z, acc[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma4(dl[3], w[q].w, acc[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    w[q] = *reinterpret_cast<const float4*>(wt_s + (hoff + (grp * 4 + q) * 16 + i) * kLdT + 8 * g + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma4(dl[4], w[q].x, acc[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma4(dl[5], w[q].y, acc[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma4(dl[6], w[q].z, acc[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma4(dl[7], w[q].w, acc[q]);
                float c[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = grp * 4 + q;
                    const float xv = x_s[k * 16 + i];
                    c[q] = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = xv + yc[k][r] > 0.f ? acc[q][r] : 0.f;
                        dya[k][r] += v;
                        c[q] += v;
                    }
                }
                // column sums over the tile's 16 rows: a reduce-scatter over the four 16-lane groups (3 exchanges for
                // 4 tiles); group g ends up with tile grp*4 + g, so the store is one full 64-lane row segment
                const bool hi = g >= 2, odd = g & 1;
                const float r0 = __shfl_xor(hi ? c[0] : c[2], 32), r1 = __shfl_xor(hi ? c[1] : c[3], 32);
                const float k0 = (hi ? c[2] : c[0]) + r0, k1 = (hi ? c[3] : c[1]) + r1;
                const float r2 = __shfl_xor(odd ? k0 : k1, 16);
                px[grp * 64] = (odd ? k1 : k0) + r2;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef JF_LOAD_T
    // fold the four frame lanes' dya tiles in order (fixed order: deterministic), then one coalesced write
    for (int w = 0; w < 4; ++w) {
        if (tw == w) {
#pragma unroll
            for (int k = 0; k < NTH; ++k) {
                if (EXACT || t0n + k < nt_n) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* p = red_s + (4 * g + r) * H + hoff + k * 16 + i;
                        *p = (w == 0 ? 0.f : *p) + dya[k][r];
                    }
                }
            }
        }
        __syncthreads();
    }
    const int h4 = H >> 2;
    for (int idx = tid; idx < 16 * h4; idx += 512) {
        const int row = idx / h4, c4 = idx - row * h4;
        if (u0 + row < U1)
            *reinterpret_cast<float4*>(a.part_y + (((long)tcb * B + b) * U1 + u0 + row) * H + 4 * c4) =
                *reinterpret_cast<const float4*>(red_s + row * H + 4 * c4);
    }
}

// ---- backward, weight ----------------------------------------------------------------------------------------------
// kWeightGrid persistent blocks of 4 waves; wave w owns hidden units [w H/4, (w+1) H/4) (NQ 16-wide tiles) and both
// class tiles.  MFMA: A[row = class (lane & 15)][k = g] = dl[u0 + 4 s + g][class] at step s,
//                     B[k = g][col = hidden] = relu(xa[t][hidden] + ya[u0 + 4 s + g][hidden]).
template <int NQ, bool EXACT>
__global__ __launch_bounds__(256) void joint_fused_bwd_weight_kernel(JArgs a, int nitems, int ntw) {
    const int H = a.H, K = a.K, T = a.T, U1 = a.U1;
    const int nq = EXACT ? NQ : H >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int hbase = wave * (H >> 2);
    const bool valid0 = i < K, valid1 = 16 + i < K;
    f32x4 acc[2][NQ];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db0 = 0.f, db1 = 0.f;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        int it = item;
        const int tcb = it % ntw; it /= ntw;
        const int ut = it % a.nut;
        const int b = it / a.nut;
        const int u0 = ut * 16;
        float yb[4][NQ];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* yr = a.ya + ((long)b * U1 + min(u0 + 4 * s + g, U1 - 1)) * H + hbase + i;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (EXACT || q < nq) yb[s][q] = yr[q * 16];
        }
        const int t_beg = tcb * kWeightTc, t_end = min(T, t_beg + kWeightTc);
        // lattice offsets of this lane's four rows (frame-independent part) and their validity
        long roff[4];
        bool ok0[4], ok1[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int uu = u0 + 4 * s + g;
            const bool vu = uu < U1;
            roff[s] = ((long)b * T * U1 + (vu ? uu : 0)) * K + i;
            ok0[s] = vu && valid0;
            ok1[s] = vu && valid1;
        }
        const long tstride = (long)U1 * K;
        const float* xbase = a.xa + (long)b * T * H + hbase + i;
        // raw operands of one frame; the next frame's are in flight while this frame's 64 MFMAs run
        float g0r[4], l0r[4], g1r[4], l1r[4], xq[NQ];
#define JF_LOAD_FRAME(t_)                                                   \
    do {                                                                    \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                     \
            const long o0 = ok0[s] ? roff[s] + (long)(t_) * tstride : 0;    \
            const long o1 = ok1[s] ? roff[s] + (long)(t_) * tstride + 16 : 0; \
            g0r[s] = a.glp[o0]; l0r[s] = a.logp[o0];                        \
            g1r[s] = a.glp[o1]; l1r[s] = a.logp[o1];                        \
        }                                                                   \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q)                      \
            if (EXACT || q < nq) xq[q] = xbase[(long)(t_) * H + q * 16];             \
    } while (0)
        JF_LOAD_FRAME(t_beg);
        for (int t = t_beg; t < t_end; ++t) {
            float d0[4], d1[4], xc[NQ];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float g0 = ok0[s] ? g0r[s] : 0.f, g1 = ok1[s] ? g1r[s] : 0.f;
                const float l0 = ok0[s] ? l0r[s] : -INFINITY, l1 = ok1[s] ? l1r[s] : -INFINITY;
                const float ssum = row16_sum(g0 + g1);
                d0[s] = g0 - sa_exp2(l0 * SA_LOG2E) * ssum;
                d1[s] = g1 - sa_exp2(l1 * SA_LOG2E) * ssum;
                db0 += d0[s];
                db1 += d1[s];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) xc[q] = xq[q];
            {
                const int tn = t + 1 < t_end ? t + 1 : t;
                JF_LOAD_FRAME(tn);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (EXACT || q < nq) {
                        const float z = fmaxf(xc[q] + yb[s][q], 0.f);
                        acc[0][q] = mfma4(d0[s], z, acc[0][q]);
                        acc[1][q] = mfma4(d1[s], z, acc[1][q]);
                    }
                }
            }
        }
#undef JF_LOAD_FRAME
    }
    // C/D layout: row (class) = c * 16 + 4 g + r, col (hidden) = hbase + q * 16 + i
    float* pw = a.part_w + (long)blockIdx.x * kKP * H;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (EXACT || q < nq) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[(long)(c * 16 + 4 * g + r) * H + hbase + q * 16 + i] = acc[c][q][r];
            }
    if (wave == 0) {  // every wave saw the same lattice rows: the bias gradient is taken from wave 0
        db0 = rows4_sum(db0);
        db1 = rows4_sum(db1);
        if (g == 0) {
            a.part_b[(long)blockIdx.x * kKP + i] = db0;
            a.part_b[(long)blockIdx.x * kKP + 16 + i] = db1;
        }
    }
}

// dst[idx] = sum over p < n of src[p * inner + idx], idx < count (fixed order)
__global__ __launch_bounds__(256) void sum_leading_kernel(const float* __restrict__ src, float* __restrict__ dst, int n,
                                                          long inner, long count) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= count) return;
    float s = 0.f;
    int p = 0;
    for (; p + 4 <= n; p += 4) {
        const float v0 = src[(long)p * inner + idx], v1 = src[(long)(p + 1) * inner + idx];
        const float v2 = src[(long)(p + 2) * inner + idx], v3 = src[(long)(p + 3) * inner + idx];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; p < n; ++p) s += src[(long)p * inner + idx];
    dst[idx] = s;
}

struct JShape {
    int nut, ntc_f, ntc_d, ntw;
    size_t off_x, off_y, off_w, off_b, total;
};

bool joint_shape(int B, int T, int U1, int H, int K, JShape* s) {
    if (B <= 0 || T <= 0 || U1 <= 0 || H <= 0 || K <= 0) return false;
    if (H % 64 != 0 || H > 512 || K > kKP) return false;
    s->nut = (U1 + 15) / 16;
    s->ntc_f = (T + kFwdTc - 1) / kFwdTc;
    s->ntc_d = (T + kDataTc - 1) / kDataTc;
    s->ntw = (T + kWeightTc - 1) / kWeightTc;
    if ((long)B * s->nut * s->ntw > 0x7fffffffL || (long)B * s->nut * s->ntc_f > 0x7fffffffL) return false;
    size_t o = 0;
    s->off_x = o; o += sa_align_up((size_t)s->nut * B * T * H * sizeof(float), 256);
    s->off_y = o; o += sa_align_up((size_t)s->ntc_d * B * U1 * H * sizeof(float), 256);
    s->off_w = o; o += sa_align_up((size_t)kWeightGrid * kKP * H * sizeof(float), 256);
    s->off_b = o; o += sa_align_up((size_t)kWeightGrid * kKP * sizeof(float), 256);
    s->total = o;
    return true;
}

template <typename F>
ctcStatus_t set_lds(F fn, size_t bytes) {
    return hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
               ? CTC_STATUS_SUCCESS
               : CTC_STATUS_EXECUTION_FAILED;
}

}  // namespace

extern "C" size_t sa_joint_fused_workspace_bytes(int B, int T, int U1, int H, int K) {
    JShape s;
    return joint_shape(B, T, U1, H, K, &s) ? s.total : 0;
}

extern "C" ctcStatus_t sa_joint_fused_fwd(const float* xa, const float* ya, const float* w2, const float* b2,
                                          float* logp, int B, int T, int U1, int H, int K, void* stream) {
    SA_CLEAR_ERR();
    JShape s;
    if (!xa || !ya || !w2 || !b2 || !logp || !joint_shape(B, T, U1, H, K, &s)) return CTC_STATUS_INVALID_VALUE;
    JArgs a{};
    a.xa = xa; a.ya = ya; a.w2 = w2; a.b2 = b2; a.logp = logp;
    a.B = B; a.T = T; a.U1 = U1; a.H = H; a.K = K; a.nut = s.nut; a.ntc = s.ntc_f;
    const size_t lds = ((size_t)kKP * (H + 4) + 8 * 2 * H) * sizeof(float);
    const dim3 grid((unsigned)(s.ntc_f * s.nut * B));
    ctcStatus_t st;
#define JF_FWD(NJ_, EX_)                                                                                        \
    do {                                                                                                        \
        if ((st = set_lds(joint_fused_fwd_kernel<NJ_, EX_>, lds)) != CTC_STATUS_SUCCESS) return st;             \
        hipLaunchKernelGGL((joint_fused_fwd_kernel<NJ_, EX_>), grid, dim3(512), lds, (hipStream_t)stream, a);   \
    } while (0)
    if (H == 128) JF_FWD(8, true);
    else if (H == 256) JF_FWD(16, true);
    else if (H == 512) JF_FWD(32, true);
    else JF_FWD(32, false);
#undef JF_FWD
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_joint_fused_bwd(const float* glp, const float* logp, const float* xa, const float* ya,
                                          const float* w2, float* dxa, float* dya, float* dw2, float* db2, int B, int T,
                                          int U1, int H, int K, void* workspace, size_t workspace_bytes, void* stream) {
    SA_CLEAR_ERR();
    JShape s;
    if (!glp || !logp || !xa || !ya || !w2 || !dxa || !dya || !dw2 || !db2 || !workspace ||
        !joint_shape(B, T, U1, H, K, &s))
        return CTC_STATUS_INVALID_VALUE;
    if (workspace_bytes < s.total) return CTC_STATUS_INVALID_VALUE;
    hipStream_t st_ = (hipStream_t)stream;
    char* ws = (char*)workspace;
    JArgs a{};
    a.xa = xa; a.ya = ya; a.w2 = w2; a.logp = const_cast<float*>(logp); a.glp = glp;
    a.part_x = (float*)(ws + s.off_x); a.part_y = (float*)(ws + s.off_y);
    a.part_w = (float*)(ws + s.off_w); a.part_b = (float*)(ws + s.off_b);
    a.B = B; a.T = T; a.U1 = U1; a.H = H; a.K = K; a.nut = s.nut; a.ntc = s.ntc_d;
    ctcStatus_t st;
    const size_t lds = ((size_t)H * kLdT + 16 * H + 8 * (H == 128 ? 64 : H == 256 ? 128 : 256)) * sizeof(float);
    const dim3 grid_d((unsigned)(s.ntc_d * s.nut * B));
#define JF_DATA(NT_, EX_)                                                                                      \
    do {                                                                                                       \
        if ((st = set_lds(joint_fused_bwd_data_kernel<NT_, EX_>, lds)) != CTC_STATUS_SUCCESS) return st;       \
        hipLaunchKernelGGL((joint_fused_bwd_data_kernel<NT_, EX_>), grid_d, dim3(512), lds, st_, a);           \
    } while (0)
    if (H == 128) JF_DATA(8, true);
    else if (H == 256) JF_DATA(16, true);
    else if (H == 512) JF_DATA(32, true);
    else JF_DATA(32, false);
#undef JF_DATA
    const int nitems = B * s.nut * s.ntw;
    const int gw = nitems < kWeightGrid ? nitems : kWeightGrid;
#define JF_WEIGHT(NQ_, EX_) \
    hipLaunchKernelGGL((joint_fused_bwd_weight_kernel<NQ_, EX_>), dim3(gw), dim3(256), 0, st_, a, nitems, s.ntw)
    if (H == 128) JF_WEIGHT(2, true);
    else if (H == 256) JF_WEIGHT(4, true);
    else if (H == 512) JF_WEIGHT(8, true);
    else JF_WEIGHT(8, false);
#undef JF_WEIGHT
    const long nx = (long)B * T * H, ny = (long)B * U1 * H;
    hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st_, a.part_x, dxa, s.nut,
                       nx, nx);
    hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)((ny + 255) / 256)), dim3(256), 0, st_, a.part_y, dya,
                       s.ntc_d, ny, ny);
    hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)(((long)K * H + 255) / 256)), dim3(256), 0, st_, a.part_w, dw2,
                       gw, (long)kKP * H, (long)K * H);
    hipLaunchKernelGGL(sum_leading_kernel, dim3(1), dim3(256), 0, st_, a.part_b, db2, gw, (long)kKP, (long)K);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
