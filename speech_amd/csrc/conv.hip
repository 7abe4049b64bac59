// conv.hip -- Conv2d(in_c, out_c, (kh, kw), stride (s, s), padding 0) + ReLU of the encoder front end
// (reference: /root/reference/speech/models/model.py:19-29,61-62) and the layout change of model.py:66-71.
//
// Round-1 formulation: im2col into a workspace (coalesced reads of the padded feature tensor: a window row is kw
// contiguous floats) followed by the fp32 MFMA GEMM  y[pos, c] = cols[pos, :] . w[c, :] + bias[c]  with ReLU and the
// output scatter fused into the GEMM epilogue, so the last conv writes the GRU-ready (B, T', out_c * F')
// channel-major feature layout directly (no transpose pass).  Backward: mask + repack dy, dW = dyp^T cols (split-K
// GEMM over ~4e5 positions), db = column sums, dx = col2im gather of dyp W (only for stacked convs).
#include "common.h"
#include "internal.h"

namespace {

struct ConvGeom {
    int B, C, T, F, O, kh, kw, s, To, Fo, K;  // K = C * kh * kw
};

__host__ __device__ inline int conv_out(int n, int k, int s) { return (n - k + 1 + s - 1) / s; }

// cols[pos][k], pos = (b, t', f'), k = (c, i, j).  grid (npos / 16) blocks of 256 threads; a block copies 16 output
// positions: the (c, i) rows of a position's window are kw contiguous floats in x and in cols, so the block walks the
// K / kw rows of its 16 positions with kw-wide coalesced segments (4 floats per lane when kw % 4 == 0).  The index
// decomposition (the divisions) is done once per (position, row), not per element.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols,
                                                     ConvGeom g) {
    const long npos = (long)g.B * g.To * g.Fo;
    const int rows = g.C * g.kh;                      // (c, i) rows per position
    const int vec = (g.kw & 3) == 0 ? 4 : 1;
    const int lanes_per_row = g.kw / vec;             // threads covering one kw-wide row
    const int rows_per_pass = 256 / lanes_per_row;    // (position, row) pairs handled per pass of the block
    if (rows_per_pass == 0) return;                   // kw > 1024: not a shape this library builds
    const int lr = threadIdx.x / lanes_per_row, lj = (threadIdx.x - lr * lanes_per_row) * vec;
    if (lr >= rows_per_pass) return;
    const long pos0 = (long)blockIdx.x * 16;
    const int work = 16 * rows;
    for (int wi = lr; wi < work; wi += rows_per_pass) {
        const int p = wi / rows, r = wi - p * rows;   // position within the block's 16, row (c, i)
        const long pos = pos0 + p;
        if (pos >= npos) break;
        const int c = r / g.kh, i = r - c * g.kh;
        const int fo = (int)(pos % g.Fo);
        const long bt = pos / g.Fo;
        const int to = (int)(bt % g.To), b = (int)(bt / g.To);
        const float* src = x + (((long)b * g.C + c) * g.T + (g.s * to + i)) * g.F + g.s * fo + lj;
        float* dst = cols + pos * g.K + (long)r * g.kw + lj;
        if (vec == 4) {
            const float4 v = make_float4(src[0], src[1], src[2], src[3]);  // x rows are not 16-byte aligned in general
            *reinterpret_cast<float4*>(dst) = v;                            // cols rows are (K % 4 == 0)
        } else {
            *dst = *src;
        }
    }
}

// dyp[pos][o] = dy[b, o, t', f'] * (y > 0), reading dy / y through the caller's strides
__global__ __launch_bounds__(256) void relu_mask_pack_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             float* __restrict__ dyp, ConvGeom g, long ys_b, long ys_c,
                                                             long ys_t) {
    const long total = (long)g.B * g.To * g.Fo * g.O;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        // thread order: f' fastest (coalesced reads), then o, then (b, t')
        const int fo = (int)(idx % g.Fo);
        const int o = (int)((idx / g.Fo) % g.O);
        const long bt = idx / ((long)g.Fo * g.O);
        const int to = (int)(bt % g.To);
        const int b = (int)(bt / g.To);
        const long off = (long)b * ys_b + (long)o * ys_c + (long)to * ys_t + fo;
        const float v = y[off] > 0.f ? dy[off] : 0.f;
        dyp[((bt * g.Fo) + fo) * g.O + o] = v;
    }
}

// dx[b, c, t, f] = sum over windows covering (t, f) of dcols[(b, t', f')][(c, i, j)]   (deterministic).
// One WAVE per output row (b, c, t); lane = tap j (kw <= 64).  For each kernel row i with t' = (t - i) / s valid, the
// wave walks f' = 0 .. Fo-1 reading the kw contiguous taps of dcols[(b, t', f')][(c, i, :)] (one coalesced segment)
// into a per-lane accumulator that holds the partial sum of output column f = s * f' + lane; after each position the
// s leftmost columns are complete (no later window reaches them): they are emitted and the accumulators slide left by
// s lanes (DPP wave shifts) -- an anti-diagonal sum without any scattered access.  The row's F sums are collected in
// LDS over the kh kernel rows and written once.
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx,
                                                     ConvGeom g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* row = reinterpret_cast<float*>(smem_raw) + wave * g.F;  // [F] sums of this wave's output row
    const long nrows = (long)g.B * g.C * g.T;
    const long rid = (long)blockIdx.x * 4 + wave;
    if (rid >= nrows) return;
    const int t = (int)(rid % g.T);
    const int c = (int)((rid / g.T) % g.C);
    const int b = (int)(rid / ((long)g.T * g.C));
    for (int f = lane; f < g.F; f += 64) row[f] = 0.f;
    for (int i = t % g.s; i < g.kh; i += g.s) {
        const int to = (t - i) / g.s;
        if (t - i < 0 || to >= g.To) continue;
        const float* src = dcols + (((long)b * g.To + to) * g.Fo) * g.K + (long)(c * g.kh + i) * g.kw + lane;
        float acc = 0.f;
        for (int fo = 0; fo < g.Fo; ++fo) {
            acc += lane < g.kw ? src[(long)fo * g.K] : 0.f;
            for (int k = 0; k < g.s; ++k) {  // columns s*fo .. s*fo + s-1 are final: emit lane 0, slide left
                if (lane == 0) row[g.s * fo + k] += acc;
                acc = sa_wave_shl1(acc, 0.f);
            }
        }
        // the kw - s columns still in flight: f = s*Fo + lane, lane < kw - s
        const int f = g.s * g.Fo + lane;
        if (lane < g.kw - g.s && f < g.F) row[f] += acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float* out = dx + rid * g.F;
    for (int f = lane; f < g.F; f += 64) out[f] = row[f];
}

// The same gather for narrow kernels (kw <= 32: the stacked 32-channel convs of the shipped TIMIT / WSJ configs, kw = 8),
// where one lane per tap leaves most of a wave idle and every load a 32-byte sliver.  One BLOCK per output frame (b, t);
// a wave owns 64 / kw channels and a lane is (channel, tap): for each kernel row i with a valid t' and each f' the wave
// loads its channels' taps of dcols[(b, t', f')] and adds them into the channels' LDS rows at column s f' + tap (within
// one step the columns of a channel are distinct, steps are sequential in the wave: no atomics, a fixed order).
// Loads are issued four positions ahead.  dynamic LDS: 4 waves x (64 / kw) channels x F floats.
__global__ __launch_bounds__(256) void col2im_narrow_kernel(const float* __restrict__ dcols, float* __restrict__ dx,
                                                            ConvGeom g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cpw = 64 / g.kw, cpb = 4 * cpw;
    const int cl = lane / g.kw, j = lane - cl * g.kw;
    const bool live = cl < cpw;
    float* rows = reinterpret_cast<float*>(smem_raw) + (long)wave * cpw * g.F;  // this wave's [cpw][F]
    const int t = blockIdx.x % g.T, b = blockIdx.x / g.T;
    for (int c0 = 0; c0 < g.C; c0 += cpb) {
        const int c = c0 + wave * cpw + cl;
        const bool on = live && c < g.C;
        for (int idx = lane; idx < cpw * g.F; idx += 64) rows[idx] = 0.f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float* myrow = rows + cl * g.F + j;
        for (int i = t % g.s; i < g.kh; i += g.s) {
            const int to = (t - i) / g.s;
            if (t - i < 0 || to >= g.To) continue;
            const float* src = dcols + (((long)b * g.To + to) * g.Fo) * g.K + (long)((on ? c : 0) * g.kh + i) * g.kw + j;
            int fo = 0;
            for (; fo + 4 <= g.Fo; fo += 4) {
                const float v0 = src[(long)fo * g.K], v1 = src[(long)(fo + 1) * g.K];
                const float v2 = src[(long)(fo + 2) * g.K], v3 = src[(long)(fo + 3) * g.K];
                if (on) {
                    myrow[g.s * fo] += v0;
                    myrow[g.s * (fo + 1)] += v1;
                    myrow[g.s * (fo + 2)] += v2;
                    myrow[g.s * (fo + 3)] += v3;
                }
            }
            for (; fo < g.Fo; ++fo) {
                const float v = src[(long)fo * g.K];
                if (on) myrow[g.s * fo] += v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int idx = lane; idx < cpw * g.F; idx += 64) {
            const int cc = c0 + wave * cpw + idx / g.F;
            if (cc < g.C) dx[(((long)b * g.C + cc) * g.T + t) * g.F + idx % g.F] = rows[idx];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// dW[o][k] = sum_pos dyp[pos][o] * cols[pos][k] for small O*K (the first conv: 32 x 160): an outer-product reduction
// over ~4e5 positions.  As a GEMM it has M = O = 32 of a 128-row tile (75 % of the MFMAs wasted) and needs a 128-way
// split-K; here each block reduces a slab of positions with the O*K accumulators register-tiled over its threads
// (thread = 4 output channels x up to 8 taps: 4 + 8 LDS operand reads feed 32 FMAs), then a second kernel adds the
// per-block partials in fixed order.
constexpr int kDwTO = 4;       // output channels per thread
constexpr int kDwTK = 8;       // taps per thread (strided by the number of tap groups)
constexpr int kDwPosTile = 32; // positions staged per LDS tile

__host__ __device__ inline bool conv_dw_fits(int O, int K) {
    const int nog = (O + kDwTO - 1) / kDwTO;
    if (nog > 256 || (O % kDwTO) != 0) return false;
    const int nkg = 256 / nog;
    return nkg >= 1 && (long)nkg * kDwTK >= K;
}

__global__ __launch_bounds__(256) void conv_dw_partial_kernel(const float* __restrict__ dyp,
                                                              const float* __restrict__ cols, long npos, int O, int K,
                                                              int pos_per_block, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_dy = sm;                      // [kDwPosTile][O]
    float* s_co = sm + kDwPosTile * O;     // [kDwPosTile][K]
    const int tid = threadIdx.x;
    const int nog = O / kDwTO, nkg = 256 / nog;
    const int og = tid / nkg, kg = tid - og * nkg;
    const bool worker = og < nog;
    float acc[kDwTO][kDwTK];
#pragma unroll
    for (int i = 0; i < kDwTO; ++i)
#pragma unroll
        for (int j = 0; j < kDwTK; ++j) acc[i][j] = 0.f;
    const long p0 = (long)blockIdx.x * pos_per_block;
    const long p1 = min(npos, p0 + pos_per_block);
    for (long pt = p0; pt < p1; pt += kDwPosTile) {
        const int np = (int)min((long)kDwPosTile, p1 - pt);
        __syncthreads();
        for (int i = tid; i < np * O; i += 256) s_dy[i] = dyp[pt * O + i];
        for (int i = tid; i < np * K; i += 256) s_co[i] = cols[pt * K + i];
        __syncthreads();
        if (!worker) continue;
        for (int p = 0; p < np; ++p) {
            const float4 d = *reinterpret_cast<const float4*>(&s_dy[p * O + og * kDwTO]);
            float c[kDwTK];
#pragma unroll
            for (int j = 0; j < kDwTK; ++j) {
                const int k = kg + j * nkg;
                c[j] = k < K ? s_co[p * K + k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < kDwTK; ++j) {
                acc[0][j] = fmaf(d.x, c[j], acc[0][j]);
                acc[1][j] = fmaf(d.y, c[j], acc[1][j]);
                acc[2][j] = fmaf(d.z, c[j], acc[2][j]);
                acc[3][j] = fmaf(d.w, c[j], acc[3][j]);
            }
        }
    }
    if (!worker) return;
    float* out = part + (long)blockIdx.x * O * K;
#pragma unroll
    for (int i = 0; i < kDwTO; ++i)
#pragma unroll
        for (int j = 0; j < kDwTK; ++j) {
            const int k = kg + j * nkg;
            if (k < K) out[(og * kDwTO + i) * K + k] = acc[i][j];
        }
}

__global__ __launch_bounds__(256) void conv_dw_final_kernel(const float* __restrict__ part, int nblocks, int OK,
                                                            float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OK) return;
    float s0 = 0.f, s1 = 0.f;
    int b = 0;
    for (; b + 1 < nblocks; b += 2) { s0 += part[(long)b * OK + idx]; s1 += part[(long)(b + 1) * OK + idx]; }
    if (b < nblocks) s0 += part[(long)b * OK + idx];
    dw[idx] = s0 + s1;
}

// ------------------------------------------------------------------------------------------ direct first convolution
// The first conv of every shipped config has ONE input channel (the spectrogram) and at most 32 output channels
// ([32, 5, 32, 2]): its im2col matrix is 160 columns of re-arranged input -- 255 MB written and read back at S-LIBRI for
// 10 MB of features -- and as a GEMM it fills a quarter of a 128-wide tile.  The direct kernels below never build it:
// a block stages the input rows its output frames touch in LDS (a slab of (TT - 1) s + kh rows x F floats) and feeds
// v_mfma_f32_32x32x2_f32 with operands gathered from that slab.
//   forward : D[channel][position] += W[channel][tap] * x[window(position)][tap]      (M = 32 channels, N = positions)
//             rows = channels so that a register of the accumulator tile holds 32 CONSECUTIVE positions of one channel:
//             the epilogue (bias, ReLU, the caller's layout) writes contiguous segments.
//   backward: dW[channel][tap] += dyp[position][channel] * x[window(position)][tap]   (reduction over positions),
//             tap tiles of 32 plus one column of ones (-> the bias gradient); a block walks several slabs with its
//             partial sums in registers, the per-block partials are folded in a fixed order (conv_dw_final_kernel).
typedef float f32x16c __attribute__((ext_vector_type(16)));
constexpr int kDirMaxTT = 16;  // output frames per slab, at most

// frames per slab: the most (<= 16) whose LDS footprint stays <= 64 KB in both kernels (2+ blocks per CU)
__host__ __device__ inline int conv_direct_tt(int kh, int kw, int s, int F, int Fo) {
    for (int tt = kDirMaxTT; tt >= 1; tt >>= 1) {
        const long rows = (long)(tt - 1) * s + kh;
        const long fwd = rows * F + (long)kh * kw * 32, bwd = rows * F + (long)tt * Fo * 33;
        if ((fwd > bwd ? fwd : bwd) * 4 <= 64 * 1024) return tt;
    }
    return 0;
}
__host__ __device__ inline bool conv_direct_ok(int C, int O, int kh, int kw, int s, int F, int Fo) {
    const int K = kh * kw;
    return C == 1 && O <= 32 && (K & 1) == 0 && K <= 224 && conv_direct_tt(kh, kw, s, F, Fo) > 0;
}

struct DirGeom {
    int B, T, F, O, kh, kw, s, To, Fo, K, TT;
    long ys_b, ys_c, ys_t;
};

// Stage the slab of input rows [s * t0, s * t0 + rows) of utterance b (zero beyond T) into xs[rows][F].
__device__ __forceinline__ void dir_stage_x(const float* __restrict__ x, float* __restrict__ xs, const DirGeom& g, int b,
                                            int t0, int rows) {
    const float* src = x + ((long)b * g.T + (long)g.s * t0) * g.F;
    const int n = rows * g.F;
    const int valid = max(0, min(rows, g.T - g.s * t0)) * g.F;
    for (int i = threadIdx.x; i < n; i += 256) xs[i] = i < valid ? src[i] : 0.f;
}

// grid (ceil(To / g.TT), B).  LDS: xs[rows][F] | ws[K][32] (tap-major weights, channels >= O zero).
__global__ __launch_bounds__(256) void conv1_fwd_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              DirGeom g) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int rows = (g.TT - 1) * g.s + g.kh;
    float* xs = dsm;
    float* ws = dsm + rows * g.F;
    const int b = blockIdx.y, t0 = blockIdx.x * g.TT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dir_stage_x(x, xs, g, b, t0, rows);
    for (int i = tid; i < g.K * 32; i += 256) {
        const int k = i >> 5, c = i & 31;
        ws[i] = c < g.O ? w[(long)c * g.K + k] : 0.f;
    }
    __syncthreads();
    const int nt = min(g.TT, g.To - t0);
    const int npos = nt * g.Fo;
    const int h = lane >> 5, r = lane & 31;
    for (int tile = wave; tile * 32 < npos; tile += 4) {
        const int p = tile * 32 + r;
        const bool pv = p < npos;
        const int tl = pv ? p / g.Fo : 0, fo = pv ? p - tl * g.Fo : 0;
        const float* xb = xs + (g.s * tl) * g.F + g.s * fo;       // + i * F + j for tap k = i * kw + j (k = 2 ks + h)
        const float* wb = ws + h * 32 + r;                          // + 64 ks
        f32x16c acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        int j0 = 0, roff = 0;  // tap 2 ks = (roff / F, j0): kw is not assumed even, so a pair may straddle a row
        for (int ks0 = 0; ks0 < g.K / 2; ks0 += 4) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool on = ks0 + u < g.K / 2;
                int jj = j0 + h, ro = roff;
                if (jj >= g.kw) { jj -= g.kw; ro += g.F; }
                av[u] = on ? wb[64 * (ks0 + u)] : 0.f;
                bv[u] = on ? xb[ro + jj] : 0.f;
                j0 += 2;
                if (j0 >= g.kw) { j0 -= g.kw; roff += g.F; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        // C/D layout: column (position) = lane & 31, row (channel) = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
        if (pv) {
            float* yo = y + (long)b * g.ys_b + (long)(t0 + tl) * g.ys_t + fo;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int c = (q & 3) + 8 * (q >> 2) + 4 * h;
                if (c < g.O) yo[(long)c * g.ys_c] = fmaxf(acc[q] + bias[c], 0.f);
            }
        }
    }
}

// Weight / bias gradient of the same conv.  grid = nblocks persistent blocks walking the (b, slab) list; LDS:
// xs[rows][F] | dyp[g.TT * Fo][33] (dy masked by y > 0; pitch 33: the A fragment reads 32 channels of one position).
// Each wave takes every 4th PAIR of positions of a slab (the MFMA's k = 2 lanes) and keeps NT tap tiles of 16 registers;
// tile n covers taps 32 n .. 32 n + 31, column K of the last tile multiplies by 1 (the bias gradient).
template <int NT>
__global__ __launch_bounds__(256) void conv1_dw_direct_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy, float* __restrict__ part,
                                                             DirGeom g, int nslab_t) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int rows = (g.TT - 1) * g.s + g.kh;
    float* xs = dsm;
    float* dyp = dsm + rows * g.F;                 // [g.TT * Fo][33]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, r = lane & 31;
    f32x16c acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
    // this lane's tap per tile: k = 32 n + r -> offset i * F + j in the slab; tap == K: the constant 1; beyond: 0
    int toff[NT];
    float tone[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int k = 32 * n + r;
        toff[n] = k < g.K ? (k / g.kw) * g.F + (k % g.kw) : -1;
        tone[n] = k == g.K ? 1.f : 0.f;
    }
    const int nslabs = g.B * nslab_t;
    for (int slab = blockIdx.x; slab < nslabs; slab += gridDim.x) {
        const int b = slab / nslab_t, t0 = (slab - b * nslab_t) * g.TT;
        const int nt = min(g.TT, g.To - t0);
        const int npos = nt * g.Fo;
        __syncthreads();  // the previous slab's readers are done
        dir_stage_x(x, xs, g, b, t0, rows);
        for (int i = tid; i < npos * 32; i += 256) {  // (channel, frame, f') with f' fastest: the caller's layout is f'-contiguous
            const int fo = i % g.Fo, q = i / g.Fo;
            const int tl = q % nt, c = q / nt;
            float v = 0.f;
            if (c < g.O) {
                const long o = (long)b * g.ys_b + (long)c * g.ys_c + (long)(t0 + tl) * g.ys_t + fo;
                v = y[o] > 0.f ? dy[o] : 0.f;
            }
            dyp[(tl * g.Fo + fo) * 33 + c] = v;
        }
        __syncthreads();
        for (int p0 = 2 * wave; p0 < npos; p0 += 16) {  // two position pairs per trip: their LDS reads go out together
            float a[2], bv[2][NT];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = p0 + 8 * e + h;
                const bool pv = p < npos;
                const int tl = pv ? p / g.Fo : 0, fo = pv ? p - tl * g.Fo : 0;
                a[e] = pv ? dyp[p * 33 + r] : 0.f;
                const float* xb = xs + (g.s * tl) * g.F + g.s * fo;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float v = toff[n] >= 0 ? xb[toff[n]] : tone[n];
                    bv[e][n] = pv ? v : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bv[e][n], acc[n], 0, 0, 0);
        }
    }
    // fold the 4 waves in wave order (fixed: deterministic) through one [32][32 NT] LDS tile, write the block's partial
    float* red = dsm;  // 32 * 32 NT floats <= the LDS request (checked by the host)
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = (q & 3) + 8 * (q >> 2) + 4 * h;
                    float* o = &red[c * (32 * NT) + 32 * n + r];
                    *o = wv == 0 ? acc[n][q] : *o + acc[n][q];
                }
        }
    }
    __syncthreads();
    float* out = part + (long)blockIdx.x * 32 * (32 * NT);
    for (int i = tid; i < 32 * 32 * NT; i += 256) out[i] = red[i];
}

// dw[c][k] (c < O, k < K) and dbias[c] = the fixed-order sum of the per-block partials [nb][32][NTK]
// 8 lanes per output: lane j adds the partials j, j + 8, ... in order, then a fixed xor tree folds the 8 sums
__global__ __launch_bounds__(256) void conv1_dw_fold_kernel(const float* __restrict__ part, int nb, int NTK, int O, int K,
                                                           float* __restrict__ dw, float* __restrict__ dbias) {
    const int idx = (blockIdx.x * 256 + threadIdx.x) >> 3, j = threadIdx.x & 7;
    const bool on = idx < O * (K + 1);
    const int c = on ? idx / (K + 1) : 0, k = on ? idx - c * (K + 1) : 0;
    const float* p = part + (long)c * NTK + k;
    float s0 = 0.f;
    if (on)
        for (int b = j; b < nb; b += 8) s0 += p[(long)b * 32 * NTK];
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) s0 += __shfl_xor(s0, o, 64);
    if (!on || j != 0) return;
    if (k < K) dw[(long)c * K + k] = s0;
    else dbias[c] = s0;
}

constexpr int kDirDwBlocks = 512;

constexpr int kDwBlocks = 512;

bool make_geom(ConvGeom* g, int B, int C, int T, int F, int O, int kh, int kw, int s) {
    if (B <= 0 || C <= 0 || T <= 0 || F <= 0 || O <= 0 || kh <= 0 || kw <= 0 || s <= 0) return false;
    g->B = B; g->C = C; g->T = T; g->F = F; g->O = O; g->kh = kh; g->kw = kw; g->s = s;
    g->To = conv_out(T, kh, s); g->Fo = conv_out(F, kw, s); g->K = C * kh * kw;
    return g->To > 0 && g->Fo > 0;
}

int grid_for(long total) {
    long gsz = (total + 255) / 256;
    return (int)(gsz > 8192 ? 8192 : (gsz < 1 ? 1 : gsz));
}

}  // namespace

static DirGeom dir_geom(const ConvGeom& g, long ys_b, long ys_c, long ys_t) {
    DirGeom d;
    d.B = g.B; d.T = g.T; d.F = g.F; d.O = g.O; d.kh = g.kh; d.kw = g.kw; d.s = g.s; d.To = g.To; d.Fo = g.Fo; d.K = g.K;
    d.ys_b = ys_b; d.ys_c = ys_c; d.ys_t = ys_t;
    d.TT = conv_direct_tt(g.kh, g.kw, g.s, g.F, g.Fo);
    return d;
}
static int dir_nt(int K) { const int need = (K + 1 + 31) / 32; return need <= 2 ? 2 : (need <= 3 ? 3 : (need <= 4 ? 4 : (need <= 6 ? 6 : 8))); }
static size_t dir_dw_lds(const ConvGeom& g) {
    const int TT = conv_direct_tt(g.kh, g.kw, g.s, g.F, g.Fo);
    const size_t stage = ((size_t)((TT - 1) * g.s + g.kh) * g.F + (size_t)TT * g.Fo * 33) * sizeof(float);
    const size_t fold = (size_t)32 * 32 * dir_nt(g.K) * sizeof(float);
    return stage > fold ? stage : fold;
}

extern "C" int sa_conv2d_is_direct(int in_c, int F, int out_c, int kh, int kw, int s) {
    const int Fo = conv_out(F, kw, s);
    return Fo > 0 && conv_direct_ok(in_c, out_c, kh, kw, s, F, Fo) ? 1 : 0;
}

extern "C" size_t sa_conv2d_fwd_workspace_bytes(int B, int in_c, int T, int F, int out_c, int kh, int kw, int s) {
    ConvGeom g;
    if (!make_geom(&g, B, in_c, T, F, out_c, kh, kw, s)) return 0;
    if (conv_direct_ok(g.C, g.O, g.kh, g.kw, g.s, g.F, g.Fo)) return 256;  // the direct kernel needs none
    const long npos = (long)g.B * g.To * g.Fo;
    return sa_align_up((size_t)npos * g.K * sizeof(float), 256) +
           sa_align_up(sa_gemm_workspace_bytes((int)npos, g.O, g.K), 256);
}

extern "C" ctcStatus_t sa_conv2d_relu_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                                          int in_c, int T, int F, int out_c, int kh, int kw, int s, long ys_b,
                                          long ys_c, long ys_t, float* keep_cols, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    ConvGeom g;
    if (!x || !w || !bias || !y || !workspace || !make_geom(&g, B, in_c, T, F, out_c, kh, kw, s))
        return CTC_STATUS_INVALID_VALUE;
    if (workspace_bytes < sa_conv2d_fwd_workspace_bytes(B, in_c, T, F, out_c, kh, kw, s)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    if (conv_direct_ok(g.C, g.O, g.kh, g.kw, g.s, g.F, g.Fo)) {  // no im2col matrix at all (keep_cols is not filled)
        const int TT = conv_direct_tt(g.kh, g.kw, g.s, g.F, g.Fo);
        const size_t lds = ((size_t)((TT - 1) * g.s + g.kh) * g.F + (size_t)g.K * 32) * sizeof(float);
        if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)conv1_fwd_direct_kernel,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return CTC_STATUS_EXECUTION_FAILED;
        hipLaunchKernelGGL(conv1_fwd_direct_kernel, dim3((g.To + TT - 1) / TT, g.B), dim3(256), lds, stream, x, w,
                           bias, y, dir_geom(g, ys_b, ys_c, ys_t));
        SA_CHECK_LAUNCH();
        return CTC_STATUS_SUCCESS;
    }
    const long npos = (long)g.B * g.To * g.Fo;
    float* cols = keep_cols ? keep_cols : (float*)workspace;  // keep_cols: caller-owned (npos x K) buffer reused by bwd
    char* gws = (char*)workspace + sa_align_up((size_t)npos * g.K * sizeof(float), 256);
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((npos + 15) / 16)), dim3(256), 0, stream, x, cols, g);
    SA_CHECK_LAUNCH();
    SaGemmEpilogue ep;
    ep.m_inner = g.Fo; ep.m_mid = g.To; ep.s_outer = ys_b; ep.s_mid = ys_t; ep.col_stride = ys_c; ep.relu = 1;
    return sa_gemm_f32_impl(0, 1, (int)npos, g.O, g.K, 1.0f, cols, g.K, w, g.K, 0.f, y, 0, bias, &ep, gws,
                            workspace_bytes - (size_t)(gws - (char*)workspace), stream);
}

extern "C" size_t sa_conv2d_bwd_workspace_bytes(int B, int in_c, int T, int F, int out_c, int kh, int kw, int s) {
    ConvGeom g;
    if (!make_geom(&g, B, in_c, T, F, out_c, kh, kw, s)) return 0;
    const long npos = (long)g.B * g.To * g.Fo;
    size_t gw = sa_gemm_workspace_bytes(g.O, g.K, (int)npos);
    const size_t gw2 = sa_gemm_workspace_bytes((int)npos, g.K, g.O);
    if (gw2 > gw) gw = gw2;
    const size_t gw3 = sa_colsum_workspace_bytes((int)npos, g.O);
    if (gw3 > gw) gw = gw3;
    const size_t gw4 = (size_t)kDwBlocks * g.O * g.K * sizeof(float);  // conv_dw_partial_kernel
    if (conv_dw_fits(g.O, g.K) && gw4 > gw) gw = gw4;
    const size_t gw5 = (size_t)kDirDwBlocks * 32 * 32 * 8 * sizeof(float);  // conv1_dw_direct_kernel partials
    if (conv_direct_ok(g.C, g.O, g.kh, g.kw, g.s, g.F, g.Fo) && gw5 > gw) gw = gw5;
    return sa_align_up((size_t)npos * g.K * sizeof(float), 256) +   // cols, reused for dcols
           sa_align_up((size_t)npos * g.O * sizeof(float), 256) +   // dyp
           sa_align_up(gw, 256);
}

extern "C" ctcStatus_t sa_conv2d_relu_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx,
                                          float* dw, float* dbias, int B, int in_c, int T, int F, int out_c, int kh,
                                          int kw, int s, long ys_b, long ys_c, long ys_t, const float* fwd_cols,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    ConvGeom g;
    if (!x || !w || !y || !dy || !dw || !dbias || !workspace || !make_geom(&g, B, in_c, T, F, out_c, kh, kw, s))
        return CTC_STATUS_INVALID_VALUE;
    if (workspace_bytes < sa_conv2d_bwd_workspace_bytes(B, in_c, T, F, out_c, kh, kw, s)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    const long npos = (long)g.B * g.To * g.Fo;
    float* cols = (float*)workspace;
    float* dyp = (float*)((char*)cols + sa_align_up((size_t)npos * g.K * sizeof(float), 256));
    char* gws = (char*)dyp + sa_align_up((size_t)npos * g.O * sizeof(float), 256);
    const size_t gws_bytes = workspace_bytes - (size_t)(gws - (char*)workspace);
    if (!dx && conv_direct_ok(g.C, g.O, g.kh, g.kw, g.s, g.F, g.Fo)) {  // direct weight / bias gradient
        const int NT = dir_nt(g.K);
        const size_t lds = dir_dw_lds(g);
        const int TT = conv_direct_tt(g.kh, g.kw, g.s, g.F, g.Fo);
        const int nslab_t = (g.To + TT - 1) / TT;
        int nb = g.B * nslab_t;
        if (nb > kDirDwBlocks) nb = kDirDwBlocks;
        if (gws_bytes < (size_t)nb * 32 * 32 * NT * sizeof(float)) return CTC_STATUS_INVALID_VALUE;
        void (*fn)(const float*, const float*, const float*, float*, DirGeom, int) =
            NT == 2 ? conv1_dw_direct_kernel<2> : NT == 3 ? conv1_dw_direct_kernel<3> : NT == 4 ? conv1_dw_direct_kernel<4>
            : NT == 6 ? conv1_dw_direct_kernel<6> : conv1_dw_direct_kernel<8>;
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return CTC_STATUS_EXECUTION_FAILED;
        hipLaunchKernelGGL(fn, dim3(nb), dim3(256), lds, stream, x, y, dy, (float*)gws, dir_geom(g, ys_b, ys_c, ys_t),
                           nslab_t);
        hipLaunchKernelGGL(conv1_dw_fold_kernel, dim3((g.O * (g.K + 1) * 8 + 255) / 256), dim3(256), 0, stream,
                           (const float*)gws, nb, 32 * NT, g.O, g.K, dw, dbias);
        SA_CHECK_LAUNCH();
        return CTC_STATUS_SUCCESS;
    }
    if (fwd_cols && !dx) {
        cols = const_cast<float*>(fwd_cols);  // the forward pass's im2col matrix, kept by the caller (read only here)
    } else {
        hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((npos + 15) / 16)), dim3(256), 0, stream, x, cols, g);
    }
    hipLaunchKernelGGL(relu_mask_pack_kernel, dim3(grid_for(npos * g.O)), dim3(256), 0, stream, dy, y, dyp, g, ys_b,
                       ys_c, ys_t);
    SA_CHECK_LAUNCH();
    // dW[o, k] = sum_pos dyp[pos, o] * cols[pos, k]
    ctcStatus_t st;
    const int OK = g.O * g.K;
    if (conv_dw_fits(g.O, g.K) && gws_bytes >= (size_t)kDwBlocks * OK * sizeof(float)) {
        int nb = (int)min((long)kDwBlocks, (npos + kDwPosTile - 1) / kDwPosTile);
        int ppb = (int)((npos + nb - 1) / nb);
        ppb = (ppb + kDwPosTile - 1) / kDwPosTile * kDwPosTile;
        nb = (int)((npos + ppb - 1) / ppb);
        hipLaunchKernelGGL(conv_dw_partial_kernel, dim3(nb), dim3(256), (size_t)kDwPosTile * (g.O + g.K) * sizeof(float),
                           stream, dyp, cols, npos, g.O, g.K, ppb, (float*)gws);
        hipLaunchKernelGGL(conv_dw_final_kernel, dim3((OK + 255) / 256), dim3(256), 0, stream, (const float*)gws, nb,
                           OK, dw);
        SA_CHECK_LAUNCH();
    } else {
        st = sa_gemm_f32_impl(1, 0, g.O, g.K, (int)npos, 1.0f, dyp, g.O, cols, g.K, 0.f, dw, g.K, nullptr, nullptr,
                              gws, gws_bytes, stream);
        if (st != CTC_STATUS_SUCCESS) return st;
    }
    st = sa_colsum_f32(dyp, g.O, (int)npos, g.O, dbias, 0, gws, gws_bytes, stream_);
    if (st != CTC_STATUS_SUCCESS) return st;
    if (dx) {
        // dcols[pos, k] = sum_o dyp[pos, o] * w[o, k]   (overwrites cols), then gather into dx
        st = sa_gemm_f32_impl(0, 0, (int)npos, g.K, g.O, 1.0f, dyp, g.O, w, g.K, 0.f, cols, g.K, nullptr, nullptr,
                              gws, gws_bytes, stream);
        if (st != CTC_STATUS_SUCCESS) return st;
        if (g.kw > 64) return CTC_STATUS_INVALID_VALUE;  // one lane per tap
        if (g.kw <= 32 && (size_t)4 * (64 / g.kw) * g.F * sizeof(float) <= 64 * 1024 && (long)g.B * g.T < 0x7fffffffL) {
            hipLaunchKernelGGL(col2im_narrow_kernel, dim3((unsigned)(g.B * g.T)), dim3(256),
                               (size_t)4 * (64 / g.kw) * g.F * sizeof(float), stream, cols, dx, g);
        } else {
            hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)(((long)g.B * g.C * g.T + 3) / 4)), dim3(256),
                               (size_t)4 * g.F * sizeof(float), stream, cols, dx, g);
        }
        SA_CHECK_LAUNCH();
    }
    return CTC_STATUS_SUCCESS;
}
