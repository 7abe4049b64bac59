// gemm_f32.hip -- fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate),
// the GEMM-shaped part of the encoder: nn.GRU's input projections and nn.Linear (reference:
// /root/reference/speech/models/model.py:35-39,126-133) and their weight / input gradients.
//
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] (+ bias[N]) (+ beta * C)
//
// Tiling (CDNA4): 128x128 block tile, BK = 32, 256 threads = 4 waves as 2x2, each wave a 64x64 sub-tile =
// 2x2 MFMA 32x32 tiles (4 x 16 accumulator VGPRs).  Operand tiles are staged in LDS k-major ([k][m], [k][n]):
// an MFMA fragment read is then 32 consecutive floats per half-wave (conflict-free ds_read_b32).  An operand that
// is k-contiguous in memory (A of an NT product, nn.Linear weights) is transposed on the way into LDS; an operand
// that is m/n-contiguous is copied with 16-byte accesses.  Global loads for tile i+1 are issued before the MFMAs of
// tile i (register staging, double-buffered LDS, one barrier per K tile).
// Tall-K products with few output tiles (dW = dA^T X, K = B*T' ~ 16k) are split along K across blockIdx.z into a
// workspace and reduced by a second kernel in a fixed order (deterministic, unlike atomics).
#include "common.h"
#include "internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = 132;  // LDS row pitch in floats (128 + 4): keeps 16-byte alignment, spreads transposed writes

constexpr int kMaxGroup = 8;

struct GemmArgs {
    // a group of up to 8 problems of identical shape shares one launch (blockIdx.z = problem * splits + split):
    // the per-chunk input projections of all layers of the GRU wavefront
    int nprob, splits;
    const float* Ag[kMaxGroup];
    const float* Bg[kMaxGroup];
    float* Cg[kMaxGroup];
    const float* biasg[kMaxGroup];
    long lda, ldb, ldc;
    int M, N, K;
    float alpha, beta;
    int k_per_split;  // multiple of BK
    int vecA, vecB;   // 16-byte loads are legal for this operand
    // optional output remap (conv: row m = (b, t', f'), column n = channel):
    //   addr(m, n) = (m / (m_inner * m_mid)) * s_outer + ((m / m_inner) % m_mid) * s_mid + (m % m_inner)
    //                + n * col_stride                                           (m_inner == 0: m * ldc + n)
    int m_inner, m_mid;
    long s_outer, s_mid, col_stride;
    int relu;
    float* partial;   // split-K workspace [nprob][splits][M][N] or null
    // optional fused column sums of A (TA products only: A is stored (K, M), so these are the bias gradients that
    // belong to a weight gradient dW = dA^T X): colsum[m] = (beta != 0 ? colsum[m] : 0) + sum_k A[k][m], computed by
    // the blocks of the first column tile from the A tiles they stage anyway.  Under split-K the per-split sums go to
    // cs_partial [nprob][splits][M] and the reduce kernel folds them with the products.
    float* colsumg[kMaxGroup];
    float* cs_partial;
};

__device__ __forceinline__ long remap_row(const GemmArgs& g, int row) {
    const int q = row / g.m_inner;
    return (long)(q / g.m_mid) * g.s_outer + (long)(q % g.m_mid) * g.s_mid + (row % g.m_inner);
}

// Load this thread's slice of one operand tile (128 x BK) into registers: NL = 4 float4 per thread.
// KCONTIG: memory is [rows][k] (k contiguous)  -> row = tid/8 + 32p, k = 4*(tid%8)      (128 B per row: full lines)
// else:    memory is [k][cols] (cols contiguous) -> k = tid/32 + 8p, col = 4*(tid%32)
// FAST: the block's tile is interior, K-tiles are full and 16-byte loads are legal: no guards, no branches.
constexpr int NL = 4;
template <bool KCONTIG, bool FAST>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int r0, int R, int k0, int kend,
                                          int vec, int tid, float4 (&v)[NL]) {
#pragma unroll
    for (int p = 0; p < NL; ++p) {
        if (FAST) {
            if (KCONTIG)
                v[p] = *reinterpret_cast<const float4*>(P + (long)(r0 + (tid >> 3) + 32 * p) * ld + k0 + 4 * (tid & 7));
            else
                v[p] = *reinterpret_cast<const float4*>(P + (long)(k0 + (tid >> 5) + 8 * p) * ld + r0 + 4 * (tid & 31));
            continue;
        }
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONTIG) {
            const int row = r0 + (tid >> 3) + 32 * p;
            const int k = k0 + 4 * (tid & 7);
            if (row < R) {
                const float* q = P + (long)row * ld + k;
                if (vec && k + 3 < kend) {
                    x = *reinterpret_cast<const float4*>(q);
                } else {
                    if (k + 0 < kend) x.x = q[0];
                    if (k + 1 < kend) x.y = q[1];
                    if (k + 2 < kend) x.z = q[2];
                    if (k + 3 < kend) x.w = q[3];
                }
            }
        } else {
            const int k = k0 + (tid >> 5) + 8 * p;
            const int col = r0 + 4 * (tid & 31);
            if (k < kend) {
                const float* q = P + (long)k * ld + col;
                if (vec && col + 3 < R) {
                    x = *reinterpret_cast<const float4*>(q);
                } else {
                    if (col + 0 < R) x.x = q[0];
                    if (col + 1 < R) x.y = q[1];
                    if (col + 2 < R) x.z = q[2];
                    if (col + 3 < R) x.w = q[3];
                }
            }
        }
        v[p] = x;
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_tile(float* __restrict__ S /* [BK][LDT] */, int tid, const float4 (&v)[NL]) {
#pragma unroll
    for (int p = 0; p < NL; ++p) {
        if (KCONTIG) {
            const int row = (tid >> 3) + 32 * p;
            const int k = 4 * (tid & 7);
            S[(k + 0) * LDT + row] = v[p].x;
            S[(k + 1) * LDT + row] = v[p].y;
            S[(k + 2) * LDT + row] = v[p].z;
            S[(k + 3) * LDT + row] = v[p].w;
        } else {
            const int k = (tid >> 5) + 8 * p;
            const int col = 4 * (tid & 31);
            *reinterpret_cast<float4*>(&S[k * LDT + col]) = v[p];
        }
    }
}

// 32 MFMAs of one K tile.  Fragments of k-step ks+1 are read from LDS before the MFMAs of k-step ks are issued.
__device__ __forceinline__ void mma_tile(const float* __restrict__ a_s, const float* __restrict__ b_s,
                                         f32x16 (&acc)[2][2]) {
    float a0 = a_s[0], a1 = a_s[32], b0 = b_s[0], b1 = b_s[32];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
        float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
        if (ks + 1 < BK / 2) {
            na0 = a_s[(2 * ks + 2) * LDT]; na1 = a_s[(2 * ks + 2) * LDT + 32];
            nb0 = b_s[(2 * ks + 2) * LDT]; nb1 = b_s[(2 * ks + 2) * LDT + 32];
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
}

// The K loop: register-staged, double-buffered LDS, one barrier per K tile.  The next tile's global loads are issued
// before this tile's 64 MFMAs per wave (4096 matrix-pipe cycles ~ 1.8 us: a block that sits alone on its CU -- the
// small per-chunk GEMMs of the GRU wavefront -- still covers an L2/HBM round trip with them).
template <bool TA, bool TB>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& g, const float* __restrict__ gA,
                                              const float* __restrict__ gB, float* smem, int m0, int n0, int kbeg,
                                              int kend, int tid, bool fast, f32x16 (&acc)[2][2], bool do_colsum,
                                              float4& csum) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    auto As = [&](int i) { return smem + i * (BK * LDT); };
    auto Bs = [&](int i) { return smem + (2 + i) * (BK * LDT); };
    const int frag = (lane >> 5) * LDT + (lane & 31);
    float4 ra[NL], rb[NL];
    auto load = [&](int tile) {
        const int k0 = kbeg + tile * BK;
        if (fast) {  // block-uniform: only the loads differ
            load_tile<!TA, true>(gA, g.lda, m0, g.M, k0, kend, g.vecA, tid, ra);
            load_tile<TB, true>(gB, g.ldb, n0, g.N, k0, kend, g.vecB, tid, rb);
        } else {
            load_tile<!TA, false>(gA, g.lda, m0, g.M, k0, kend, g.vecA, tid, ra);
            load_tile<TB, false>(gB, g.ldb, n0, g.N, k0, kend, g.vecB, tid, rb);
        }
    };
    // TA: a thread's NL float4 of an A tile are 4 consecutive m at NL different k -> its share of the column sums
    auto colsum_acc = [&]() {
        if (TA && do_colsum) {
#pragma unroll
            for (int p = 0; p < NL; ++p) { csum.x += ra[p].x; csum.y += ra[p].y; csum.z += ra[p].z; csum.w += ra[p].w; }
        }
    };
    if (ntiles > 0) {
        load(0);
        colsum_acc();
        store_tile<!TA>(As(0), tid, ra);
        store_tile<TB>(Bs(0), tid, rb);
    }
    __syncthreads();
    for (int it = 0; it < ntiles; ++it) {
        const int cur = it & 1;
        const bool more = it + 1 < ntiles;
        if (more) load(it + 1);
        mma_tile(As(cur) + frag + wm * 64, Bs(cur) + frag + wn * 64, acc);
        if (more) {
            colsum_acc();
            store_tile<!TA>(As(cur ^ 1), tid, ra);
            store_tile<TB>(Bs(cur ^ 1), tid, rb);
        }
        __syncthreads();
    }
}

// TA: A is stored (K, M) (m-contiguous);  !TA: A is stored (M, K) (k-contiguous)
// TB: B is stored (N, K) (k-contiguous);  !TB: B is stored (K, N) (n-contiguous)
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * LDT];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int prob = blockIdx.z / g.splits, split = blockIdx.z - prob * g.splits;
    const float* __restrict__ gA = g.Ag[prob];
    const float* __restrict__ gB = g.Bg[prob];
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool fast = g.vecA && g.vecB && m0 + BM <= g.M && n0 + BN <= g.N && ((kend - kbeg) % BK) == 0;
    const bool do_colsum = TA && g.colsumg[prob] != nullptr && blockIdx.x == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    gemm_mainloop<TA, TB>(g, gA, gB, smem, m0, n0, kbeg, kend, tid, fast, acc, do_colsum, csum);
    if (do_colsum) {  // fold the 8 k-rows of threads (tid >> 5) that share 4 columns; fixed order: deterministic
        float* cs = smem;  // the mainloop's last barrier has released the tiles
        *reinterpret_cast<float4*>(&cs[(tid >> 5) * BM + 4 * (tid & 31)]) = csum;
        __syncthreads();
        if (tid < BM && m0 + tid < g.M) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += cs[r * BM + tid];
            if (g.partial) {
                g.cs_partial[(long)blockIdx.z * g.M + m0 + tid] = t;
            } else {
                float* o = g.colsumg[prob] + m0 + tid;
                *o = g.beta != 0.f ? g.beta * *o + t : t;
            }
        }
    }

    // epilogue.  32x32 C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool splitk = g.partial != nullptr;
    float* __restrict__ gC = g.Cg[prob];
    const float* __restrict__ gbias = g.biasg[prob];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            float bv = 0.f;
            if (!splitk && gbias) bv = gbias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                if (splitk) {
                    g.partial[((long)blockIdx.z * g.M + row) * g.N + col] = acc[i][j][r];
                } else {
                    float* c = g.m_inner > 0 ? gC + remap_row(g, row) + col * g.col_stride
                                             : gC + (long)row * g.ldc + col;
                    float v = g.alpha * acc[i][j][r] + bv;
                    if (g.beta != 0.f) v += g.beta * *c;
                    if (g.relu) v = fmaxf(v, 0.f);
                    *c = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g, int splits) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)g.M * g.N;
    const int prob = blockIdx.y;
    if (idx >= total) {  // the tail threads fold the fused column sums
        const long m = idx - total;
        if (m < g.M && g.colsumg[prob]) {
            float t = 0.f;
            for (int z = 0; z < splits; ++z) t += g.cs_partial[((long)prob * splits + z) * g.M + m];
            float* o = g.colsumg[prob] + m;
            *o = g.beta != 0.f ? g.beta * *o + t : t;
        }
        return;
    }
    const int row = (int)(idx / g.N), col = (int)(idx % g.N);
    float s = 0.f;
    for (int z = 0; z < splits; ++z)
        s += g.partial[((long)prob * splits + z) * total + idx];  // fixed order: deterministic
    float* gC = g.Cg[prob];
    const float* gbias = g.biasg[prob];
    float* c = g.m_inner > 0 ? gC + remap_row(g, row) + col * g.col_stride : gC + (long)row * g.ldc + col;
    float v = g.alpha * s + (gbias ? gbias[col] : 0.f);
    if (g.beta != 0.f) v += g.beta * *c;
    if (g.relu) v = fmaxf(v, 0.f);
    *c = v;
}

int choose_splits(int M, int N, int K, int nprob = 1) {
    // Blocks of this kernel sit 2-3 per CU and share each SIMD's matrix pipe, so the launch is balanced when the block
    // count is just under a multiple of the CU count: pick the split (>= 4 K-tiles each) whose tiles * split fills
    // 256 / 512 / 768 slots best, preferring fewer splits on ties (less partial-sum traffic).
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * nprob;
    if (tiles >= 192 || K < 8 * BK) return 1;
    int max_s = K / (4 * BK);
    if (max_s > 128) max_s = 128;
    if (max_s < 1) return 1;
    int best = 1;
    double best_score = 0.0;
    for (int s = 1; s <= max_s; ++s) {
        const long blocks = tiles * s;
        if (blocks > 768) break;
        const long slots = ((blocks + 255) / 256) * 256;
        double score = (double)blocks / (double)slots;          // fill of the last round
        score *= blocks >= 512 ? 1.0 : (blocks >= 256 ? 0.92 : 0.5 * blocks / 256.0 + 0.3);  // want >= 2 blocks per CU
        if (score > best_score + 0.02) { best_score = score; best = s; }
    }
    return best;
}

}  // namespace

ctcStatus_t sa_gemm_f32_group_impl(int nprob, int trans_a, int trans_b, int M, int N, int K, float alpha,
                                   const float* const* A, long lda, const float* const* B, long ldb, float beta,
                                   float* const* C, long ldc, const float* const* bias, const SaGemmEpilogue* ep,
                                   void* workspace, size_t workspace_bytes, hipStream_t stream,
                                   const SaGemmOpts* opts) {
    SA_CLEAR_ERR();
    if (opts && opts->colsum && (!trans_a || alpha != 1.f)) return CTC_STATUS_INVALID_VALUE;
    if (M < 0 || N < 0 || K < 0 || nprob < 1 || nprob > kMaxGroup) return CTC_STATUS_INVALID_VALUE;
    if (M == 0 || N == 0) return CTC_STATUS_SUCCESS;
    GemmArgs g;
    g.nprob = nprob;
    g.vecA = (lda & 3) == 0;
    g.vecB = (ldb & 3) == 0;
    for (int p = 0; p < nprob; ++p) {
        if (!A[p] || !B[p] || !C[p]) return CTC_STATUS_INVALID_VALUE;
        g.Ag[p] = A[p]; g.Bg[p] = B[p]; g.Cg[p] = C[p]; g.biasg[p] = bias ? bias[p] : nullptr;
        g.colsumg[p] = (opts && opts->colsum) ? opts->colsum[p] : nullptr;
        g.vecA = g.vecA && (((uintptr_t)A[p] & 15) == 0);
        g.vecB = g.vecB && (((uintptr_t)B[p] & 15) == 0);
    }
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta;
    g.m_inner = ep ? ep->m_inner : 0;
    g.m_mid = ep ? ep->m_mid : 1;
    g.s_outer = ep ? ep->s_outer : 0;
    g.s_mid = ep ? ep->s_mid : 0;
    g.col_stride = ep ? ep->col_stride : 1;
    g.relu = ep ? ep->relu : 0;
    int splits = choose_splits(M, N, K, nprob);
    if (opts && opts->no_split) splits = 1;
    if (splits > 1) {
        const size_t need = (size_t)nprob * splits * ((size_t)M * N + M) * sizeof(float);
        if (!workspace || workspace_bytes < need) splits = 1;  // no room: fall back to one pass (still correct)
    }
    int kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    if (kps < BK) kps = BK;
    splits = K > 0 ? (K + kps - 1) / kps : 1;
    g.k_per_split = kps;
    g.splits = splits;
    g.partial = splits > 1 ? (float*)workspace : nullptr;
    g.cs_partial = splits > 1 ? (float*)workspace + (size_t)nprob * splits * M * N : nullptr;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, nprob * splits);
    // "polite" launches (opts->pad_lds): dynamic LDS on top of the kernel's 66 KB so that a CU admits ONE block of this
    // launch -- what a side-stream GEMM wants while a persistent recurrence kernel holds every CU (gru.hip): the
    // recurrence blocks then always find room, before and after this launch's blocks arrive.
    size_t dyn = 0;
    if (opts && opts->pad_lds) {
        static bool attr_set = false;
        dyn = 81 * 1024 - sizeof(float) * 2 * 2 * BK * LDT;
        if (!attr_set) {
            const void* fns[4] = {(const void*)gemm_f32_kernel<true, true>, (const void*)gemm_f32_kernel<true, false>,
                                  (const void*)gemm_f32_kernel<false, true>, (const void*)gemm_f32_kernel<false, false>};
            for (int i = 0; i < 4; ++i)
                if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess)
                    return CTC_STATUS_EXECUTION_FAILED;
            attr_set = true;
        }
    }
    if (trans_a) {
        if (trans_b) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), dyn, stream, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), dyn, stream, g);
    } else {
        if (trans_b) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), dyn, stream, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), dyn, stream, g);
    }
    SA_CHECK_LAUNCH();
    if (splits > 1) {
        const long total = (long)M * N + M;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256), nprob), dim3(256), 0,
                           stream, g, splits);
        SA_CHECK_LAUNCH();
    }
    return CTC_STATUS_SUCCESS;
}

size_t sa_gemm_group_workspace_bytes(int nprob, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || nprob <= 0) return 0;
    const int s = choose_splits(M, N, K, nprob);
    return s > 1 ? (size_t)nprob * s * ((size_t)M * N + M) * sizeof(float) : 0;
}

ctcStatus_t sa_gemm_f32_impl(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, long lda,
                             const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
                             const SaGemmEpilogue* ep, void* workspace, size_t workspace_bytes,
                             hipStream_t stream) {
    return sa_gemm_f32_group_impl(1, trans_a, trans_b, M, N, K, alpha, &A, lda, &B, ldb, beta, &C, ldc, &bias, ep,
                                  workspace, workspace_bytes, stream, nullptr);
}

extern "C" size_t sa_gemm_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = choose_splits(M, N, K);
    return s > 1 ? (size_t)s * ((size_t)M * N + M) * sizeof(float) : 0;
}

extern "C" ctcStatus_t sa_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A,
                                   long lda, const float* B, long ldb, float beta, float* C, long ldc,
                                   const float* bias, void* workspace, size_t workspace_bytes, void* stream) {
    SA_CLEAR_ERR();
    return sa_gemm_f32_impl(trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, nullptr, workspace,
                            workspace_bytes, (hipStream_t)stream);
}
