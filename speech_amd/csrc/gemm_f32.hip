// gemm_f32.hip -- fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate),
// the GEMM-shaped part of the encoder: nn.GRU's input projections and nn.Linear (reference:
// /root/reference/speech/models/model.py:35-39,126-133) and their weight / input gradients.
//
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] (+ bias[N]) (+ beta * C)
//
// Tiling (CDNA4): 128x128 block tile, BK = 32, 256 threads = 4 waves as 2x2, each wave a 64x64 sub-tile =
// 2x2 MFMA 32x32 tiles (4 x 16 accumulator registers).
// Round 2 main loop (round 1 read one float per MFMA operand from a k-major LDS tile right before using it; with one
// block on a CU -- the small per-chunk products of the GRU wavefront, or a block sharing its CU with a persistent
// recurrence block -- every k-step then exposed an LDS round trip and the kernel ran at ~40 % of the matrix pipe):
//   * the MFMA's two k-lanes are fed k = j and k = j + 16 of the tile (the same permutation on both operands, so the
//     product is unchanged).  An operand that is k-contiguous in memory sits in LDS ROW-major, [row][k] with a 36-float
//     pitch: lane (h, r) then needs 16 CONSECUTIVE floats of row r -- four conflict-free ds_read_b128 -- for all 16
//     MFMAs of the k-tile.  An operand that is m/n-contiguous in memory keeps that orientation in LDS, [k][row] with a
//     132-float pitch, and its fragments are 16 ds_read_b32 (32 consecutive floats per half-wave: conflict-free).
//     Either way the tile goes into LDS with 16-byte stores and NO transpose -- transposing stores of 4-row-strided
//     floats into a 16-byte-aligned pitch land on 2 of the 32 banks (measured: 88 % of the LDS cycles of the
//     weight-gradient product were bank-conflict cycles);
//   * a k-tile is consumed in four quarters of 16 MFMAs per wave with every operand already in registers: the
//     fragments of the next quarter are fetched from LDS while the current one's MFMAs run.  Global loads run TWO
//     tiles ahead through two register stages; a staged tile's 16-byte LDS stores are slipped in between the MFMAs
//     of the first two quarters.  One barrier per k-tile, no LDS or memory latency inside an MFMA run;
//   * interior blocks (full tiles, aligned operands) run a guard-free instance of the loop; edge blocks a guarded one.
// Tall-K products with few output tiles (dW = dA^T X, K = B*T' ~ 16k) are split along K across blockIdx.z into a
// workspace and reduced by a second kernel in a fixed order (deterministic, unlike atomics).
#include <stdlib.h>

#include "common.h"
#include "internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = 36;            // [row][k] tiles: row pitch in floats (16-byte aligned rows, conflict-free b128 reads)
constexpr int KPITCH = 132;          // [k][row] tiles: k pitch in floats
constexpr int TILE_F = BM * PITCH;   // floats reserved per operand tile (>= BK * KPITCH)

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
    // XCD-filtered mode (xcc_mask != 0): the launch has (8 / allowed XCDs) x as many blocks as tiles (+ slack); a block
    // that finds itself on an XCD outside the mask leaves at once, every other block draws ONE tile off `tile_counter`
    // (zeroed by the host) and leaves when none is left.  A side-stream GEMM can so be kept OFF the XCDs a persistent
    // recurrence launch occupies (a bidirectional layer's groups sit on XCDs 0 .. u-1): no shared CUs, no interference
    // -- hipExtStreamCreateWithCUMask cannot express this (measured: its bits select CUs inside every XCD alike,
    // tools/ubench/cumask_probe.hip).  One tile per block, not a persistent loop: the dispatcher hands out a grid in
    // order, so resident long-lived blocks on the idle XCDs would keep the NEXT recurrence launch from starting (its
    // own exit-at-once blocks for those XCDs find no room: profiles/r02_bidirectional_overlap_trace.txt).
    unsigned xcc_mask;
    unsigned* tile_counter;
    int grid_x, grid_y, grid_z;
    // packed kernels only: A's packed row block of output row block `by` is by (by < a_rb_split) or by + a_rb_jump -- a
    // product that uses rows [0, s) and [s + j, ...) of a packed operand shared with another product (sa_gemm_pk_group)
    int a_rb_split, a_rb_jump;
    unsigned a_jump_probs;  // bit p: problem p of the group uses the jump
    int b_kb_stride;        // packed kernels only: k-tiles between B's packed row blocks (0: ceil(K / 16), as A's always are)
    // optional dropout on the output (dropout.h): element at address c is multiplied by the mask factor of index
    // c - drop_base (the output is a window of the masked tensor); off when drop.thresh == 0.  Not with split-K.
    SaDrop drop;
    unsigned drop_stream;
    const float* drop_base;
};

__device__ __forceinline__ int pk_a_rb(const GemmArgs& g, int prob, int rb) {
    return (rb >= g.a_rb_split && ((g.a_jump_probs >> prob) & 1u)) ? rb + g.a_rb_jump : rb;
}

__device__ __forceinline__ long remap_row(const GemmArgs& g, int row) {
    const int q = row / g.m_inner;
    return (long)(q / g.m_mid) * g.s_outer + (long)(q % g.m_mid) * g.s_mid + (row % g.m_inner);
}

// Load this thread's slice of one operand tile (128 rows x BK) into registers: NL = 4 float4 per thread.
// KCONTIG: memory is [rows][k] (k contiguous)  -> row = tid/8 + 32p, k = 4*(tid%8)      (128 B per row: full lines)
// else:    memory is [k][cols] (cols contiguous) -> k = tid/32 + 8p, col = 4*(tid%32)
// FAST: the block's tile is interior, K-tiles are full and 16-byte loads are legal: no guards, no branches.
constexpr int NL = 4;
template <bool KCONTIG, bool FAST>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int r0, int R, int k0, int kend,
                                          int vec, int tid, float4 (&v)[NL]) {
#pragma unroll
    for (int p = 0; p < NL; ++p) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) {
            if (KCONTIG)
                x = *reinterpret_cast<const float4*>(P + (long)(r0 + (tid >> 3) + 32 * p) * ld + k0 + 4 * (tid & 7));
            else
                x = *reinterpret_cast<const float4*>(P + (long)(k0 + (tid >> 5) + 8 * p) * ld + r0 + 4 * (tid & 31));
        } else if (KCONTIG) {
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

// registers -> LDS tile: S[row][k] (pitch PITCH) for a k-contiguous operand, S[k][row] (pitch KPITCH) otherwise
template <bool KCONTIG>
__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const float4 (&v)[NL]) {
#pragma unroll
    for (int p = 0; p < NL; ++p) {
        if (KCONTIG) *reinterpret_cast<float4*>(&S[((tid >> 3) + 32 * p) * PITCH + 4 * (tid & 7)]) = v[p];
        else *reinterpret_cast<float4*>(&S[((tid >> 5) + 8 * p) * KPITCH + 4 * (tid & 31)]) = v[p];
    }
}

// The fragments of a QUARTER of a k-tile for this lane: per operand 2 MFMA tiles x 4 k-steps.  Element (t, j) feeds
// k-step j (of the quarter q) of tile t: k_local = 4 q + j + 16 h for lane (h = lane >> 5, r = lane & 31).
struct QFrag { float a[2][4], b[2][4]; };
// `s` points at the lane's first element: [row][k] tiles  S[(row0 + r) * PITCH + 16 h + 4 q],
//                                         [k][row] tiles  S[(16 h + 4 q) * KPITCH + row0 + r]
template <bool KCONTIG>
__device__ __forceinline__ void read_frag(const float* __restrict__ s, float (&f)[2][4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (KCONTIG) {
            const float4 v = *reinterpret_cast<const float4*>(s + t * 32 * PITCH);
            f[t][0] = v.x; f[t][1] = v.y; f[t][2] = v.z; f[t][3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[t][j] = s[j * KPITCH + t * 32];
        }
    }
}
__device__ __forceinline__ void mma_step(const QFrag& f, int j, f32x16 (&acc)[2][2]) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][j], f.b[0][j], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][j], f.b[1][j], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1][j], f.b[0][j], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1][j], f.b[1][j], acc[1][1], 0, 0, 0);
}
// 16 MFMAs: 4 k-steps x (2 x 2) tiles, all operands in registers
__device__ __forceinline__ void mma_quarter(const QFrag& f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) mma_step(f, j, acc);
}
// the same 16 MFMAs with 4 of the 2 x NL 16-byte LDS stores of a staged tile slipped in between them (PART 0: the A
// tile, PART 1: the B tile): a store issues while the MFMA before it executes, so staging costs no matrix-pipe time
template <bool KCONTIG>
__device__ __forceinline__ void mma_quarter_store(const QFrag& f, f32x16 (&acc)[2][2], float* __restrict__ S, int tid,
                                                  const float4 (&r)[NL]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mma_step(f, j, acc);
        if (KCONTIG) *reinterpret_cast<float4*>(&S[((tid >> 3) + 32 * j) * PITCH + 4 * (tid & 7)]) = r[j];
        else *reinterpret_cast<float4*>(&S[((tid >> 5) + 8 * j) * KPITCH + 4 * (tid & 31)]) = r[j];
    }
}

// One k-tile of the steady state, in four quarters of 16 MFMAs per wave.  On entry: LDS buffer `cur` holds tile `it`,
// fx its first quarter's fragments, (sa, sb) the staged registers of tile it + 1 (their loads were issued a whole tile
// ago).  LOAD: issue the global loads of tile it + 2 into (na, nb).  STORE: tile it + 1 goes to the other LDS buffer
// between the MFMAs of quarters 0 and 1.  The barrier sits before quarter 3: by then every wave has READ its last
// fragments of buffer `cur` (the next tile's stores may overwrite it) and the other buffer is complete; the next
// tile's first fragments are fetched right behind it and land during quarter 3.
template <bool TA, bool TB, bool FAST, bool LOAD, bool STORE>
__device__ __forceinline__ void gemm_ktile(const GemmArgs& g, const float* __restrict__ gA,
                                           const float* __restrict__ gB, float* smem, int m0, int n0, int k_next2,
                                           int kend, int tid, int cur, int fa, int fb, f32x16 (&acc)[2][2],
                                           QFrag& fx, QFrag& fy, float4 (&sa)[NL], float4 (&sb)[NL],
                                           float4 (&na)[NL], float4 (&nb)[NL], bool do_colsum, float4& csum) {
    constexpr bool AK = !TA, BKC = TB;
    constexpr int qa = AK ? 4 : 4 * KPITCH, qb = BKC ? 4 : 4 * KPITCH;  // step from one quarter to the next
    float* Ac = smem + cur * (2 * TILE_F);
    float* Bc = Ac + TILE_F;
    float* An = smem + (cur ^ 1) * (2 * TILE_F);
    float* Bn = An + TILE_F;
    if (LOAD) {
        load_tile<!TA, FAST>(gA, g.lda, m0, g.M, k_next2, kend, g.vecA, tid, na);
        load_tile<TB, FAST>(gB, g.ldb, n0, g.N, k_next2, kend, g.vecB, tid, nb);
    }
    read_frag<AK>(Ac + fa + qa, fy.a);
    read_frag<BKC>(Bc + fb + qb, fy.b);
    if (STORE) {
        if (TA && do_colsum) {  // a thread's NL float4 of an A tile are 4 consecutive m at NL different k
#pragma unroll
            for (int p = 0; p < NL; ++p) { csum.x += sa[p].x; csum.y += sa[p].y; csum.z += sa[p].z; csum.w += sa[p].w; }
        }
        mma_quarter_store<AK>(fx, acc, An, tid, sa);
    } else {
        mma_quarter(fx, acc);
    }
    read_frag<AK>(Ac + fa + 2 * qa, fx.a);
    read_frag<BKC>(Bc + fb + 2 * qb, fx.b);
    if (STORE) mma_quarter_store<BKC>(fy, acc, Bn, tid, sb);
    else mma_quarter(fy, acc);
    read_frag<AK>(Ac + fa + 3 * qa, fy.a);
    read_frag<BKC>(Bc + fb + 3 * qb, fy.b);
    mma_quarter(fx, acc);
    __syncthreads();
    if (STORE) {
        read_frag<AK>(An + fa, fx.a);
        read_frag<BKC>(Bn + fb, fx.b);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the barrier BEFORE the last quarter's MFMAs
    mma_quarter(fy, acc);
}

// The same k-tile with ONE register stage (the XCD-filtered kernel: 32 registers fewer, see gemm_f32_filtered_kernel): the
// loads of tile it + 1 go out at the top of tile `it` and are stored between the MFMAs of quarter 2, half a tile later.
template <bool TA, bool TB, bool FAST, bool MORE>
__device__ __forceinline__ void gemm_ktile_d1(const GemmArgs& g, const float* __restrict__ gA,
                                              const float* __restrict__ gB, float* smem, int m0, int n0, int k_next,
                                              int kend, int tid, int cur, int fa, int fb, f32x16 (&acc)[2][2],
                                              QFrag& fx, QFrag& fy, float4 (&sa)[NL], float4 (&sb)[NL], bool do_colsum,
                                              float4& csum) {
    constexpr bool AK = !TA, BKC = TB;
    constexpr int qa = AK ? 4 : 4 * KPITCH, qb = BKC ? 4 : 4 * KPITCH;
    float* Ac = smem + cur * (2 * TILE_F);
    float* Bc = Ac + TILE_F;
    float* An = smem + (cur ^ 1) * (2 * TILE_F);
    float* Bn = An + TILE_F;
    if (MORE) {
        load_tile<!TA, FAST>(gA, g.lda, m0, g.M, k_next, kend, g.vecA, tid, sa);
        load_tile<TB, FAST>(gB, g.ldb, n0, g.N, k_next, kend, g.vecB, tid, sb);
    }
    read_frag<AK>(Ac + fa + qa, fy.a);
    read_frag<BKC>(Bc + fb + qb, fy.b);
    mma_quarter(fx, acc);
    read_frag<AK>(Ac + fa + 2 * qa, fx.a);
    read_frag<BKC>(Bc + fb + 2 * qb, fx.b);
    mma_quarter(fy, acc);
    read_frag<AK>(Ac + fa + 3 * qa, fy.a);
    read_frag<BKC>(Bc + fb + 3 * qb, fy.b);
    if (MORE) {
        if (TA && do_colsum) {
#pragma unroll
            for (int p = 0; p < NL; ++p) { csum.x += sa[p].x; csum.y += sa[p].y; csum.z += sa[p].z; csum.w += sa[p].w; }
        }
        store_tile<!TA>(An, tid, sa);
        store_tile<TB>(Bn, tid, sb);
    }
    mma_quarter(fx, acc);
    __syncthreads();
    if (MORE) {
        read_frag<AK>(An + fa, fx.a);
        read_frag<BKC>(Bn + fb, fx.b);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_quarter(fy, acc);
}

template <bool TA, bool TB, bool FAST>
__device__ __forceinline__ void gemm_mainloop_d1(const GemmArgs& g, const float* __restrict__ gA,
                                                 const float* __restrict__ gB, float* smem, int m0, int n0, int kbeg,
                                                 int kend, int tid, f32x16 (&acc)[2][2], bool do_colsum, float4& csum) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    constexpr bool AK = !TA, BKC = TB;
    const int fa = AK ? (wm * 64 + (lane & 31)) * PITCH + 16 * (lane >> 5) : 16 * (lane >> 5) * KPITCH + wm * 64 + (lane & 31);
    const int fb = BKC ? (wn * 64 + (lane & 31)) * PITCH + 16 * (lane >> 5) : 16 * (lane >> 5) * KPITCH + wn * 64 + (lane & 31);
    float4 ra[NL], rb[NL];
    if (ntiles <= 0) return;
    load_tile<!TA, FAST>(gA, g.lda, m0, g.M, kbeg, kend, g.vecA, tid, ra);
    load_tile<TB, FAST>(gB, g.ldb, n0, g.N, kbeg, kend, g.vecB, tid, rb);
    if (TA && do_colsum) {
#pragma unroll
        for (int p = 0; p < NL; ++p) { csum.x += ra[p].x; csum.y += ra[p].y; csum.z += ra[p].z; csum.w += ra[p].w; }
    }
    store_tile<!TA>(smem, tid, ra);
    store_tile<TB>(smem + TILE_F, tid, rb);
    __syncthreads();
    QFrag f0, f1;
    read_frag<AK>(smem + fa, f0.a);
    read_frag<BKC>(smem + TILE_F + fb, f0.b);
    int it = 0;
    for (; it + 1 < ntiles; ++it)
        gemm_ktile_d1<TA, TB, FAST, true>(g, gA, gB, smem, m0, n0, kbeg + (it + 1) * BK, kend, tid, it & 1, fa, fb, acc, f0,
                                          f1, ra, rb, do_colsum, csum);
    gemm_ktile_d1<TA, TB, FAST, false>(g, gA, gB, smem, m0, n0, 0, kend, tid, it & 1, fa, fb, acc, f0, f1, ra, rb,
                                       do_colsum, csum);
}

// The K loop (see the file header).  LDS: stage s holds the A tile at smem + s * 2 * TILE_F and the B tile behind it.
// Global loads run TWO tiles ahead of the MFMAs (two register stages, the loop is unrolled by two so that they swap
// roles without moves): a load has a whole k-tile (>= 4096 matrix-pipe cycles) to land before its LDS store.
template <bool TA, bool TB, bool FAST>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& g, const float* __restrict__ gA,
                                              const float* __restrict__ gB, float* smem, int m0, int n0, int kbeg,
                                              int kend, int tid, f32x16 (&acc)[2][2], bool do_colsum, float4& csum) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    // this lane's first fragment element (half 0) in an A tile / a B tile
    constexpr bool AK = !TA, BKC = TB;  // operand is k-contiguous in memory -> [row][k] tile
    const int fa = AK ? (wm * 64 + (lane & 31)) * PITCH + 16 * (lane >> 5) : 16 * (lane >> 5) * KPITCH + wm * 64 + (lane & 31);
    const int fb = BKC ? (wn * 64 + (lane & 31)) * PITCH + 16 * (lane >> 5) : 16 * (lane >> 5) * KPITCH + wn * 64 + (lane & 31);
    float4 ra[NL], rb[NL], qa[NL], qb[NL];
    if (ntiles <= 0) return;
    load_tile<!TA, FAST>(gA, g.lda, m0, g.M, kbeg, kend, g.vecA, tid, ra);
    load_tile<TB, FAST>(gB, g.ldb, n0, g.N, kbeg, kend, g.vecB, tid, rb);
    if (TA && do_colsum) {
#pragma unroll
        for (int p = 0; p < NL; ++p) { csum.x += ra[p].x; csum.y += ra[p].y; csum.z += ra[p].z; csum.w += ra[p].w; }
    }
    store_tile<!TA>(smem, tid, ra);
    store_tile<TB>(smem + TILE_F, tid, rb);
    if (ntiles > 1) {  // tile 1 -> stage (ra, rb)
        load_tile<!TA, FAST>(gA, g.lda, m0, g.M, kbeg + BK, kend, g.vecA, tid, ra);
        load_tile<TB, FAST>(gB, g.ldb, n0, g.N, kbeg + BK, kend, g.vecB, tid, rb);
    }
    __syncthreads();
    QFrag f0, f1;
    read_frag<AK>(smem + fa, f0.a);
    read_frag<BKC>(smem + TILE_F + fb, f0.b);
    int it = 0;
    for (; it + 3 < ntiles; it += 2) {  // steady state: tiles it, it + 1 (loads of it + 2, it + 3 go out)
        gemm_ktile<TA, TB, FAST, true, true>(g, gA, gB, smem, m0, n0, kbeg + (it + 2) * BK, kend, tid, 0, fa, fb, acc, f0,
                                             f1, ra, rb, qa, qb, do_colsum, csum);
        gemm_ktile<TA, TB, FAST, true, true>(g, gA, gB, smem, m0, n0, kbeg + (it + 3) * BK, kend, tid, 1, fa, fb, acc, f0,
                                             f1, qa, qb, ra, rb, do_colsum, csum);
    }
    // the last 1..3 tiles (`it` is even: LDS buffer 0 holds tile `it`, (ra, rb) stage tile it + 1)
    const int left = ntiles - it;
    if (left == 3) {
        gemm_ktile<TA, TB, FAST, true, true>(g, gA, gB, smem, m0, n0, kbeg + (it + 2) * BK, kend, tid, 0, fa, fb, acc, f0,
                                             f1, ra, rb, qa, qb, do_colsum, csum);
        gemm_ktile<TA, TB, FAST, false, true>(g, gA, gB, smem, m0, n0, 0, kend, tid, 1, fa, fb, acc, f0, f1, qa, qb, ra,
                                              rb, do_colsum, csum);
        gemm_ktile<TA, TB, FAST, false, false>(g, gA, gB, smem, m0, n0, 0, kend, tid, 0, fa, fb, acc, f0, f1, ra, rb, qa,
                                               qb, do_colsum, csum);
    } else if (left == 2) {
        gemm_ktile<TA, TB, FAST, false, true>(g, gA, gB, smem, m0, n0, 0, kend, tid, 0, fa, fb, acc, f0, f1, ra, rb, qa,
                                              qb, do_colsum, csum);
        gemm_ktile<TA, TB, FAST, false, false>(g, gA, gB, smem, m0, n0, 0, kend, tid, 1, fa, fb, acc, f0, f1, qa, qb, ra,
                                               rb, do_colsum, csum);
    } else {
        gemm_ktile<TA, TB, FAST, false, false>(g, gA, gB, smem, m0, n0, 0, kend, tid, 0, fa, fb, acc, f0, f1, ra, rb, qa,
                                               qb, do_colsum, csum);
    }
}

// ------------------------------------------------------------------------------- split-bf16 GEMM on packed operands ("x6")
// The f32-input MFMA runs at the fp32 VECTOR rate (157 TFLOP/s); the bf16 MFMA is 16x faster.  An fp32 number is the
// exact sum of three bf16 pieces, a = a1 + a2 + a3 (each piece the round-to-nearest bf16 of what the previous ones left,
// v_cvt_pk_bf16_f32; 3 x 8 significant bits and a sign each cover the 24), so an fp32 product is the sum of nine
// bf16 x bf16 products, every one of them EXACT in the fp32 accumulator.  Six of them carry everything down to 2^-26 of
// the product:
//     a b  =  a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)  +  [a2 b3 + a3 b2 + a3 b3:  < 2^-26 |a b|, dropped]
// -- a quarter of an fp32 ulp of the product, below the rounding of the fp32 accumulation every GEMM has.  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles, K = 2): 192 instead of
// 512 matrix-pipe cycles per 16 k.  Accumulation is fp32 throughout, as in the exact kernel (measured against an fp64
// product the split kernel's error is the exact kernel's: tools/gemm_bench.py, tests/test_gpu_blocks.py).
// option gemm.exact = 1 selects the f32-input MFMA kernel everywhere; small products, the XCD-filtered launches, the direct
// convolutions and the recurrence kernels always use f32-input MFMAs.
//
// Splitting costs ~5.5 VALU instructions per element.  Done while a tile is staged (first version of this kernel) every
// block redoes it for every tile it touches and the loop is issue-bound at 39 % of the matrix pipe (163 TFLOP/s on
// k-contiguous operands, less with a transposing store).  So an operand is split ONCE, by a PACK kernel, into the
// layout the main loop wants, and the main loop is nothing but LDS-DMA, fragment reads and MFMAs:
//   * packed operand of a logical X[R][K] (R = M or N): tiles of 128 rows x 16 k, tile (rb, kb) at (rb KB + kb) 12 KB;
//     inside a tile  chunk(plane p, row r, half h) = p 4096 + r 32 + (h ^ ((r >> 3) & 1)) 16  bytes holds the eight
//     bf16 pieces k = 8 h .. 8 h + 7 of row r (the XOR keeps the 16 lanes of a ds_read_b128 group on disjoint banks).
//     Rows beyond R and k beyond K are zero, so the main loop has no edge cases; both memory orientations of an
//     operand (k-contiguous, m/n-contiguous) pack into the same layout, so ONE kernel serves all four transpose forms;
//   * a block streams its A and B tiles -- consecutive in memory along k -- into a 3-stage LDS ring with
//     global_load_lds_dwordx4 (no registers, no ds_write; 6 per wave and stage), two stages ahead of the MFMAs;
//   * lane (h, r) feeds an MFMA the chunk (p, r, h) of A and of B alike, so the hardware's own k numbering never matters.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PK_K = 16;                       // k per packed tile
constexpr int PK_PLANE = BM * PK_K * 2;        // 4096 bytes: one bf16 plane of a tile
constexpr int PK_TILE = 3 * PK_PLANE;          // 12288 bytes
constexpr int PK_STAGE = 2 * PK_TILE;          // A tile + B tile
constexpr int PK_NST = 3;                      // LDS ring: 72 KB per block, two blocks per CU

__host__ __device__ inline size_t pk_bytes(int R, int K) {
    return (size_t)((R + BM - 1) / BM) * (size_t)((K + PK_K - 1) / PK_K) * PK_TILE;
}
__host__ __device__ __forceinline__ int pk_off(int p, int row, int h) {
    return p * PK_PLANE + row * 32 + ((h ^ ((row >> 3) & 1)) << 4);
}

// the split itself: sa_split2 / sa_cvt_pk_bf16 (common.h)
__device__ __forceinline__ void split2(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    sa_split2(x, y, p1, p2, p3);
}

struct PackArgs {
    const float* src[kMaxGroup];
    char* dst;             // problem p at dst + p * dst_stride
    size_t dst_stride;
    long ld;
    int R, K, KB, kt;      // rows, reduction length, k-tiles, k-tiles per block
    int vec;               // 16-byte loads are legal (k-contiguous form)
    float* cs_part;        // m-contiguous form: row sums [problem][part][Rpad] (part = 2 blockIdx.x + wave / 2), or null
    int Rpad;
    // m-contiguous form: rows >= R_lo come from src_hi (same ld), row r at src_hi[p][k * ld + (r - R_lo)]; R_lo = R: unused
    const float* src_hi[kMaxGroup];
    int R_lo;
    // tiles per packed row block (KB, or more: several operands interleaved along k -- problem p at k-tile offset
    // p * dst_stride / tile bytes inside each row block, the products' reduction then runs over all of them)
    int kb_stride;
    // m-contiguous form: the matrix starts kb_lead tiles into each row block, and the tiles in front of it are written as
    // zeros (zero_lead) -- an operand that a product also reads SHIFTED by kb_lead k-tiles (h_prev of a GRU layer is its
    // output one time step earlier, zeros at t = 0)
    int kb_lead, zero_lead;
};

// X[r][k] = src[r * ld + k] (k-contiguous memory).  grid (ceil(KB / kt), RB, nprob); thread t: rows t/4 and t/4 + 64,
// k = 4 (t % 4) .. + 3: a wave reads 16 rows x 64 bytes and writes 16 rows x 32 bytes (contiguous) per plane.
__global__ __launch_bounds__(256) void pk_pack_kcontig_kernel(PackArgs a) {
    const float* __restrict__ src = a.src[blockIdx.z];
    char* __restrict__ dst = a.dst + blockIdx.z * a.dst_stride + (size_t)blockIdx.y * a.kb_stride * PK_TILE;
    const int tid = threadIdx.x, k4 = tid & 3;
    const int kb_end = min(a.KB, ((int)blockIdx.x + 1) * a.kt);
    // (r6) Whole k-tiles of a full-width interior block: FOUR k-tiles' loads (eight 16-byte loads per thread) in flight before the first
    // split -- one k-tile at a time was a chain of kt L2 / HBM round trips per block (x of the S-LIBRI step: 52 us for 127 MB).
    int kb = blockIdx.x * a.kt;
    if (a.vec && blockIdx.y * BM + BM <= a.R) {
        for (; kb + 4 <= kb_end && (kb + 4) * PK_K <= a.K; kb += 4) {
            float4 v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    v[u][i] = *reinterpret_cast<const float4*>(src + (long)(blockIdx.y * BM + (tid >> 2) + 64 * i) * a.ld + (kb + u) * PK_K + 4 * k4);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned a1, a2, a3, b1, b2, b3;
                    split2(v[u][i].x, v[u][i].y, a1, a2, a3);
                    split2(v[u][i].z, v[u][i].w, b1, b2, b3);
                    char* d = dst + (size_t)(kb + u) * PK_TILE + pk_off(0, (tid >> 2) + 64 * i, k4 >> 1) + 8 * (k4 & 1);
                    *reinterpret_cast<uint2*>(d) = make_uint2(a1, b1);
                    *reinterpret_cast<uint2*>(d + PK_PLANE) = make_uint2(a2, b2);
                    *reinterpret_cast<uint2*>(d + 2 * PK_PLANE) = make_uint2(a3, b3);
                }
        }
    }
    for (; kb < kb_end; ++kb) {
        char* tile = dst + (size_t)kb * PK_TILE;
        const int k = kb * PK_K + 4 * k4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = (tid >> 2) + 64 * i, row = blockIdx.y * BM + rl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < a.R) {
                const float* q = src + (long)row * a.ld + k;
                if (a.vec && k + 3 < a.K) {
                    v = *reinterpret_cast<const float4*>(q);
                } else {
                    if (k + 0 < a.K) v.x = q[0];
                    if (k + 1 < a.K) v.y = q[1];
                    if (k + 2 < a.K) v.z = q[2];
                    if (k + 3 < a.K) v.w = q[3];
                }
            }
            unsigned a1, a2, a3, b1, b2, b3;
            split2(v.x, v.y, a1, a2, a3);
            split2(v.z, v.w, b1, b2, b3);
            char* d = tile + pk_off(0, rl, k4 >> 1) + 8 * (k4 & 1);
            *reinterpret_cast<uint2*>(d) = make_uint2(a1, b1);
            *reinterpret_cast<uint2*>(d + PK_PLANE) = make_uint2(a2, b2);
            *reinterpret_cast<uint2*>(d + 2 * PK_PLANE) = make_uint2(a3, b3);
        }
    }
}

// X[r][k] = src[k * ld + r] (m/n-contiguous memory).  grid (ceil(KB / kt), RB, nprob); wave w: rows (w & 1) 64 + lane of
// the block's 128, k-tiles of parity w >> 1: a wave reads 64 consecutive floats per k (coalesced) and each thread
// ends up with the 16 k of ITS row -- the transposition costs nothing.  Optional row sums (the bias gradients that
// belong to a weight gradient dW = dA^T X: column sums of dA): one partial per (block, tile parity), folded in order.
__global__ __launch_bounds__(256) void pk_pack_mcontig_kernel(PackArgs a) {
    const float* __restrict__ src = a.src[blockIdx.z];
    char* __restrict__ dst = a.dst + blockIdx.z * a.dst_stride + (size_t)blockIdx.y * a.kb_stride * PK_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rl = (wave & 1) * 64 + lane, row = blockIdx.y * BM + rl;
    const bool live = row < a.R;
    const float* __restrict__ col = !live ? src : (row < a.R_lo ? src + row : a.src_hi[blockIdx.z] + (row - a.R_lo));
    const int kb_end = min(a.KB, ((int)blockIdx.x + 1) * a.kt);
    float rsum = 0.f;
    for (int kb = blockIdx.x * a.kt + (wave >> 1); kb < kb_end; kb += 2) {
        float v[PK_K];
#pragma unroll
        for (int kk = 0; kk < PK_K; ++kk) {
            const int k = kb * PK_K + kk;
            v[kk] = (live && k < a.K) ? col[(long)k * a.ld] : 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < PK_K; ++kk) rsum += v[kk];
        unsigned p1[8], p2[8], p3[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) split2(v[2 * q], v[2 * q + 1], p1[q], p2[q], p3[q]);
        char* tile = dst + (size_t)(kb + a.kb_lead) * PK_TILE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            char* d = tile + pk_off(0, rl, h);
            *reinterpret_cast<uint4*>(d) = make_uint4(p1[4 * h], p1[4 * h + 1], p1[4 * h + 2], p1[4 * h + 3]);
            *reinterpret_cast<uint4*>(d + PK_PLANE) = make_uint4(p2[4 * h], p2[4 * h + 1], p2[4 * h + 2], p2[4 * h + 3]);
            *reinterpret_cast<uint4*>(d + 2 * PK_PLANE) = make_uint4(p3[4 * h], p3[4 * h + 1], p3[4 * h + 2], p3[4 * h + 3]);
        }
    }
    if (a.cs_part)
        a.cs_part[((size_t)blockIdx.z * (2 * gridDim.x) + 2 * blockIdx.x + (wave >> 1)) * a.Rpad + blockIdx.y * BM + rl] = rsum;
    if (a.zero_lead && blockIdx.x == 0)  // the kb_lead tiles in front of the matrix: 768 16-byte chunks each
        for (int e = threadIdx.x; e < a.kb_lead * (PK_TILE / 16); e += 256) reinterpret_cast<uint4*>(dst)[e] = make_uint4(0u, 0u, 0u, 0u);
}

// colsum[p][m] = (beta ? beta * colsum : 0) + sum over the parts: 16 lanes per column add every 16th part in order,
// then a fixed xor tree folds the 16 sums (deterministic; a serial loop over ~250 parts per thread took 95 us).
// grid (ceil(M / 16), nprob), 256 threads: lane j = tid & 15 of column m = 16 blockIdx.x + (tid >> 4).
__global__ __launch_bounds__(256) void pk_colsum_fold_kernel(const float* __restrict__ part, int nparts, int Rpad, int M,
                                                             GemmArgs g) {
    const int m = blockIdx.x * 16 + (threadIdx.x >> 4), j = threadIdx.x & 15, prob = blockIdx.y;
    float t = 0.f;
    if (m < M)
        for (int q = j; q < nparts; q += 16) t += part[((size_t)prob * nparts + q) * Rpad + m];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (m >= M || j != 0 || !g.colsumg[prob]) return;
    float* o = g.colsumg[prob] + m;
    *o = g.beta != 0.f ? g.beta * *o + t : t;
}

// XCD-aware tile order for the packed kernels.  Workgroup w of a launch runs on XCD w % 8 (observed, a speed matter only),
// each XCD has its own 4 MB L2, and a block streams 24 - 48 KB of operand tiles per k-step: with the plain blockIdx ->
// tile map the blocks that SHARE an A row block (neighbours in x) or a B row block sit on eight different XCDs and every
// L2 fetches every tile for itself (weight gradients: 48 tiles x 996 k-steps x 24 KB = 1.15 GB pulled for 196 MB of
// operands).  Here XCD x takes a CONTIGUOUS range of the (split, row, column) tile order, so the blocks resident on one XCD
// are neighbours: they walk the same k range of the same few row blocks and hit in their L2.
__device__ __forceinline__ void pk_tile_of_block(int gx, int gy, int gz, int& bx, int& by, int& bz) {
    const int total = gx * gy * gz;
    const int id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int x = id & 7, slot = id >> 3;
    const int per = total >> 3, rem = total & 7;
    const int t = x * per + (x < rem ? x : rem) + slot;  // XCD x owns tiles [x per + min(x, rem), + per + (x < rem))
    bx = t % gx;
    const int r = t / gx;
    by = r % gy;
    bz = r / gy;
}

struct SFrag { bf16x8 a[2][3], b[2][3]; };  // [tile][plane]
__device__ __forceinline__ void sread_frag(const char* __restrict__ s, bf16x8 (&f)[2][3]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            f[t][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(s + t * 1024 + pl * PK_PLANE));
}
// 24 MFMAs: (2 x 2) tiles x the six piece products, the smallest first
__device__ __forceinline__ void smma_tile(const SFrag& f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int o = 0; o < 6; ++o) {
        const int pa = o == 0 ? 2 : (o == 1 ? 0 : (o == 2 ? 1 : (o == 3 ? 0 : (o == 4 ? 1 : 0))));
        const int pb = o == 0 ? 0 : (o == 1 ? 2 : (o == 2 ? 1 : (o == 3 ? 1 : (o == 4 ? 0 : 0))));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][pa], f.b[j][pb], acc[i][j], 0, 0, 0);
    }
}
// one stage = the A tile and the B tile of one k-tile: 24 KB = 24 wave-instructions of 1 KB, 6 per wave
__device__ __forceinline__ void pk_issue(const char* __restrict__ At, const char* __restrict__ Bt, char* slot, int wave,
                                         int lane) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int ci = wave * 6 + j;  // wave-uniform: 1 KB piece of the stage (0 .. 11: A, 12 .. 23: B)
        const char* src = (ci < 12 ? At + ci * 1024 : Bt + (ci - 12) * 1024) + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(slot + ci * 1024), 16, 0, 0);
    }
}
// k-tiles [kb0, kb1) of row block `by` of A and `bx` of B
__device__ __forceinline__ void pk_mainloop(const char* __restrict__ Apk, const char* __restrict__ Bpk, int KB, int KBb,
                                            char* smem, int by, int bx, int kb0, int kb1, int tid, f32x16 (&acc)[2][2]) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nt = kb1 - kb0;
    if (nt <= 0) return;
    const char* At = Apk + ((size_t)by * KB + kb0) * PK_TILE;
    const char* Bt = Bpk + ((size_t)bx * KBb + kb0) * PK_TILE;
    const int fa = pk_off(0, wm * 64 + (lane & 31), lane >> 5);
    const int fb = PK_TILE + pk_off(0, wn * 64 + (lane & 31), lane >> 5);
    pk_issue(At, Bt, smem, wave, lane);
    {
        const size_t o1 = (size_t)(nt > 1 ? 1 : 0) * PK_TILE;
        pk_issue(At + o1, Bt + o1, smem + PK_STAGE, wave, lane);
    }
    // Per tile: [wait: tile it + 1 has landed, the fragments of tile `it` are in registers] -> barrier -> DMA of tile
    // it + 3 into the slot tile `it` just left -> fragment reads of tile it + 1 (they land during the MFMAs) -> 24 MFMAs
    // on tile `it`.  PK_NST = 3 slots hold tiles it + 1, it + 2 and (arriving) it + 3.
    {
        const size_t o2 = (size_t)(nt > 2 ? 2 : nt - 1) * PK_TILE;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile 0 (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        pk_issue(At + o2, Bt + o2, smem + 2 * PK_STAGE, wave, lane);
    }
    SFrag f0, f1;
    sread_frag(smem + fa, f0.a);
    sread_frag(smem + fb, f0.b);
    int nxt = 1;  // ring slot of tile it + 1
    auto step = [&](int it, SFrag& fc, SFrag& fn) {
        // tile it + 1 landed (it + 2 and it + 3... are the 12 newest); fc's reads returned; everyone's likewise after the barrier
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int nx = it + 3 < nt ? it + 3 : nt - 1;      // past the end: re-read the last tile into a slot nobody reads
        const int free_slot = nxt == 0 ? 2 : nxt - 1;      // the slot of tile `it`
        sread_frag(smem + nxt * PK_STAGE + fa, fn.a);
        sread_frag(smem + nxt * PK_STAGE + fb, fn.b);
        __builtin_amdgcn_sched_barrier(0);  // the 12 fragment reads go out FIRST (hipcc would sink them to their use)
        pk_issue(At + (size_t)nx * PK_TILE, Bt + (size_t)nx * PK_TILE, smem + free_slot * PK_STAGE, wave, lane);
        smma_tile(fc, acc);
#pragma unroll
        for (int k = 0; k < 6; ++k) {  // one LDS-DMA instruction behind every fourth MFMA
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        nxt = nxt == 2 ? 0 : nxt + 1;
    };
    int it = 0;
    for (; it + 1 < nt; it += 2) {  // unrolled by two: the fragment sets swap roles without moves
        step(it, f0, f1);
        step(it + 1, f1, f0);
    }
    if (it < nt) step(it, f0, f1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring is reused by the epilogue of nothing, but leave it quiet
}

// Epilogue of a wave's TI x TJ accumulator tiles whose first element is (row0, col0).
// 32x32 C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
template <int TI, int TJ>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, int prob, int bz, int row0, int col0, int lane,
                                              f32x16 (&acc)[TI][TJ]) {
    const bool splitk = g.partial != nullptr;
    float* __restrict__ gC = g.Cg[prob];
    const float* __restrict__ gbias = g.biasg[prob];
    // Whole tiles of the two plain forms (split-K partial sums; alpha acc + bias) take a loop without a branch: the general
    // loop below tests six things per element, and every branch around a store costs the wait for the store before it
    // (the same arithmetic: bit-identical results).
    if (row0 + TI * 32 <= g.M && col0 + TJ * 32 <= g.N) {
        if (splitk) {
            float* __restrict__ p0 = g.partial + ((long)bz * g.M + row0 + 4 * (lane >> 5)) * g.N + col0 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        p0[(long)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.N + j * 32] = acc[i][j][r];
            return;
        }
        if (g.m_inner <= 0 && g.beta == 0.f && !g.relu && !g.drop.thresh) {
            float* __restrict__ c0 = gC + (long)(row0 + 4 * (lane >> 5)) * g.ldc + col0 + (lane & 31);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const float bv = gbias ? gbias[col0 + j * 32 + (lane & 31)] : 0.f;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        c0[(long)(i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + j * 32] = g.alpha * acc[i][j][r] + bv;
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = col0 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            float bv = 0.f;
            if (!splitk && gbias) bv = gbias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                if (splitk) {
                    g.partial[((long)bz * g.M + row) * g.N + col] = acc[i][j][r];
                } else {
                    float* c = g.m_inner > 0 ? gC + remap_row(g, row) + col * g.col_stride
                                             : gC + (long)row * g.ldc + col;
                    float v = g.alpha * acc[i][j][r] + bv;
                    if (g.beta != 0.f) v += g.beta * *c;
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (g.drop.thresh) v *= sa_drop_factor(g.drop, g.drop_stream, (uint64_t)(c - g.drop_base));
                    *c = v;
                }
            }
        }
    }
}

// TA: A is stored (K, M) (m-contiguous);  !TA: A is stored (M, K) (k-contiguous)
// TB: B is stored (N, K) (k-contiguous);  !TB: B is stored (K, N) (n-contiguous)
// DEPTH 2 / 1: the f32-input MFMA main loops; DEPTH 0: the split-bf16 main loop
template <bool TA, bool TB, int DEPTH = 2>
__device__ __forceinline__ void gemm_block(const GemmArgs& g, float* smem, int bx, int by, int bz) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;
    const int prob = bz / g.splits, split = bz - prob * g.splits;
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
    const bool do_colsum = DEPTH != 0 && TA && g.colsumg[prob] != nullptr && bx == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DEPTH == 0) {  // packed split-bf16 operands: g.Ag / g.Bg hold PACKED tiles (see pk_mainloop), no edge cases
        pk_mainloop(reinterpret_cast<const char*>(gA), reinterpret_cast<const char*>(gB), (g.K + PK_K - 1) / PK_K,
                    g.b_kb_stride > 0 ? g.b_kb_stride : (g.K + PK_K - 1) / PK_K, reinterpret_cast<char*>(smem),
                    pk_a_rb(g, prob, by), bx, kbeg / PK_K,
                    (kend + PK_K - 1) / PK_K, tid, acc);
    } else if (DEPTH == 2) {
        if (fast) gemm_mainloop<TA, TB, true>(g, gA, gB, smem, m0, n0, kbeg, kend, tid, acc, do_colsum, csum);
        else gemm_mainloop<TA, TB, false>(g, gA, gB, smem, m0, n0, kbeg, kend, tid, acc, do_colsum, csum);
    } else {
        if (fast) gemm_mainloop_d1<TA, TB, true>(g, gA, gB, smem, m0, n0, kbeg, kend, tid, acc, do_colsum, csum);
        else gemm_mainloop_d1<TA, TB, false>(g, gA, gB, smem, m0, n0, kbeg, kend, tid, acc, do_colsum, csum);
    }
    if (do_colsum) {  // fold the 8 k-rows of threads (tid >> 5) that share 4 columns; fixed order: deterministic
        float* cs = smem;  // the mainloop's last barrier has released the tiles
        *reinterpret_cast<float4*>(&cs[(tid >> 5) * BM + 4 * (tid & 31)]) = csum;
        __syncthreads();
        if (tid < BM && m0 + tid < g.M) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += cs[r * BM + tid];
            if (g.partial) {
                g.cs_partial[(long)bz * g.M + m0 + tid] = t;
            } else {
                float* o = g.colsumg[prob] + m0 + tid;
                *o = g.beta != 0.f ? g.beta * *o + t : t;
            }
        }
    }

    gemm_epilogue<2, 2>(g, prob, bz, m0 + wm * 64, n0 + wn * 64, lane, acc);
}

__device__ __forceinline__ int gemm_xcc_id() {  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE_F];
    gemm_block<TA, TB>(g, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The same product with 256 x 256 block tiles: 512 threads = 8 waves as 2 (M) x 4 (N), a wave 128 x 64 = 4 x 2 MFMA tiles.
// Why: the bf16 pieces are 1.5 x the bytes of the fp32 operands and the matrix pipe runs 2.67 x faster, so a 128 x 128 tile
// (21 flop per byte moved into LDS) asks the L2 -> LDS path for ~20 TB/s at the matrix pipe's rate -- more than half of
// what the L2s deliver at best; measured, the 128-tile kernel stalls at 47 % of the pipe.  256 x 256 tiles move half the
// bytes per flop (44 flop / byte) and issue half the LDS-DMA instructions per MFMA (6 per wave per 48).  One block per
// CU (3 x 48 KB of LDS); fragments are read at the top of a k-tile in the order the MFMAs use them (B first, then the A
// tiles one by one), so only the first nine reads are exposed, and the second wave of the SIMD covers them.
constexpr int PKB_STAGE = 4 * PK_TILE;  // two A row blocks + two B row blocks of one k-tile: 48 KB
__device__ __forceinline__ void pkb_issue(const char* __restrict__ A0, const char* __restrict__ A1,
                                          const char* __restrict__ B0, const char* __restrict__ B1, char* slot, int wave,
                                          int lane) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int ci = wave * 6 + j;  // wave-uniform 1 KB piece: 0..11 A0, 12..23 A1, 24..35 B0, 36..47 B1
        const char* base = ci < 12 ? A0 : (ci < 24 ? A1 : (ci < 36 ? B0 : B1));
        const char* src = base + (ci % 12) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(slot + ci * 1024), 16, 0, 0);
    }
}
__global__ __launch_bounds__(512, 2) void gemm_pk256_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char pkbsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int bx, by, bz;
    pk_tile_of_block(gridDim.x, gridDim.y, gridDim.z, bx, by, bz);
    const int prob = bz / g.splits, split = bz - prob * g.splits;
    const int KB = (g.K + PK_K - 1) / PK_K;
    const int RBA = (g.M + BM - 1) / BM, RBB = (g.N + BN - 1) / BN;  // packed row blocks of A and B
    const int kb0 = split * g.k_per_split / PK_K, kb1 = min(KB, (split + 1) * g.k_per_split / PK_K);
    const int nt = kb1 - kb0;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (nt > 0) {
        const char* Apk = reinterpret_cast<const char*>(g.Ag[prob]);
        const char* Bpk = reinterpret_cast<const char*>(g.Bg[prob]);
        // a block's second row block may lie beyond the matrix (odd number of row blocks): read the first one again
        const int la0 = 2 * by, la1 = min(2 * by + 1, RBA - 1), rb0 = 2 * bx, rb1 = min(2 * bx + 1, RBB - 1);
        const int ra0 = pk_a_rb(g, prob, la0), ra1 = pk_a_rb(g, prob, la1);
        const char* A0 = Apk + ((size_t)ra0 * KB + kb0) * PK_TILE;
        const char* A1 = Apk + ((size_t)ra1 * KB + kb0) * PK_TILE;
        const int KBb = g.b_kb_stride > 0 ? g.b_kb_stride : KB;
        const char* B0 = Bpk + ((size_t)rb0 * KBb + kb0) * PK_TILE;
        const char* B1 = Bpk + ((size_t)rb1 * KBb + kb0) * PK_TILE;
        // this wave's fragments: A rows = row block wm of the stage, B rows = 64 (wn & 1) .. of row block 2 + (wn >> 1)
        const int fa = wm * PK_TILE + pk_off(0, lane & 31, lane >> 5);
        const int fb = (2 + (wn >> 1)) * PK_TILE + pk_off(0, (wn & 1) * 64 + (lane & 31), lane >> 5);
        pkb_issue(A0, A1, B0, B1, pkbsm, wave, lane);
        {
            const size_t o1 = (size_t)(nt > 1 ? 1 : 0) * PK_TILE;
            pkb_issue(A0 + o1, A1 + o1, B0 + o1, B1 + o1, pkbsm + PKB_STAGE, wave, lane);
        }
        int cur = 0;
        for (int it = 0; it < nt; ++it) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile `it` (this wave's pieces; tile it + 1 may be in flight)
            __builtin_amdgcn_s_barrier();                      // ... everyone's; and everyone has read tile it - 1
            asm volatile("" ::: "memory");
            const char* sb = pkbsm + cur * PKB_STAGE;
            bf16x8 fbv[2][3], fav[4][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fbv[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fb + j * 1024 + pl * PK_PLANE));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fav[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fa + i * 1024 + pl * PK_PLANE));
            __builtin_amdgcn_sched_barrier(0);  // the 18 fragment reads go out first, in the order the MFMAs want them
            const int nx = it + 2 < nt ? it + 2 : nt - 1;
            const size_t on = (size_t)nx * PK_TILE;
            const int slot = cur == 0 ? 2 : cur - 1;  // (it + 2) % 3: the slot tile it - 1 has left
            pkb_issue(A0 + on, A1 + on, B0 + on, B1 + on, pkbsm + slot * PKB_STAGE, wave, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int o = 0; o < 6; ++o) {  // the six piece products, the smallest first
                    const int pa = o == 0 ? 2 : (o == 1 ? 0 : (o == 2 ? 1 : (o == 3 ? 0 : (o == 4 ? 1 : 0))));
                    const int pb = o == 0 ? 0 : (o == 1 ? 2 : (o == 2 ? 1 : (o == 3 ? 1 : (o == 4 ? 0 : 0))));
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fav[i][pa], fbv[j][pb], acc[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {  // one LDS-DMA instruction behind every eighth MFMA
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == 2 ? 0 : cur + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    gemm_epilogue<4, 2>(g, prob, bz, by * 256 + wm * 128, bx * 256 + wn * 64, lane, acc);
}

// The split-bf16 kernel on packed operands (see "split-bf16 GEMM on packed operands" above): one kernel for all four
// transpose forms -- the orientation went away in the pack kernels.
__global__ __launch_bounds__(256, 2) void gemm_pk_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float pksm[];
    int bx, by, bz;
    pk_tile_of_block(gridDim.x, gridDim.y, gridDim.z, bx, by, bz);
    gemm_block<false, false, 0>(g, pksm, bx, by, bz);
}

// ... and its XCD-filtered form (see GemmArgs::xcc_mask): one tile per block, drawn off the launch's counter.  168 registers
// per lane: a surplus block is admitted beside a persistent recurrence block (264 - 280) and leaves at once.
__global__ __launch_bounds__(256, 2) void gemm_pk_filtered_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float pksm[];
    if (!((g.xcc_mask >> gemm_xcc_id()) & 1u)) return;
    __shared__ int s_tile;
    const int total = g.grid_x * g.grid_y * g.grid_z;
    if (threadIdx.x == 0) s_tile = (int)atomicAdd(g.tile_counter, 1u);
    __syncthreads();
    const int tile = s_tile;
    if (tile >= total) return;
    const int bx = tile % g.grid_x, r = tile / g.grid_x;
    gemm_block<false, false, 0>(g, pksm, bx, r % g.grid_y, r / g.grid_y);
}

// The XCD-filtered launches of the f32-input kernel (see GemmArgs::xcc_mask) run this copy with ONE register stage (<= 232 registers per lane):
// a block that lands on an XCD where a persistent recurrence block (264 - 280 registers) holds every CU must still be
// ADMITTED there in order to leave -- 264 + 256 does not fit a SIMD's 512, so the two-stage kernel's surplus blocks (and
// with them the launch's completion, and everything queued behind it on the side stream) would wait for the recurrence
// to end.  (clang's amdgpu_num_vgpr attribute does not cap the allocation: checked.)
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f32_filtered_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE_F];
    if (!((g.xcc_mask >> gemm_xcc_id()) & 1u)) return;
    __shared__ int s_tile;
    const int total = g.grid_x * g.grid_y * g.grid_z;
    if (threadIdx.x == 0) s_tile = (int)atomicAdd(g.tile_counter, 1u);
    __syncthreads();
    const int tile = s_tile;
    if (tile >= total) return;
    const int bx = tile % g.grid_x, r = tile / g.grid_x;
    gemm_block<TA, TB, 1>(g, smem, bx, r % g.grid_y, r / g.grid_y);
}

// Queued behind every XCD-filtered launch: all `total` tiles were drawn iff the counter reached `total` (draws are
// sequential; blocks that drew a tile have finished it by the time this kernel runs on the same stream).
__global__ void gemm_filtered_check_kernel(const unsigned* __restrict__ tile_counter, unsigned total,
                                           unsigned* __restrict__ err_word) {
    if (__hip_atomic_load(tile_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) sa_raise(err_word, 4u);
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

// The same fold for the plain form (no row remap, beta = 0, no ReLU; N and ldc multiples of 4, C 16-byte aligned): four
// columns per thread, 16-byte accesses (round 5: the weight gradients' 88 MB fold ran at 3.2 TB/s on 4-byte ones).  The sums
// are formed in the same order: bit-identical.
__global__ __launch_bounds__(256) void gemm_splitk_reduce4_kernel(GemmArgs g, int splits) {
    const long idx4 = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)g.M * g.N, total4 = total >> 2;
    const int prob = blockIdx.y;
    if (idx4 >= total4) {  // the tail threads fold the fused column sums
        const long m = idx4 - total4;
        if (m < g.M && g.cs_partial && g.colsumg[prob]) {
            float t = 0.f;
            for (int z = 0; z < splits; ++z) t += g.cs_partial[((long)prob * splits + z) * g.M + m];
            g.colsumg[prob][m] = t;
        }
        return;
    }
    const long idx = idx4 << 2;
    const int row = (int)(idx / g.N), col = (int)(idx - (long)row * g.N);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < splits; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(g.partial + ((long)prob * splits + z) * total + idx);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;  // fixed order: deterministic
    }
    const float* gbias = g.biasg[prob];
    const float4 b = gbias ? make_float4(gbias[col], gbias[col + 1], gbias[col + 2], gbias[col + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(g.Cg[prob] + (long)row * g.ldc + col) =
        make_float4(g.alpha * s.x + b.x, g.alpha * s.y + b.y, g.alpha * s.z + b.z, g.alpha * s.w + b.w);
}

static bool reduce4_ok(const GemmArgs& g) {
    if (g.m_inner > 0 || g.beta != 0.f || g.relu || (g.N & 3) || (g.ldc & 3)) return false;
    for (int p = 0; p < g.nprob; ++p)
        if ((uintptr_t)g.Cg[p] & 15) return false;
    return ((uintptr_t)g.partial & 15) == 0;
}
static void launch_splitk_reduce(const GemmArgs& g, int splits, int nprob, hipStream_t stream) {
    if (reduce4_ok(g)) {
        const long total = ((long)g.M * g.N >> 2) + g.M;
        hipLaunchKernelGGL(gemm_splitk_reduce4_kernel, dim3((unsigned)((total + 255) / 256), nprob), dim3(256), 0, stream, g, splits);
    } else {
        const long total = (long)g.M * g.N + g.M;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256), nprob), dim3(256), 0, stream, g, splits);
    }
}

// Products big enough that two pack launches (and a pass over both operands) pay: >= 8 GFLOP per launch, a reduction of
// at least 256, and an output at least half a tile wide in both directions.  option gemm.exact = 1 switches the path off,
// gemm.exact = 0 forces it for every unfiltered product.
bool pk_worth_it(int M, int N, int K, int nprob) {
    const long e = sa_opt(SA_OPT_GEMM_EXACT);  // 1: never; 0: always (tests: small shapes through the packed path)
    if (e == 0 || e == 1) return e == 0;
    return 2.0 * M * N * K * nprob >= 8.0e9 && K >= 256 && M >= 64 && N >= 64;
}

int choose_splits(int M, int N, int K, int nprob = 1) {
    // Cost model in units of one k-tile of a block that shares its CU with a second block (two blocks per CU keep each
    // other's matrix pipe busy; measured ~3.9 us).  A launch of `blocks` blocks takes ceil(blocks / 512) rounds of
    // (k-tiles per block + ~6 tiles of prologue / epilogue); a block that has its CU to itself (<= 256 blocks) runs
    // its tiles ~0.6 x as long.  Splitting K adds the reduce launch: ~2 tiles of launch gap + its traffic at ~3 TB/s.
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * nprob;
    const int ktiles = (K + BK - 1) / BK;
    if (ktiles < 8) return 1;
    int max_s = ktiles / 4;
    if (max_s > 128) max_s = 128;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= max_s; ++s) {
        const long blocks = tiles * s;
        if (blocks > 4096 && s > 1) break;
        const double rounds = (double)((blocks + 511) / 512);
        const double per_tile = blocks <= 256 ? 0.6 : 1.0;
        double cost = rounds * ((double)((ktiles + s - 1) / s) + 6.0) * per_tile;
        if (s > 1) cost += 2.0 + (double)s * M * N * nprob * 4.0 / 3.0e12 / 3.9e-6;
        if (cost < best_cost * 0.97) { best_cost = cost; best = s; }  // prefer fewer splits unless clearly better
    }
    return best;
}

// Launch the packed split-bf16 kernel for GemmArgs whose Ag / Bg hold PACKED operands.
// 256 x 256 block tiles (half the LDS traffic per flop, one block per CU) when they fill the chip: no split-K and >= 85 %
// of whole rounds of 256 CUs; else 128 x 128 tiles, two blocks per CU (measured, tools/gemm_bench.py: 4096^3 191 vs 182
// TFLOP/s, d x of layer 0 132 vs 130; but the layer-0 projection 136 vs 156 and the weight gradients 89 vs 122).
// A split-K factor that lets the 256 x 256 kernel (one block per CU, half the L2 -> LDS bytes per flop) take a product whose
// OUTPUT is too small to fill the chip with 256-tiles -- the weight gradients: 6 x 2 x 7 = 84 tiles, 252 blocks at s = 3.
// 0: none (the output is large, the tiles would be padded by more than 10 %, or a split would be shorter than 128 k-tiles).
int big_tile_splits(int M, int N, int K, int nprob) {
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256) * nprob;
    const double waste = (double)((M + 255) / 256 * 256) * ((N + 255) / 256 * 256) / ((double)M * N);
    if (waste > 1.10 || t256 >= 200) return 0;
    const int ktiles = (K + PK_K - 1) / PK_K;
    for (int s = 2; s <= 16; ++s) {
        if (ktiles / s < 128) break;
        const long blocks = t256 * s;
        if ((double)blocks / (double)((blocks + 255) / 256 * 256) >= 0.90) return s;
    }
    return 0;
}

ctcStatus_t pk_launch(const GemmArgs& gp, int splits, hipStream_t stream, unsigned* err_word = nullptr, bool force_big = false) {
    static bool pk_attr_dev[48] = {false};
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 16) devid = 0;
    const int M = gp.M, N = gp.N, nprob = gp.nprob;
    if (gp.xcc_mask && gp.tile_counter) {  // XCD-filtered: block b lands on XCD b % 8 -- enough blocks that the allowed
                                           // XCDs alone receive one per tile
        if (!pk_attr_dev[devid + 32]) {
            if (hipFuncSetAttribute((const void*)gemm_pk_filtered_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    PK_NST * PK_STAGE) != hipSuccess)
                return CTC_STATUS_EXECUTION_FAILED;
            pk_attr_dev[devid + 32] = true;
        }
        int allowed = 0;
        for (int x = 0; x < 8; ++x) allowed += (gp.xcc_mask >> x) & 1u;
        const long tiles = (long)gp.grid_x * gp.grid_y * gp.grid_z;
        const dim3 fgrid((unsigned)((tiles + allowed - 1) / allowed * 8 + 8), 1, 1);
        hipLaunchKernelGGL(gemm_pk_filtered_kernel, fgrid, dim3(256), PK_NST * PK_STAGE, stream, gp);
        if (err_word)
            hipLaunchKernelGGL(gemm_filtered_check_kernel, dim3(1), dim3(1), 0, stream, (const unsigned*)gp.tile_counter,
                               (unsigned)tiles, err_word);
        return CTC_STATUS_SUCCESS;
    }
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256) * nprob;
    const bool big_tile = force_big || (splits == 1 && t256 >= 200 && (double)t256 / (double)((t256 + 255) / 256 * 256) >= 0.85);
    if (big_tile) {
        if (!pk_attr_dev[devid + 16]) {
            if (hipFuncSetAttribute((const void*)gemm_pk256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    PK_NST * PKB_STAGE) != hipSuccess)
                return CTC_STATUS_EXECUTION_FAILED;
            pk_attr_dev[devid + 16] = true;
        }
        const dim3 bgrid((N + 255) / 256, (M + 255) / 256, nprob * splits);
        hipLaunchKernelGGL(gemm_pk256_kernel, bgrid, dim3(512), PK_NST * PKB_STAGE, stream, gp);
    } else {
        if (!pk_attr_dev[devid]) {
            if (hipFuncSetAttribute((const void*)gemm_pk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    PK_NST * PK_STAGE) != hipSuccess)
                return CTC_STATUS_EXECUTION_FAILED;
            pk_attr_dev[devid] = true;
        }
        const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, nprob * splits);
        hipLaunchKernelGGL(gemm_pk_kernel, grid, dim3(256), PK_NST * PK_STAGE, stream, gp);
    }
    return CTC_STATUS_SUCCESS;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- thin products
// The classifier of the CTC model is (B T', H) x (H, |V| + 1) with 29 classes (ctc_model.py:19,29): forward, its input
// gradient and its weight gradient are products with ONE dimension of at most 32 -- a quarter of a 128-wide tile, 9 TFLOP/s
// on the tiled kernels (0.16 ms of the 8 ms step for 1.4 GFLOP).  They are bound by reading or writing the (B T', H)
// matrix once (32.6 MB at S-LIBRI: ~8 us), so each gets a kernel shaped for that: 16-row MFMA tiles (v_mfma_f32_16x16x4_f32,
// exact fp32 products as everywhere on this path), the thin operand held in LDS, the big one streamed once.
//   thin_nt_kernel  C[M, N<=32]  = A[M, K] B[N, K]^T + bias      (forward: logits)
//   thin_nn_kernel  C[M, N]      = A[M, K<=32] B[K, N]           (input gradient)
//   thin_tn_kernel  C[M<=32, N]  = A[K, M]^T B[K, N]             (weight gradient; K = B T' split over the grid, partial
//                                                                sums folded in a fixed order: deterministic)
typedef float tf32x4 __attribute__((ext_vector_type(4)));
constexpr int kThinSplit = 64;  // K chunks of thin_tn_kernel

// U: 16-byte loads of the streamed operand a lane keeps in flight (K / 16 is a multiple of U): the kernel is a latency chain
// of load trips, and four in flight left it at 19 us for 32.6 MB (round 5: eight at K = 512 -- 8 KB per wave)
template <int U>
__global__ __launch_bounds__(256) void thin_nt_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                      long ldb, const float* __restrict__ bias, float* __restrict__ C,
                                                      long ldc, int M, int N, int K, float beta) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];  // Bs[32][K + 4]
    const int pitch = K + 4;
    {   // the thin operand into LDS, eight 16-byte loads per thread in flight (unconditional loads on clamped indices: one
        // load per trip with its own wait was a chain of 16 L2 round trips, 6 of the kernel's 19 us)
        const int k4n = K / 4, total = 32 * k4n;
        for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = min(e0 + 256 * q, total - 1), n = e / k4n, k4 = e - n * k4n;
                v[q] = *reinterpret_cast<const float4*>(B + (long)min(n, N - 1) * ldb + 4 * k4);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + 256 * q, n = e / k4n, k4 = e - n * k4n;
                if (e < total) *reinterpret_cast<float4*>(tsm + n * pitch + 4 * k4) = n < N ? v[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int m0 = (blockIdx.x * 4 + wave) * 16;
    const float* a_row = A + (long)min(m0 + i, M - 1) * lda + 4 * g;
    const int nit = K / 16;
    float4 a[U];  // the first trip's rows are requested before the barrier: they travel while the block fills its LDS
#pragma unroll
    for (int q = 0; q < U; ++q) a[q] = *reinterpret_cast<const float4*>(a_row + 16 * q);
    __syncthreads();
    if (m0 >= M) return;
    const float* b0 = tsm + i * pitch + 4 * g;
    const float* b1 = tsm + (16 + i) * pitch + 4 * g;
    tf32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    for (int it0 = 0; it0 < nit; it0 += U) {
        float4 ac[U];
#pragma unroll
        for (int q = 0; q < U; ++q) ac[q] = a[q];
        const int itn = it0 + U < nit ? it0 + U : it0;  // the next trip's rows (the last trip re-reads its own: unused)
#pragma unroll
        for (int q = 0; q < U; ++q) a[q] = *reinterpret_cast<const float4*>(a_row + 16 * (itn + q));
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const float4 w0 = *reinterpret_cast<const float4*>(b0 + 16 * (it0 + q));
            const float4 w1 = *reinterpret_cast<const float4*>(b1 + 16 * (it0 + q));
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q].x, w0.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q].x, w1.x, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q].y, w0.y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q].y, w1.y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q].z, w0.z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q].z, w1.z, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q].w, w0.w, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q].w, w1.w, c1, 0, 0, 0);
        }
    }
    // C / D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * g + r;
        if (m >= M) continue;
        float* crow = C + (long)m * ldc;
        if (i < N) { float v = c0[r] + (bias ? bias[i] : 0.f); if (beta != 0.f) v += beta * crow[i]; crow[i] = v; }
        if (16 + i < N) { float v = c1[r] + (bias ? bias[16 + i] : 0.f); if (beta != 0.f) v += beta * crow[16 + i]; crow[16 + i] = v; }
    }
}

// A wave = 16 rows of C, every column.  The product is formed TRANSPOSED -- the thin operand's 16-column tile as the MFMA's
// row operand, the streamed rows as its column operand -- so that a lane ends up with four CONSECUTIVE columns of ONE row
// of C: one 16-byte store per tile (64 bytes contiguous per row and instruction) instead of four 4-byte ones.  BETA is a
// template parameter and whole row tiles take a loop without a branch: with a load of C (beta) or a branch around the stores
// inside the loop hipcc waits for vmcnt(0) every trip, i.e. for the previous tile's stores to be ACKNOWLEDGED -- 32 trips x
// ~1 us, the 33 us this kernel took for a 33 MB result in rounds 4-5 (ISA: s_waitcnt vmcnt(0) at the loop head).
template <bool BETA>
__global__ __launch_bounds__(256) void thin_nn_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                      long ldb, float* __restrict__ C, long ldc, int M, int N, int K,
                                                      float beta) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];  // Bt[N][36]: the thin operand transposed, k padded to 32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int m0 = (blockIdx.x * 4 + wave) * 16;
    const float* a_row = A + (long)min(m0 + i, M - 1) * lda;
    float a[2][4];  // unconditional loads on clamped indices, requested before the LDS fill
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * it + 4 * g + j;
            const float v = a_row[min(k, K - 1)];
            a[it][j] = k < K ? v : 0.f;
        }
    {   // the thin operand, transposed, into LDS: eight 16-byte loads per thread in flight (one 4-byte load per trip with
        // its own wait was a chain of 64 L2 round trips -- most of the 33 us this kernel took in rounds 4-5)
        const int n4n = N / 4, total = 32 * n4n;
        for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = min(e0 + 256 * q, total - 1), k = e / n4n, n4 = e - k * n4n;
                v[q] = *reinterpret_cast<const float4*>(B + (long)min(k, K - 1) * ldb + 4 * n4);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + 256 * q, k = e / n4n, n4 = e - k * n4n;
                if (e < total) {
                    const bool on = k < K;
                    float* t = tsm + (4 * n4) * 36 + k;
                    t[0] = on ? v[q].x : 0.f; t[36] = on ? v[q].y : 0.f; t[72] = on ? v[q].z : 0.f; t[108] = on ? v[q].w : 0.f;
                }
            }
        }
    }
    __syncthreads();
    if (m0 >= M) return;
    // lane (i, g): row m0 + i of C, columns 16 c + 4 g .. + 3 of tile c
    auto tile = [&](int c) {
        const float* bt = tsm + (16 * c + i) * 36 + 4 * g;
        const float4 w0 = *reinterpret_cast<const float4*>(bt), w1 = *reinterpret_cast<const float4*>(bt + 16);
        tf32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, a[0][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, a[0][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, a[0][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, a[0][3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, a[1][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, a[1][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, a[1][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, a[1][3], acc, 0, 0, 0);
        return acc;
    };
    const int nc = N / 16;
    if (m0 + 16 <= M) {
        float* crow = C + (long)(m0 + i) * ldc + 4 * g;
        int c = 0;
        for (; c + 4 <= nc; c += 4) {  // four independent MFMA chains per trip
            tf32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = tile(c + q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4* cp = reinterpret_cast<float4*>(crow + 16 * (c + q));
                float4 v = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
                if constexpr (BETA) { const float4 o = *cp; v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w; }
                *cp = v;
            }
        }
        for (; c < nc; ++c) {
            const tf32x4 acc = tile(c);
            float4* cp = reinterpret_cast<float4*>(crow + 16 * c);
            float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if constexpr (BETA) { const float4 o = *cp; v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w; }
            *cp = v;
        }
    } else {  // the last, ragged row tile of the matrix (one wave of the launch)
        const bool ok = m0 + i < M;
        float* crow = C + (long)min(m0 + i, M - 1) * ldc + 4 * g;
        for (int c = 0; c < nc; ++c) {
            const tf32x4 acc = tile(c);
            if (ok) {
                float4* cp = reinterpret_cast<float4*>(crow + 16 * c);
                float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
                if constexpr (BETA) { const float4 o = *cp; v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w; }
                *cp = v;
            }
        }
    }
}

// grid (ceil(N / 64), kThinSplit): a wave = a 64-column group of C, both 16-row tiles, a quarter of the block's K chunk.
// A lane reads 16 bytes of a row of B -- columns n0 + 4 i .. + 3, one for each of four MFMA column tiles (tile q <-> columns
// n0 + 4 i + q) -- so an instruction of the wave covers four rows of k x 256 contiguous bytes (round 4: 4-byte loads, 64
// bytes per row, one 16-column tile per wave, 15.7 us for the 33 MB operand), and the next trip's loads are in flight during
// a trip's 32 MFMAs.  Loads are unconditional on clamped indices; what lies beyond K, M or N is zeroed by selects.
// CS: the column sums of A (the bias gradient of the layer, db = dlogits^T 1: ctc_model.py:29) ride along as one more MFMA per
// k step against a column of ones, in the blocks of column group 0 -- two launches less than a separate column-sum pass.
template <bool CS>
__global__ __launch_bounds__(256) void thin_tn_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                      long ldb, float* __restrict__ part, float* __restrict__ cs_part, int M,
                                                      int N, int K) {
    __shared__ __attribute__((aligned(16))) float red[4][32][64];  // [wave][slot = 16 t + 4 r + q][lane]
    __shared__ float red_cs[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 64;
    const bool do_cs = CS && blockIdx.x == 0;
    const int per = (((K + kThinSplit - 1) / kThinSplit) + 63) / 64 * 64;  // rows of K per block: four waves x whole 16-k trips
    const int kb = blockIdx.y * per + wave * (per / 4);
    const int kend = min(K, kb + per / 4);
    tf32x4 c[2][4], cs[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        cs[t] = tf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) c[t][q] = tf32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool r0 = i < M, r1 = 16 + i < M, cn = n0 + 4 * i < N;  // (N is a multiple of 16: a lane's four columns are in or out)
    const float* pa0 = A + min(i, M - 1);
    const float* pa1 = A + min(16 + i, M - 1);
    const float* pb = B + min(n0 + 4 * i, N - 4);
    float a0[4], a1[4];
    float4 b[4];
    auto fetch = [&](int k0) {  // (a wave's four lane groups read four different k: a 16-k slab per trip)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + g + 4 * j;
            const bool ok = k < kend;
            const long kc = ok ? k : (K - 1);
            const float v0 = pa0[kc * lda], v1 = pa1[kc * lda];
            const float4 vb = *reinterpret_cast<const float4*>(pb + kc * ldb);
            // masked by a bit-AND, not a select: hipcc sinks a load whose value only one arm of a select uses under that
            // arm's condition -- a branch around every load and a wait behind each (the clamped rows are rows of the matrix).
            // (Round 5 masked by a PRODUCT with 0 / 1: a non-finite value in the clamped row then put 0 * inf = NaN into sums
            // that should have received exactly 0 -- ADVICE r05; the AND gives +0 whatever the bits.)
            const unsigned m0 = ok && r0 ? ~0u : 0u, m1 = ok && r1 ? ~0u : 0u, mb = ok && cn ? ~0u : 0u;
            auto keep = [](float v, unsigned m) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m); };
            a0[j] = keep(v0, m0);
            a1[j] = keep(v1, m1);
            b[j] = make_float4(keep(vb.x, mb), keep(vb.y, mb), keep(vb.z, mb), keep(vb.w, mb));
        }
    };
    if (kb < kend) fetch(kb);
    for (int k0 = kb; k0 < kend; k0 += 16) {
        float x0[4], x1[4];
        float4 y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { x0[j] = a0[j]; x1[j] = a1[j]; y[j] = b[j]; }
        fetch(k0 + 16 < kend ? k0 + 16 : k0);  // the next trip's slab (the last trip re-reads its own: unused)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], y[j].x, c[0][0], 0, 0, 0);
            c[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], y[j].x, c[1][0], 0, 0, 0);
            c[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], y[j].y, c[0][1], 0, 0, 0);
            c[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], y[j].y, c[1][1], 0, 0, 0);
            c[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], y[j].z, c[0][2], 0, 0, 0);
            c[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], y[j].z, c[1][2], 0, 0, 0);
            c[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], y[j].w, c[0][3], 0, 0, 0);
            c[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], y[j].w, c[1][3], 0, 0, 0);
            if (do_cs) {
                cs[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], 1.f, cs[0], 0, 0, 0);
                cs[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], 1.f, cs[1], 0, 0, 0);
            }
        }
    }
    // C / D layout: lane (i, g), register r <-> row 16 t + 4 g + r, column n0 + 4 i + q of tile q
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave][16 * t + 4 * r + q][lane] = c[t][q][r];
    if (do_cs && i == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red_cs[wave][16 * t + 4 * g + r] = cs[t][r];
    }
    __syncthreads();
    // part[split][32][N]: thread (wave w, lane) folds slots 8 w .. 8 w + 7 = row tile t = w / 2, registers r = 2 (w % 2) + {0, 1},
    // the four column tiles: two 16-byte stores of four consecutive columns each; the four waves' sums in a fixed order
    {
        const int t = wave >> 1;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * (wave & 1) + rr, slot = 16 * t + 4 * r;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = (red[0][slot + q][lane] + red[1][slot + q][lane]) + (red[2][slot + q][lane] + red[3][slot + q][lane]);
            const int row = 16 * t + 4 * g + r;
            if (cn) *reinterpret_cast<float4*>(part + ((long)blockIdx.y * 32 + row) * N + n0 + 4 * i) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (do_cs && threadIdx.x < 32)
        cs_part[blockIdx.y * 32 + threadIdx.x] = (red_cs[0][threadIdx.x] + red_cs[1][threadIdx.x]) + (red_cs[2][threadIdx.x] + red_cs[3][threadIdx.x]);
}

// C = the sum of the kThinSplit partial products (in a fixed order: deterministic); elements [M N, M N + M) of the grid: the
// column sums of A likewise (cs_part != null)
__global__ __launch_bounds__(256) void thin_tn_fold_kernel(const float* __restrict__ part, const float* __restrict__ cs_part,
                                                           float* __restrict__ C, long ldc, float* __restrict__ colsum, int M,
                                                           int N, float beta) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long mn = (long)M * N;
    if (e >= mn) {
        const int m = (int)(e - mn);
        if (!cs_part || m >= M) return;
        float v = 0.f;
        for (int sblk = 0; sblk < kThinSplit; ++sblk) v += cs_part[sblk * 32 + m];
        colsum[m] = v;
        return;
    }
    const int row = (int)(e / N), col = (int)(e - (long)row * N);
    float v = 0.f;
    for (int s0 = 0; s0 < kThinSplit; s0 += 8) {  // eight partials in flight, added in order
        float w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = part[((long)(s0 + q) * 32 + row) * N + col];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += w[q];
    }
    float* cp = C + (long)row * ldc + col;
    *cp = beta != 0.f ? v + beta * *cp : v;
}

// which thin kernel (0 = none) a single plain product takes; a function of the shape and the strides only
int thin_kind(int trans_a, int trans_b, int M, int N, int K, long lda, long ldb) {
    if (sa_opt(SA_OPT_GEMM_THIN) == 0) return 0;
    if (!trans_a && trans_b && N <= 32 && M >= 2048 && K >= 16 && K <= 1024 && (K % 16) == 0 && (lda & 3) == 0 && (ldb & 3) == 0)
        return 1;
    if (!trans_a && !trans_b && K <= 32 && M >= 2048 && N >= 16 && N <= 1024 && (N % 16) == 0) return 2;
    if (trans_a && !trans_b && M <= 32 && K >= 2048 && N >= 16 && (N % 16) == 0) return 3;
    return 0;
}
size_t thin_workspace_bytes(int trans_a, int trans_b, int M, int N, int K) {
    // (the strides of a contiguous operand; a caller with other strides that misses the kernel simply takes the tiled path)
    return thin_kind(trans_a, trans_b, M, N, K, 4, 4) == 3 ? (size_t)kThinSplit * 32 * (N + 1) * sizeof(float) : 0;  // (+ column sums)
}

ctcStatus_t sa_gemm_f32_group_impl(int nprob, int trans_a, int trans_b, int M, int N, int K, float alpha,
                                   const float* const* A, long lda, const float* const* B, long ldb, float beta,
                                   float* const* C, long ldc, const float* const* bias, const SaGemmEpilogue* ep,
                                   void* workspace, size_t workspace_bytes, hipStream_t stream,
                                   const SaGemmOpts* opts) {
    SA_CLEAR_ERR();
    if (opts && opts->colsum && (!trans_a || alpha != 1.f)) return CTC_STATUS_INVALID_VALUE;
    if (M < 0 || N < 0 || K < 0 || nprob < 1 || nprob > kMaxGroup) return CTC_STATUS_INVALID_VALUE;
    if (M == 0 || N == 0) return CTC_STATUS_SUCCESS;
    if (nprob == 1 && alpha == 1.f && !ep && !(opts && (opts->xcc_mask || opts->drop)) && A[0] && B[0] && C[0] &&
        (((uintptr_t)A[0] | (uintptr_t)B[0]) & 15) == 0) {
        int kind = thin_kind(trans_a, trans_b, M, N, K, lda, ldb);
        if (opts && opts->colsum && (kind != 3 || beta != 0.f || !opts->colsum[0])) kind = 0;  // column sums: the weight-gradient form only
        const float* bs = bias ? bias[0] : nullptr;
        if (kind == 1) {
            const size_t smem = (size_t)32 * (K + 4) * sizeof(float);
            const int nit = K / 16;
            auto nt_fn = (nit % 8) == 0 ? thin_nt_kernel<8> : (nit % 4) == 0 ? thin_nt_kernel<4> : (nit % 2) == 0 ? thin_nt_kernel<2>
                                                                                                              : thin_nt_kernel<1>;
            if (smem <= 48 * 1024 || hipFuncSetAttribute((const void*)nt_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         (int)smem) == hipSuccess) {
                hipLaunchKernelGGL(nt_fn, dim3((M + 63) / 64), dim3(256), smem, stream, A[0], lda, B[0], ldb, bs, C[0],
                                   ldc, M, N, K, beta);
                SA_CHECK_LAUNCH();
                return CTC_STATUS_SUCCESS;
            }
            (void)hipGetLastError();
        } else if (kind == 2 && !bs && ((ldc | ldb) & 3) == 0 && ((uintptr_t)C[0] & 15) == 0) {  // (16-byte accesses)
            const size_t smem = (size_t)N * 36 * sizeof(float);
            auto nn_fn = beta != 0.f ? thin_nn_kernel<true> : thin_nn_kernel<false>;
            if (smem <= 48 * 1024 || hipFuncSetAttribute((const void*)nn_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         (int)smem) == hipSuccess) {
                hipLaunchKernelGGL(nn_fn, dim3((M + 63) / 64), dim3(256), smem, stream, A[0], lda, B[0], ldb, C[0], ldc,
                                   M, N, K, beta);
                SA_CHECK_LAUNCH();
                return CTC_STATUS_SUCCESS;
            }
            (void)hipGetLastError();
        } else if (kind == 3 && !bs && (ldb & 3) == 0 && workspace &&
                   workspace_bytes >= (size_t)kThinSplit * 32 * (N + 1) * sizeof(float)) {
            float* part = (float*)workspace;
            float* cs_part = part + (size_t)kThinSplit * 32 * N;
            float* cs_out = opts && opts->colsum ? opts->colsum[0] : nullptr;  // (beta = 0 semantics: written, not accumulated)
            const dim3 tgrid((N + 63) / 64, kThinSplit);
            if (cs_out) hipLaunchKernelGGL(thin_tn_kernel<true>, tgrid, dim3(256), 0, stream, A[0], lda, B[0], ldb, part, cs_part, M, N, K);
            else hipLaunchKernelGGL(thin_tn_kernel<false>, tgrid, dim3(256), 0, stream, A[0], lda, B[0], ldb, part, cs_part, M, N, K);
            hipLaunchKernelGGL(thin_tn_fold_kernel, dim3((unsigned)(((long)M * N + (cs_out ? M : 0) + 255) / 256)), dim3(256), 0,
                               stream, (const float*)part, cs_out ? (const float*)cs_part : nullptr, C[0], ldc, cs_out, M, N, beta);
            SA_CHECK_LAUNCH();
            return CTC_STATUS_SUCCESS;
        }
    }
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
    // The split-bf16 kernel on packed operands for every product that is worth two pack launches, when the caller's
    // workspace has room for the packed copies; option gemm.exact = 1: the f32-input MFMA kernel everywhere.
    bool use_pk = !(opts && opts->exact) && pk_worth_it(M, N, K, nprob);
    const size_t pkA = sa_align_up(pk_bytes(M, K), 256), pkB = sa_align_up(pk_bytes(N, K), 256);
    const int pk_kt = 8, pk_parts = 2 * ((((K + PK_K - 1) / PK_K) + pk_kt - 1) / pk_kt);
    const int Mpad = (M + BM - 1) / BM * BM;
    const size_t pk_cs = (opts && opts->colsum) ? sa_align_up((size_t)nprob * pk_parts * Mpad * sizeof(float), 256) : 0;
    const size_t pk_need = (size_t)nprob * (pkA + pkB) + pk_cs;
    // Which arithmetic a product runs in is a function of its shape (pk_worth_it; sa_gemm_is_split_bf16 tells a caller),
    // never of the buffer it was handed: a split-bf16 product without room for its packed operands is an error.
    if (use_pk && (!workspace || workspace_bytes < pk_need)) return CTC_STATUS_INVALID_VALUE;
    char* pk_base = (char*)workspace;
    if (use_pk) {  // the packed copies sit in front of the split-K partials
        workspace = (char*)workspace + pk_need;
        workspace_bytes -= pk_need;
    }
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
    g.xcc_mask = 0; g.tile_counter = nullptr;
    g.a_rb_split = 1 << 30; g.a_rb_jump = 0; g.a_jump_probs = 0u; g.b_kb_stride = 0;
    g.drop = sa_drop_make(0.f, 0ull); g.drop_stream = 0u; g.drop_base = nullptr;
    g.grid_x = (int)grid.x; g.grid_y = (int)grid.y; g.grid_z = (int)grid.z;
    if (opts && opts->xcc_mask && opts->tile_counter) {
        g.xcc_mask = opts->xcc_mask; g.tile_counter = opts->tile_counter;
        int allowed = 0;
        for (int x = 0; x < 8; ++x) allowed += (opts->xcc_mask >> x) & 1u;
        const long tiles = (long)grid.x * grid.y * grid.z;
        // block b lands on XCD b % 8: launch enough blocks that the allowed XCDs alone receive `tiles` of them
        grid = dim3((unsigned)((tiles + allowed - 1) / allowed * 8 + 8), 1, 1);
    }
    const size_t dyn = 0;
    if (g.xcc_mask && !use_pk) {
        if (trans_a) {
            if (trans_b) hipLaunchKernelGGL((gemm_f32_filtered_kernel<true, true>), grid, dim3(256), 0, stream, g);
            else hipLaunchKernelGGL((gemm_f32_filtered_kernel<true, false>), grid, dim3(256), 0, stream, g);
        } else {
            if (trans_b) hipLaunchKernelGGL((gemm_f32_filtered_kernel<false, true>), grid, dim3(256), 0, stream, g);
            else hipLaunchKernelGGL((gemm_f32_filtered_kernel<false, false>), grid, dim3(256), 0, stream, g);
        }
        if (opts->err_word)
            hipLaunchKernelGGL(gemm_filtered_check_kernel, dim3(1), dim3(1), 0, stream, (const unsigned*)g.tile_counter,
                               (unsigned)(g.grid_x * g.grid_y * g.grid_z), opts->err_word);
    } else if (use_pk) {
        // pack A (logical [M][K]) and B (logical [N][K]) of every problem, then ONE kernel whatever the transposes were
        const int KB = (K + PK_K - 1) / PK_K;
        PackArgs pa;
        pa.K = K; pa.KB = KB; pa.kb_stride = KB; pa.kb_lead = 0; pa.zero_lead = 0; pa.kt = pk_kt; pa.cs_part = nullptr; pa.Rpad = Mpad;
        for (int p = 0; p < kMaxGroup; ++p) pa.src_hi[p] = nullptr;
        for (int side = 0; side < 2; ++side) {
            const bool kcontig = side == 0 ? !trans_a : (trans_b != 0);
            const int R = side == 0 ? M : N;
            for (int p = 0; p < nprob; ++p) pa.src[p] = side == 0 ? A[p] : B[p];
            pa.dst = pk_base + (side == 0 ? 0 : (size_t)nprob * pkA);
            pa.dst_stride = side == 0 ? pkA : pkB;
            pa.ld = side == 0 ? lda : ldb;
            pa.R = R; pa.R_lo = R;
            pa.vec = side == 0 ? g.vecA : g.vecB;
            pa.cs_part = (side == 0 && pk_cs) ? (float*)(pk_base + (size_t)nprob * (pkA + pkB)) : nullptr;
            const dim3 pgrid((KB + pk_kt - 1) / pk_kt, (R + BM - 1) / BM, nprob);
            if (kcontig) hipLaunchKernelGGL(pk_pack_kcontig_kernel, pgrid, dim3(256), 0, stream, pa);
            else hipLaunchKernelGGL(pk_pack_mcontig_kernel, pgrid, dim3(256), 0, stream, pa);
        }
        if (pk_cs)
            hipLaunchKernelGGL(pk_colsum_fold_kernel, dim3((M + 15) / 16, nprob), dim3(256), 0, stream,
                               (const float*)(pk_base + (size_t)nprob * (pkA + pkB)), pk_parts, Mpad, M, g);
        GemmArgs gp = g;
        for (int p = 0; p < nprob; ++p) {
            gp.Ag[p] = (const float*)(pk_base + (size_t)p * pkA);
            gp.Bg[p] = (const float*)(pk_base + (size_t)nprob * pkA + (size_t)p * pkB);
            gp.colsumg[p] = nullptr;  // done by the pack kernel
        }
        const ctcStatus_t pst = pk_launch(gp, splits, stream, opts ? opts->err_word : nullptr);
        if (pst != CTC_STATUS_SUCCESS) return pst;
        g = gp;  // the split-K reduce below must not fold column sums either
    } else if (trans_a) {
        if (trans_b) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), dyn, stream, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), dyn, stream, g);
    } else {
        if (trans_b) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), dyn, stream, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), dyn, stream, g);
    }
    SA_CHECK_LAUNCH();
    if (splits > 1) {
        launch_splitk_reduce(g, splits, nprob, stream);
        SA_CHECK_LAUNCH();
    }
    return CTC_STATUS_SUCCESS;
}

size_t sa_gemm_group_workspace_bytes(int nprob, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || nprob <= 0) return 0;
    const int s = choose_splits(M, N, K, nprob);
    size_t w = s > 1 ? (size_t)nprob * s * ((size_t)M * N + M) * sizeof(float) : 0;
    if (nprob == 1) {  // the weight gradient of a thin layer (thin_tn_kernel): its partial sums; the query does not know the
                       // transpose form, so it covers it whenever the shape could be that product
        const size_t t = thin_workspace_bytes(1, 0, M, N, K);
        if (t > w) w = t;
    }
    if (pk_worth_it(M, N, K, nprob)) {  // packed split-bf16 copies of both operands (+ the row-sum partials)
        const int parts = 2 * ((((K + PK_K - 1) / PK_K) + 7) / 8);
        w += (size_t)nprob * (sa_align_up(pk_bytes(M, K), 256) + sa_align_up(pk_bytes(N, K), 256)) +
             sa_align_up((size_t)nprob * parts * ((M + BM - 1) / BM * BM) * sizeof(float), 256);
    }
    return w;
}

ctcStatus_t sa_gemm_f32_impl(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, long lda,
                             const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
                             const SaGemmEpilogue* ep, void* workspace, size_t workspace_bytes,
                             hipStream_t stream) {
    return sa_gemm_f32_group_impl(1, trans_a, trans_b, M, N, K, alpha, &A, lda, &B, ldb, beta, &C, ldc, &bias, ep,
                                  workspace, workspace_bytes, stream, nullptr);
}

extern "C" size_t sa_gemm_workspace_bytes(int M, int N, int K) { return sa_gemm_group_workspace_bytes(1, M, N, K); }
extern "C" int sa_gemm_is_split_bf16(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && pk_worth_it(M, N, K, 1) ? 1 : 0; }

extern "C" ctcStatus_t sa_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A,
                                   long lda, const float* B, long ldb, float beta, float* C, long ldc,
                                   const float* bias, void* workspace, size_t workspace_bytes, void* stream) {
    SA_CLEAR_ERR();
    return sa_gemm_f32_impl(trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, nullptr, workspace,
                            workspace_bytes, (hipStream_t)stream);
}


// C (M, N) = A^T B for A stored (K, M), B (K, N), and colsum (M) = the column sums of A, in one call: the weight and bias
// gradients of a linear layer from the gradient of its output (A = d out (rows, classes), B = the layer's input): what
// autograd derives for LinearND (/root/reference/speech/models/model.py:118-133; CTC.loss -> loss.backward(), train.py:30).
// Thin A (M <= 32): thin_tn_kernel<CS> forms both in one pass over the operands; other shapes: the tiled kernel with its
// column-sum epilogue.  Workspace: sa_gemm_workspace_bytes(M, N, K).
extern "C" ctcStatus_t sa_gemm_tn_colsum_f32(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C,
                                             long ldc, float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
    SA_CLEAR_ERR();
    if (!A || !B || !C || !colsum) return CTC_STATUS_INVALID_VALUE;
    SaGemmOpts o;
    o.no_split = 0; o.colsum = &colsum; o.xcc_mask = 0; o.tile_counter = nullptr;
    const float* bias = nullptr;
    return sa_gemm_f32_group_impl(1, 1, 0, M, N, K, 1.f, &A, lda, &B, ldb, 0.f, &C, ldc, &bias, nullptr, workspace, workspace_bytes,
                                  (hipStream_t)stream, &o);
}

// ---- packed operands as a caller-visible (library-internal) object: internal.h ----------------------------------------
size_t sa_pk_operand_bytes(int R, int K) { return sa_align_up(pk_bytes(R, K), 256); }
int sa_pk_rowsum_parts(int K) { return 2 * ((((K + PK_K - 1) / PK_K) + 7) / 8); }
bool sa_pk_enabled(int M, int N, int K, int nprob) { return pk_worth_it(M, N, K, nprob); }

ctcStatus_t sa_pk_pack(int nprob, const float* const* src, const float* const* src_hi, int R_lo, long ld, int R, int K,
                       int kcontig, char* dst, size_t dst_stride, float* cs_part, hipStream_t stream, int kb_stride, int kb_lead,
                       int zero_lead) {
    if (nprob < 1 || nprob > kMaxGroup || !dst || R <= 0 || K <= 0) return CTC_STATUS_INVALID_VALUE;
    PackArgs pa;
    const int KB = (K + PK_K - 1) / PK_K;
    pa.K = K; pa.KB = KB; pa.kt = 8; pa.cs_part = cs_part; pa.Rpad = (R + BM - 1) / BM * BM;
    pa.kb_stride = kb_stride > 0 ? kb_stride : KB;
    pa.kb_lead = kb_lead; pa.zero_lead = zero_lead;
    if (kb_lead < 0 || (kb_lead > 0 && (kcontig || pa.kb_stride < KB + kb_lead))) return CTC_STATUS_INVALID_VALUE;
    pa.dst = dst; pa.dst_stride = dst_stride; pa.ld = ld; pa.R = R; pa.R_lo = src_hi ? R_lo : R;
    pa.vec = (ld & 3) == 0;
    for (int p = 0; p < kMaxGroup; ++p) { pa.src[p] = nullptr; pa.src_hi[p] = nullptr; }
    for (int p = 0; p < nprob; ++p) {
        if (!src[p]) return CTC_STATUS_INVALID_VALUE;
        pa.src[p] = src[p];
        pa.src_hi[p] = src_hi ? src_hi[p] : nullptr;
        pa.vec = pa.vec && (((uintptr_t)src[p] & 15) == 0);
    }
    const dim3 pgrid((KB + pa.kt - 1) / pa.kt, (R + BM - 1) / BM, nprob);
    if (kcontig) {
        if (cs_part || src_hi) return CTC_STATUS_INVALID_VALUE;
        hipLaunchKernelGGL(pk_pack_kcontig_kernel, pgrid, dim3(256), 0, stream, pa);
    } else {
        hipLaunchKernelGGL(pk_pack_mcontig_kernel, pgrid, dim3(256), 0, stream, pa);
    }
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

namespace {
// out[p][m] = (beta ? beta * out : 0) + sum over the parts of row (m < split ? m : m + jump), in a fixed order
// blockIdx.z selects one of up to two row maps (split, jump) with its own outputs: both bias families of a layer stack
// (db_ih: rows [0, 3H) of the 4H-row gate operand, db_hh: rows [0, 2H) + [3H, 4H)) in one launch
struct RowsumOut { float* out[2][kMaxGroup]; int split[2], jump[2]; };
__global__ __launch_bounds__(256) void pk_rowsum_fold_kernel(const float* __restrict__ part, int nparts, int Rpad, int M,
                                                             RowsumOut o, float beta) {
    const int m = blockIdx.x * 16 + (threadIdx.x >> 4), j = threadIdx.x & 15, prob = blockIdx.y, z = blockIdx.z;
    const int r = m < o.split[z] ? m : m + o.jump[z];
    float t = 0.f;
    if (m < M)
        for (int q = j; q < nparts; q += 16) t += part[((size_t)prob * nparts + q) * Rpad + r];
#pragma unroll
    for (int sft = 8; sft > 0; sft >>= 1) t += __shfl_xor(t, sft, 64);
    if (m >= M || j != 0 || !o.out[z][prob]) return;
    float* dstp = o.out[z][prob] + m;
    *dstp = beta != 0.f ? beta * *dstp + t : t;
}
}  // namespace

ctcStatus_t sa_pk_rowsum_fold(int nprob, const float* cs_part, int nparts, int Rpad, int M, int split, int jump,
                              float* const* out, float beta, hipStream_t stream, int split2, int jump2, float* const* out2) {
    if (nprob < 1 || nprob > kMaxGroup || !cs_part || !out) return CTC_STATUS_INVALID_VALUE;
    RowsumOut o;
    for (int p = 0; p < kMaxGroup; ++p) { o.out[0][p] = p < nprob ? out[p] : nullptr; o.out[1][p] = p < nprob && out2 ? out2[p] : nullptr; }
    o.split[0] = split; o.jump[0] = jump; o.split[1] = split2; o.jump[1] = jump2;
    hipLaunchKernelGGL(pk_rowsum_fold_kernel, dim3((M + 15) / 16, nprob, out2 ? 2 : 1), dim3(256), 0, stream, cs_part, nparts,
                       Rpad, M, o, beta);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

size_t sa_gemm_pk_group_workspace_bytes(int nprob, int M, int N, int K) {
    int s = choose_splits(M, N, K, nprob);
    const int sb = big_tile_splits(M, N, K, nprob);  // the 256-tile kernel's own split factor (sa_gemm_pk_group)
    if (sb > s) s = sb;
    return s > 1 ? (size_t)nprob * s * ((size_t)M * N + M) * sizeof(float) : 0;
}

// C[p] (M x N) = (beta C[p]) + A[p] B[p]^T on operands that are ALREADY packed (sa_pk_pack): A[p] rows [0, a_split) and
// [a_split + a_jump, ...) of its packed operand when bit p of a_jump_probs is set (row counts in multiples of 128), else
// rows [0, M); B[p] rows [0, N).
ctcStatus_t sa_gemm_pk_group(int nprob, int M, int N, int K, const char* const* Apk, int a_split, int a_jump,
                             unsigned a_jump_probs, const char* const* Bpk, float beta, float* const* C, long ldc, void* workspace,
                             size_t workspace_bytes, hipStream_t stream, const SaGemmOpts* opts) {
    SA_CLEAR_ERR();
    if (nprob < 1 || nprob > kMaxGroup || M <= 0 || N <= 0 || K <= 0 || (a_split % BM) || (a_jump % BM))
        return CTC_STATUS_INVALID_VALUE;
    GemmArgs g;
    g.nprob = nprob; g.vecA = g.vecB = 1;
    for (int p = 0; p < kMaxGroup; ++p) { g.Ag[p] = g.Bg[p] = nullptr; g.Cg[p] = nullptr; g.biasg[p] = nullptr; g.colsumg[p] = nullptr; }
    for (int p = 0; p < nprob; ++p) {
        if (!Apk[p] || !Bpk[p] || !C[p]) return CTC_STATUS_INVALID_VALUE;
        g.Ag[p] = (const float*)Apk[p]; g.Bg[p] = (const float*)Bpk[p]; g.Cg[p] = C[p];
    }
    g.lda = g.ldb = 0; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.f; g.beta = beta;
    g.m_inner = 0; g.m_mid = 1; g.s_outer = g.s_mid = 0; g.col_stride = 1; g.relu = 0;
    int splits = choose_splits(M, N, K, nprob);
    g.drop = sa_drop_make(0.f, 0ull); g.drop_stream = 0u; g.drop_base = nullptr;
    if (opts && opts->drop && opts->drop->on()) {
        g.drop = *opts->drop; g.drop_stream = opts->drop_stream; g.drop_base = opts->drop_base;
        splits = 1;  // the mask goes on in the epilogue that writes the result
    }
    bool force_big = false;
    // (r6) grouped weight gradients 780 -> 726 us at S-LIBRI (224 -> 241 TF): 7 x 48 tiles x 3 splits of the 128-tile kernel, two
    // blocks per CU, against 84 x 3 = 252 blocks of the 256-tile kernel, one per CU
    if (!(opts && opts->drop && opts->drop->on()) && !(opts && opts->xcc_mask)) {
        const int sb = big_tile_splits(M, N, K, nprob);
        if (sb > 1 && workspace && workspace_bytes >= (size_t)nprob * sb * ((size_t)M * N + M) * sizeof(float)) { splits = sb; force_big = true; }
    }
    if (splits > 1 && (!workspace || workspace_bytes < (size_t)nprob * splits * ((size_t)M * N + M) * sizeof(float))) splits = 1;
    int kps = (K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    if (kps < BK) kps = BK;
    splits = (K + kps - 1) / kps;
    g.k_per_split = kps; g.splits = splits;
    g.partial = splits > 1 ? (float*)workspace : nullptr;
    g.cs_partial = nullptr;
    g.xcc_mask = 0; g.tile_counter = nullptr;
    g.a_rb_split = a_split / BM; g.a_rb_jump = a_jump / BM; g.a_jump_probs = a_jump ? a_jump_probs : 0u;
    g.b_kb_stride = opts ? opts->b_kb_stride : 0;
    g.grid_x = (N + BN - 1) / BN; g.grid_y = (M + BM - 1) / BM; g.grid_z = nprob * splits;
    if (opts && opts->xcc_mask && opts->tile_counter) { g.xcc_mask = opts->xcc_mask; g.tile_counter = opts->tile_counter; }
    const ctcStatus_t st = pk_launch(g, splits, stream, opts ? opts->err_word : nullptr, force_big && splits > 1);
    if (st != CTC_STATUS_SUCCESS) return st;
    SA_CHECK_LAUNCH();
    if (splits > 1) {
        launch_splitk_reduce(g, splits, nprob, stream);
        SA_CHECK_LAUNCH();
    }
    return CTC_STATUS_SUCCESS;
}
