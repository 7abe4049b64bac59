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
};

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

// TA: A is stored (K, M) (m-contiguous);  !TA: A is stored (M, K) (k-contiguous)
// TB: B is stored (N, K) (k-contiguous);  !TB: B is stored (K, N) (n-contiguous)
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
    const bool do_colsum = TA && g.colsumg[prob] != nullptr && bx == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DEPTH == 2) {
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
                    g.partial[((long)bz * g.M + row) * g.N + col] = acc[i][j][r];
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

__device__ __forceinline__ int gemm_xcc_id() {  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE_F];
    gemm_block<TA, TB>(g, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The XCD-filtered launches (see GemmArgs::xcc_mask) run this copy with ONE register stage (<= 232 registers per lane):
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
    if (__hip_atomic_load(tile_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) atomicOr(err_word, 4u);
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
    g.xcc_mask = 0; g.tile_counter = nullptr;
    g.grid_x = (int)grid.x; g.grid_y = (int)grid.y; g.grid_z = (int)grid.z;
    if (opts && opts->xcc_mask && opts->tile_counter) {
        g.xcc_mask = opts->xcc_mask; g.tile_counter = opts->tile_counter;
        int allowed = 0;
        for (int x = 0; x < 8; ++x) allowed += (opts->xcc_mask >> x) & 1u;
        const long tiles = (long)grid.x * grid.y * grid.z;
        // block b lands on XCD b % 8: launch enough blocks that the allowed XCDs alone receive `tiles` of them
        grid = dim3((unsigned)((tiles + allowed - 1) / allowed * 8 + 8), 1, 1);
    }
    // "polite" launches (opts->pad_lds): dynamic LDS on top of the kernel's 66 KB so that a CU admits ONE block of this
    // launch -- what a side-stream GEMM wants while a persistent recurrence kernel holds every CU (gru.hip): the
    // recurrence blocks then always find room, before and after this launch's blocks arrive.
    size_t dyn = 0;
    if (opts && opts->pad_lds) {
        static bool attr_set_dev[16] = {false};  // function attributes are per device
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 16) devid = 0;
        bool& attr_set = attr_set_dev[devid];
        dyn = 81 * 1024 - sizeof(float) * 2 * 2 * TILE_F;
        if (!attr_set) {
            const void* fns[4] = {(const void*)gemm_f32_kernel<true, true>, (const void*)gemm_f32_kernel<true, false>,
                                  (const void*)gemm_f32_kernel<false, true>, (const void*)gemm_f32_kernel<false, false>};
            for (int i = 0; i < 4; ++i)
                if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess)
                    return CTC_STATUS_EXECUTION_FAILED;
            attr_set = true;
        }
    }
    if (g.xcc_mask) {
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
    } else if (trans_a) {
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
