// gru.hip -- the GRU recurrence (forward and backward through time) for gfx950.
//
// Reference: nn.GRU inside Model.encode, /root/reference/speech/models/model.py:35-39,73 (cuDNN persistent RNN on
// the reference's GPU path).  Equations: SURVEY.md Appendix B, gate row order [r; z; n], b_hn inside r * (.).
// The input projections x W_ih^T + b_ih are one big MFMA GEMM outside (gemm_f32.hip); this file owns what is
// sequential: per time step  h_t = cell(ai_t, h_{t-1} W_hh^T + b_hh)  and its reverse-time gradient.
//
// One launch per time step ("job list" kernel: blockIdx.z selects one of up to 8 independent (layer, direction, t)
// jobs, so both directions of a bidirectional layer -- or a diagonal of a layer stack -- share a launch).
// The step is a dependent-latency chain (operand fetch from other CUs' output -> MFMA -> LDS reduce -> gate math), not
// a throughput problem, so sa_gru_stack_fwd/bwd run a unidirectional stack as a CHUNKED LAYER WAVEFRONT: layer l works
// on time chunk c while layer l+1 works on chunk c-1, one launch carrying up to L layer-steps (L x fewer serial
// launches, all 256 CUs busy); the input projections of a chunk are a plain row-range GEMM because activations are
// kept time-major (T, B, .) inside the stack.
// A block owns a 16 (batch) x 16 (hidden units) output tile: the small-M product is computed on
// v_mfma_f32_16x16x4_f32 with the K dimension split over the block's 4 waves (one per SIMD), operands loaded
// straight from L2 into MFMA fragments with 16-byte loads (k-contiguous rows on both sides, so no LDS staging),
// partial tiles reduced through LDS, and the gate math fused into the epilogue.
//   forward : 3 gate tiles (r, z, n rows of W_hh) x K = H
//   backward: 1 tile of dh_{t-1} = dah_t W_hh (rows of W_hh^T, transposed once per call) x K = 3H, then the gate
//             gradients of step t-1 in the epilogue (they become the next launch's A operand).
#include <type_traits>
#include <stdlib.h>

#include <mutex>

#include "common.h"
#include "dropout.h"
#include "internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxJobs = 8;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// acc[n] += A_tile(16 x kslice) * B_tile_n(16 x kslice)^T over this wave's K slice.
// 16x16x4 f32 operand layout: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]; each lane loads
// 4 consecutive k (one float4) from its row and feeds 4 MFMAs, the four 16-lane groups cover 16 consecutive k.
// The loop is a dependent-latency problem (L2 round trip ~0.5 us per trip if loads and MFMAs alternate), so operands
// are fetched U iterations at a time -- U * (1 + NT) independent 16-byte loads in flight -- before their MFMAs issue.
// DUAL (U even): k-groups alternate between acc and acc2 (the caller adds them): one accumulator is a chain of
// dependent 16x16x4 MFMAs (40-cycle dependent latency against a 32-cycle issue interval), two run at the issue rate.
template <int NT, int U, bool DUAL = false>
__device__ __forceinline__ void smallm_mfma(const float* __restrict__ a_row, const float* const (&b_row)[NT],
                                            int kbeg, int kend, int K, f32x4 (&acc)[NT], int g,
                                            f32x4* acc2 = nullptr) {
    for (int kk0 = kbeg; kk0 < kend; kk0 += 16 * U) {  // wave-uniform trip counts (MFMA must not sit in divergent flow)
        float4 a[U];
        float4 b[U][NT];
#pragma unroll
        for (int it = 0; it < U; ++it) {
            const int k = kk0 + 16 * it + 4 * g;
            a[it] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NT; ++n) b[it][n] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk0 + 16 * it < kend) {
                if (k + 3 < K) {
                    a[it] = *reinterpret_cast<const float4*>(a_row + k);
#pragma unroll
                    for (int n = 0; n < NT; ++n) b[it][n] = *reinterpret_cast<const float4*>(b_row[n] + k);
                } else {
                    if (k + 0 < K) { a[it].x = a_row[k];
#pragma unroll
                        for (int n = 0; n < NT; ++n) b[it][n].x = b_row[n][k]; }
                    if (k + 1 < K) { a[it].y = a_row[k + 1];
#pragma unroll
                        for (int n = 0; n < NT; ++n) b[it][n].y = b_row[n][k + 1]; }
                    if (k + 2 < K) { a[it].z = a_row[k + 2];
#pragma unroll
                        for (int n = 0; n < NT; ++n) b[it][n].z = b_row[n][k + 2]; }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < U; ++it) {
            if (kk0 + 16 * it < kend) {  // wave-uniform
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4& d = (DUAL && (it & 1)) ? acc2[n] : acc[n];
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, b[it][n].x, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, b[it][n].y, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, b[it][n].z, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, b[it][n].w, d, 0, 0, 0);
                }
            }
        }
    }
}

// Row-structured arrays (ai, stash, dai, dah) address row (b, t) as  b * rb + t * rt  (time-major: rb = 1, rt = B;
// batch-major: rb = T, rt = 1).
// ------------------------------------------------------------------------------------------------------ forward step
struct FwdJob {
    const float* ai;     // rows x 3H
    const float* w_hh;   // (3H, H)
    const float* b_hh;   // (3H)
    float* h_out;        // [b * hs_b + t * hs_t + j]
    float* stash;        // rows x 5H or null: r, z, n, q, h_{t-1}
    long hs_b, hs_t;
    int t, t_prev;       // t_prev < 0: first step (h_{t-1} = 0)
};
struct FwdJobs {
    int n, B, H;
    int bt0;        // first batch tile handled by this launch (blockIdx.y is relative to it)
    int tile_rows;  // batch rows owned by a block: 16, or 8 / 4 when the batch is cut into more concurrent chains
    unsigned long long* stamp;  // profiling: block (0,0,0) writes wall_clock64() at entry / exit (null: off)
    long rb, rt;
    FwdJob j[kMaxJobs];
};

__global__ __launch_bounds__(256) void gru_fwd_step_kernel(FwdJobs P) {
    __shared__ float red[4][3][256];
    const FwdJob& J = P.j[blockIdx.z];
    const int H = P.H, B = P.B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int u0 = blockIdx.x * 16, b0 = (blockIdx.y + P.bt0) * P.tile_rows;
    const bool first = J.t_prev < 0;
    const bool stamper = P.stamp && threadIdx.x == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0;
    if (stamper) P.stamp[0] = wall_clock64();

    // epilogue operands first: their latency overlaps the operand fetch + MFMA phase
    const int bi = threadIdx.x >> 4, uj = threadIdx.x & 15;
    const int b = b0 + bi, u = u0 + uj;
    const bool live = b < B && u < H && bi < P.tile_rows;
    float e_ai_r = 0.f, e_ai_z = 0.f, e_ai_n = 0.f, e_br = 0.f, e_bz = 0.f, e_bn = 0.f, hp = 0.f;
    const long row = live ? (long)b * P.rb + (long)J.t * P.rt : 0;
    if (live) {
        const float* ai = J.ai + row * 3 * H;
        e_ai_r = ai[u]; e_ai_z = ai[H + u]; e_ai_n = ai[2 * H + u];
        e_br = J.b_hh[u]; e_bz = J.b_hh[H + u]; e_bn = J.b_hh[2 * H + u];
        if (!first) hp = J.h_out[(long)b * J.hs_b + (long)J.t_prev * J.hs_t + u];
    }

    if (!first) {
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
        smallm_mfma<3, 8>(a_row, b_rows, kbeg, kend, H, acc, g);
        // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][n][(g * 4 + r) * 16 + i] = acc[n][r];
    }
    __syncthreads();
    if (!live) { if (stamper) P.stamp[1] = wall_clock64(); return; }
    float sr = 0.f, sz = 0.f, sn = 0.f;
    if (!first) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sr += red[w][0][threadIdx.x];
            sz += red[w][1][threadIdx.x];
            sn += red[w][2][threadIdx.x];
        }
    }
    const float r = sigmoidf_(e_ai_r + sr + e_br);
    const float z = sigmoidf_(e_ai_z + sz + e_bz);
    const float q = sn + e_bn;
    const float n = tanhf(e_ai_n + r * q);
    const float h = (1.0f - z) * n + z * hp;
    J.h_out[(long)b * J.hs_b + (long)J.t * J.hs_t + u] = h;
    if (J.stash) {
        float* s = J.stash + row * 5 * H;
        s[u] = r; s[H + u] = z; s[2 * H + u] = n; s[3 * H + u] = q; s[4 * H + u] = hp;
    }
    if (stamper) P.stamp[1] = wall_clock64();
}

// -------------------------------------------------------------------------------------- persistent forward chunk
// One launch runs `nsteps` consecutive time steps of up to 8 layer-jobs.  Block (unit tile u, batch tile bt, job)
// keeps its 3 x 16 rows of W_hh in LDS for the whole chunk (the step kernels re-stream them from a cold L2 on every
// launch) and carries its own h values in registers.  What must cross CUs every step is the h_{t} slice each block
// produces: it is published with write-through (sc1) stores, then one relaxed agent-scope counter per
// (job, batch tile) group is bumped; consumers poll that counter from one lane and read h_{t-1} with sc1
// (L1-bypassing) 16-byte buffer loads -- the R1 hand-off recipe of the CDNA guide, placement independent.
// All blocks must be co-resident: grid <= #CUs and ~110 KB of LDS force one block per CU; spins are bounded.
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct PFwdJob {
    const float* ai;
    const float* w_hh;
    const float* b_hh;
    float* h_out;
    float* stash;
    unsigned* counters;  // one per batch tile; monotonic over the whole stack call
    long hs_b, hs_t;
    int t0, nsteps;      // first time index of this launch, number of steps
    int dt, t_first;     // +1 / -1 (reverse direction of a bidirectional layer); the sequence's very first time index
    unsigned base;       // arrivals per counter before this launch
    char* hx;            // gru_fwd_chunk_planes_kernel: the job's bf16-planes exchange buffer (see gru_fwd_planes_kernel), or null
};
struct PFwdJobs {
    int n, B, H;
    long rb, rt;
    // XCD-local mode (SA_GRU_PERSIST=2): the launch is always 8 groups x 32 workgroups; a workgroup takes its group
    // from the XCC id it actually runs on and its unit tile from an arrival counter, so a sync group never spans
    // XCDs whatever the dispatcher did (a group short of members times out into `err`, it cannot hang or corrupt).
    int xcd_mode, nbt, ntile_u;
    // batches wider than the XCD groups can host are walked in PASSES of `nbt` batch tiles: this launch handles tiles
    // [bt0, bt0 + nbt) of nbt_all (the recurrences of different batch rows are independent)
    int bt0, nbt_all;
    // flag-less hand-off (XCD-local mode only): h_out is pre-filled with a NaN sentinel by the host; a consumer wave
    // simply re-reads the rows it needs (sc1 loads, served by the XCD's L2) until no element is the sentinel any more.
    // No arrival counter, no store acknowledgement wait, no workgroup barrier before the MFMAs: a wave starts as soon
    // as ITS k-slice's producers have published.
    int flagless;
    unsigned* reg;       // [8] arrival counters, monotonic over the stack call
    unsigned reg_base;   // arrivals per counter before this launch
    unsigned long long* stamp;  // profiling: the (0, 0, 0) workgroup writes wall_clock64() at entry / exit (null: off)
    unsigned* err;       // the library's STICKY error word (PersistHealth::dev): failures are OR-ed in, never cleared here
    int prio;            // raise the waves' issue priority (they share their CU with side-stream GEMM blocks)
    int spin_limit;      // polls before a hand-off counts as failed (SA_GRU_SPIN_LIMIT; default 1 << 20)
    int fault;           // fault injection (SA_GRU_FAULT=1, tests only): unit tile 1 of group 0 leaves before its first step
    unsigned long long* timing;  // debug (SA_GRU_TIMING=1): per block 4 phase accumulators in 10 ns ticks, else null
    float* dump;         // gru_fwd_chunk_kernel: 256 x 256 floats nobody reads (stores of rows beyond the batch)
    PFwdJob j[kMaxJobs];
};

constexpr unsigned kSentinel = 0x7fc0deadu;  // a quiet NaN no finite state or gradient can equal
constexpr int kXRing = 4;  // time slots of the backward kernels' exchange ring (a power of two; see PBwdJobs::packed)
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
// (r6) the whole vector is cast, THEN its elements are compared: __builtin_bit_cast(unsigned, v.y) on an element of an
// ext_vector_type compiled to a compare of element 0 (hipcc 7.2: all four terms read v.x, and a 16-byte load feeding only this
// test was narrowed to a dword) -- rounds 1-5 tested the first dword of every 16 bytes only
__device__ __forceinline__ bool has_sentinel(f32x4v v) {
    const u32x4v q = __builtin_bit_cast(u32x4v, v);
    return q.x == kSentinel || q.y == kSentinel || q.z == kSentinel || q.w == kSentinel;
}

// One XCD-local persistent workgroup per CU, by construction: the empty asm clobbers v255 and a7, which pins the
// kernel's register allocation at >= 264 of the 512 registers a SIMD lane owns -- two such workgroups can never share a
// CU, whatever the compiler's own count.  (Round 1 got the same exclusivity from an 84 KB LDS request; registers leave
// the LDS free, so that a side-stream GEMM block -- 152 registers, 81 KB of LDS -- FITS beside the recurrence block and
// uses the matrix-pipe cycles the latency-bound recurrence leaves idle.)  The recurrence waves run at raised priority:
// whenever they are ready to issue they go first.
#define SA_PERSIST_EXCLUSIVE(prio_)                        \
    do {                                                   \
        asm volatile("" ::: "v255", "a7");                 \
        if (prio_) __builtin_amdgcn_s_setprio(3);          \
    } while (0)

__device__ __forceinline__ int xcc_id() {  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
}

template <bool WREG>  // W_hh fragments resident in registers (always true: the LDS-weights variant served the retired
                      // chip-wide groups)
__global__ __launch_bounds__(256) void gru_fwd_persist_kernel(PFwdJobs P) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    if (WREG) SA_PERSIST_EXCLUSIVE(P.prio);
    int role_x = blockIdx.x, role_y = blockIdx.y, role_z = blockIdx.z;
    if (P.xcd_mode) {
        __shared__ int s_role[2];
        if (threadIdx.x == 0) {
            const int x = xcc_id();
            s_role[0] = x;
            s_role[1] = (int)(__hip_atomic_fetch_add(P.reg + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                              P.reg_base);
        }
        __syncthreads();
        if (s_role[1] < 0 || s_role[1] >= 32) {  // more than 32 workgroups landed on this XCD
            if (threadIdx.x == 0) sa_raise(P.err, 2u);
            return;
        }
        // an XCD hosts 32 / ntile_u groups (narrower layers: 16 or 8 unit tiles per group)
        // (H / 16 not a divisor of 32 -- H = 192, 320, 384, 448: the CUs beyond an XCD's last whole group idle)
        const int sub = s_role[1] / P.ntile_u, g = s_role[0] * (32 / P.ntile_u) + sub;
        if (sub >= 32 / P.ntile_u) return;
        role_x = s_role[1] - sub * P.ntile_u;
        role_z = g / P.nbt;
        role_y = g - role_z * P.nbt + P.bt0;
        if (role_z >= P.n) return;  // idle group: fill / drain of the layer wavefront, or fewer groups than slots
        if ((P.fault & 1) && role_x == 1 && role_y == 0 && role_z == 0) return;  // injected fault: a group one member short
    }
    const PFwdJob& J = P.j[role_z];
    const int H = P.H, B = P.B;
    const bool stamper = P.xcd_mode && P.stamp && threadIdx.x == 0 && role_x + role_y + role_z == 0;
    if (stamper) P.stamp[0] = wall_clock64();
    const int LDW = H + 4;                       // padded row pitch: 16-byte aligned, spreads the fragment reads
    float* Wl = psm;                             // [48][LDW]: rows u0.., H+u0.., 2H+u0.. of W_hh (not with WREG)
    float* red = psm + (WREG ? 0 : 48 * LDW);    // [2][4][3][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ntile_u = P.xcd_mode ? P.ntile_u : (int)gridDim.x;
    const int u0 = role_x * 16, b0 = role_y * 16;

    // W_hh slice -> LDS, once per launch
    for (int idx = tid; !WREG && idx < 48 * (H / 4); idx += 256) {
        const int r = idx / (H / 4), c4 = idx - r * (H / 4);
        const int grow = (r >> 4) * H + u0 + (r & 15);
        *reinterpret_cast<float4*>(&Wl[r * LDW + 4 * c4]) =
            *reinterpret_cast<const float4*>(J.w_hh + (long)grow * H + 4 * c4);
    }
    const int bi = tid >> 4, uj = tid & 15;
    const int b = b0 + bi, u = u0 + uj;
    const bool live = b < B;
    const float e_br = J.b_hh[u], e_bz = J.b_hh[H + u], e_bn = J.b_hh[2 * H + u];
    float hp = 0.f;
    if (live && J.t0 != J.t_first) hp = J.h_out[(long)b * J.hs_b + (long)(J.t0 - J.dt) * J.hs_t + u];
    unsigned* counter = J.counters + role_y;
    const int kslice = H / 4, kbeg = wave * kslice;  // H % 64 == 0 (checked by the host)
    const int brow = min(b0 + i, B - 1);
    __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)J.h_out, 0, 0x7fffffff, 0x00020000);
    bool dead = false;
    int budget = P.spin_limit;
    // XCD-local mode, H <= 512: the W_hh fragments this lane feeds to its MFMAs (3 gates x up to 8 k-groups x 4 floats)
    // are the same every step -- resident in registers instead of re-read from LDS
    float4 wr[8][3];
    if (WREG) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int k = kbeg + 16 * it + 4 * g;
#pragma unroll
            for (int n = 0; n < 3; ++n)
                wr[it][n] = 16 * it < kslice ? *reinterpret_cast<const float4*>(J.w_hh + (long)(n * H + u0 + i) * H + k)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = 0;
    const bool timed = P.timing != nullptr && tid == 0;
    if (timed) tprev = wall_clock64();
#define SA_TICK(k) if (timed) { const unsigned long long now = wall_clock64(); tacc[k] += now - tprev; tprev = now; }

    // A step's input projection (from HBM) is requested during the PREVIOUS step, after that step's stores: vmcnt is one
    // in-order queue, so requested at the top of its own step it would hold the poll's L2-resident rows back until it has
    // returned (the backward kernel's gather went 3.0 -> 1.5 us per step on the same change).  Unconditional
    // loads on a clamped row: a branch around them would cost a vmcnt(0) at the join.
    float e_ai_r = 0.f, e_ai_z = 0.f, e_ai_n = 0.f;
    auto fetch_ai = [&](int tt) {
        const float* ai = J.ai + ((long)(live ? b : 0) * P.rb + (long)tt * P.rt) * 3 * H;
        e_ai_r = ai[u]; e_ai_z = ai[H + u]; e_ai_n = ai[2 * H + u];
    };
    fetch_ai(J.t0);
    for (int s = 0; s < J.nsteps; ++s) {
        const int t = J.t0 + s * J.dt;
        const bool has_prev = t != J.t_first;
        const long row = (long)(live ? b : 0) * P.rb + (long)t * P.rt;
        if (s > 0 && !P.flagless) {  // every unit tile of this (job, batch tile) must have published h_{t-1}
            if (tid == 0 && !dead) {
                const unsigned need = J.base + (unsigned)ntile_u * (unsigned)s;
                int spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4 * P.spin_limit) { dead = true; sa_raise(P.err, 1u); break; }
                }
            }
            __syncthreads();
        }
        SA_TICK(0)
        if (has_prev) {
            f32x4 acc[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int abase = (int)(((long)brow * J.hs_b + (long)(t - J.dt) * J.hs_t) * 4);  // byte offset of the h row
            for (int kk0 = kbeg; kk0 < kbeg + kslice; kk0 += 128) {
                f32x4v a[8];
                for (int spins = 0;; ++spins) {
                    bool stale = false;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int k = kk0 + 16 * it + 4 * g;
                        a[it] = kk0 + 16 * it < kbeg + kslice
                                    ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hres, abase + 4 * k, 0, 16))
                                    : f32x4v{0.f, 0.f, 0.f, 0.f};
                    }
                    if (!P.flagless) break;
#pragma unroll
                    for (int it = 0; it < 8; ++it) stale |= has_sentinel(a[it]);
                    if (__builtin_amdgcn_ballot_w64(stale) == 0) break;  // wave-uniform: the MFMAs below stay convergent
                    // budget: a wave-uniform scalar; after one timeout it is 0, so a lost call drains quickly
                    if (spins > budget) { if (lane == 0) sa_raise(P.err, 1u); budget = 0; break; }
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    if (kk0 + 16 * it < kbeg + kslice) {
                        const int k = kk0 + 16 * it + 4 * g;
#pragma unroll
                        for (int n = 0; n < 3; ++n) {
                            const float4 w = WREG ? wr[it][n] : *reinterpret_cast<const float4*>(&Wl[(n * 16 + i) * LDW + k]);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, w.x, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, w.y, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, w.z, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, w.w, acc[n], 0, 0, 0);
                        }
                    }
                }
            }
            float* rd = red + (P.flagless ? (s & 1) * 3072 : 0);  // flag-less: double-buffered, one barrier per step
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) rd[(wave * 3 + n) * 256 + (g * 4 + r) * 16 + i] = acc[n][r];
        }
        __syncthreads();
        SA_TICK(1)
        float sr = 0.f, sz = 0.f, sn = 0.f;
        if (has_prev) {
            const float* rd = red + (P.flagless ? (s & 1) * 3072 : 0);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                sr += rd[(w * 3 + 0) * 256 + tid];
                sz += rd[(w * 3 + 1) * 256 + tid];
                sn += rd[(w * 3 + 2) * 256 + tid];
            }
        }
        if (live) {
            const float r = sigmoidf_(e_ai_r + sr + e_br);
            const float z = sigmoidf_(e_ai_z + sz + e_bz);
            const float q = sn + e_bn;
            const float n = tanhf(e_ai_n + r * q);
            const float h = (1.0f - z) * n + z * hp;
            __hip_atomic_store(J.h_out + (long)b * J.hs_b + (long)t * J.hs_t + u, h, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);  // write-through: visible to every CU once acknowledged
            if (J.stash) {
                float* st = J.stash + row * 5 * H;
                st[u] = r; st[H + u] = z; st[2 * H + u] = n; st[3 * H + u] = q; st[4 * H + u] = hp;
            }
            hp = h;
        }
        fetch_ai(s + 1 < J.nsteps ? t + J.dt : t);  // (the last step re-reads its own row: unused)
        SA_TICK(2)
        if (P.flagless) continue;  // the data is its own flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the flag
        __syncthreads();
        if (tid == 0) {
            // XCD-local groups: the RMW executes in the XCD's own L2 (workgroup scope = no sc1) and the pollers' sc1
            // loads are served by that same L2 -- 1.3 us per barrier instead of a memory round trip
            // (tools/ubench/xcd_exchange.hip).  NB a workgroup-scope LOAD would keep hitting a stale L1 line.
            if (P.xcd_mode) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SA_TICK(3)
    }
    if (timed) {
        unsigned long long* o = P.timing + 4 * ((role_z * (P.xcd_mode ? P.nbt : (int)gridDim.y) + role_y - P.bt0) * ntile_u + role_x);
        for (int k = 0; k < 4; ++k) atomicAdd(&o[k], tacc[k]);
    }
    if (P.stamp && threadIdx.x == 0) atomicMax(P.stamp + 1, (unsigned long long)wall_clock64());  // the LAST block out
#undef SA_TICK
}

// ------------------------------------------------------------------ persistent forward chunk, flag-less, round 5 form
// gru_fwd_persist_kernel<true> with its flag-less XCD-local hand-off, rebuilt like gru_fwd_fused_kernel (see there): the
// width is a template parameter and no branch surrounds a vector-memory instruction inside the loop -- the run-time-width
// kernel waits vmcnt(0) in front of every polling trip (for the step's HBM operands and for the acknowledgement of its own
// stores) and in front of every group of twelve MFMAs.  Serves the two directions of a bidirectional layer (one launch for
// all T steps, or chunk by chunk beside the side-stream projections) and the chunked layer wavefront (option gru.fused = 0).
// Bit-identical to gru_fwd_persist_kernel (same products in the same order).
template <int IPG, bool STASH>
__global__ __launch_bounds__(256) void gru_fwd_chunk_kernel(PFwdJobs P) {
    constexpr int H = 64 * IPG, NTU = H / 16, KS = 16 * IPG;
    extern __shared__ __attribute__((aligned(16))) float psm[];
    __shared__ int s_role[2];
    SA_PERSIST_EXCLUSIVE(P.prio);
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        s_role[0] = x;
        s_role[1] = (int)(__hip_atomic_fetch_add(P.reg + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.reg_base);
    }
    __syncthreads();
    if (s_role[1] < 0 || s_role[1] >= 32) {  // more than 32 workgroups landed on this XCD
        if (threadIdx.x == 0) sa_raise(P.err, 2u);
        return;
    }
    const int sub = s_role[1] / NTU, grp = s_role[0] * (32 / NTU) + sub;
    if (sub >= 32 / NTU) return;
    const int role_x = s_role[1] - sub * NTU, role_z = grp / P.nbt, role_y = grp - role_z * P.nbt + P.bt0;
    if (role_z >= P.n) return;  // idle group: fill / drain of the layer wavefront, or fewer groups than slots
    if ((P.fault & 1) && role_x == 1 && role_y == 0 && role_z == 0) return;  // injected fault: a group one member short
    const PFwdJob& J = P.j[role_z];
    const int B = P.B;
    const bool stamper = P.stamp && threadIdx.x == 0 && role_x + role_y + role_z == 0;
    if (stamper) P.stamp[0] = wall_clock64();
    float* red = psm;  // [2][4 waves][3 sums][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int u0 = role_x * 16, b0 = role_y * 16;
    const int bi = tid >> 4, uj = tid & 15;
    const int b = b0 + bi, u = u0 + uj;
    const bool live = b < B;
    const float e_br = J.b_hh[u], e_bz = J.b_hh[H + u], e_bn = J.b_hh[2 * H + u];
    const int kbeg = wave * KS;
    const int brow = min(b0 + i, B - 1);
    int budget = P.spin_limit;  // wave-uniform; 0 after the first timeout: the call is lost, drain quickly
    unsigned* errp = P.err;
    const int t0 = J.t0, nsteps = J.nsteps, dt = J.dt, t_first = J.t_first;
    __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)J.h_out, 0, 0x7fffffff, 0x00020000);
    const long hrow = ((long)brow * J.hs_b + kbeg + 4 * g) * 4;  // byte offset of this lane's first fragment in row brow, t = 0
    const long hstep = J.hs_t * 4;
    // per-thread operand / store addresses: base + t * stride (a row beyond the batch loads row 0 and stores to its dump slot)
    const float* p_ai = J.ai + (long)(live ? b : 0) * P.rb * 3 * H + u;
    const long s_ai = (long)P.rt * 3 * H;
    float* dump = P.dump + blockIdx.x * 256 + tid;
    float* p_h = live ? J.h_out + (long)b * J.hs_b + u : dump;
    const long s_h = live ? J.hs_t : 0;
    float* p_st = (STASH && live) ? J.stash + (long)b * P.rb * 5 * H + u : dump;
    const long s_st = (STASH && live) ? (long)P.rt * 5 * H : 0;
    const int so = (STASH && live) ? H : 0;
    float hp = 0.f;
    if (live && t0 != t_first) hp = J.h_out[(long)b * J.hs_b + (long)(t0 - dt) * J.hs_t + u];
    float4 wh[IPG][3];  // the W_hh fragments this lane feeds to its MFMAs: resident in registers
#pragma unroll
    for (int it = 0; it < IPG; ++it)
#pragma unroll
        for (int n = 0; n < 3; ++n)
            wh[it][n] = *reinterpret_cast<const float4*>(J.w_hh + (long)(n * H + u0 + i) * H + kbeg + 16 * it + 4 * g);
    __syncthreads();

    f32x4v a[IPG];
    auto poll_until_fresh = [&](int tprev) {  // flag-less hand-off inside the XCD: the row block of time tprev
        const int abase = (int)(hrow + (long)tprev * hstep);
        for (int spins = 0;; ++spins) {
            asm volatile("" ::: "memory");  // every trip re-issues its loads
#pragma unroll
            for (int it = 0; it < IPG; ++it)
                a[it] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hres, abase + 64 * it, 0, 16));
            bool stale = false;
#pragma unroll
            for (int it = 0; it < IPG; ++it) stale |= has_sentinel(a[it]);
            if (__builtin_amdgcn_ballot_w64(stale) == 0) break;
            if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
        }
    };
    float nx_r, nx_z, nx_n;  // the input projection of the step about to be processed, requested a step ago
    auto fetch_ai = [&](int tt) { const float* q = p_ai + (long)tt * s_ai; nx_r = q[0]; nx_z = q[H]; nx_n = q[2 * H]; };
    auto finish = [&](int s, int t, const f32x4 (&acc)[3], float e_r, float e_z, float e_n) {
        float* rd = red + (s & 1) * 3072;
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) rd[(wave * 3 + n) * 256 + (g * 4 + r) * 16 + i] = acc[n][r];
        __syncthreads();
        float sr = 0.f, sz = 0.f, sn = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sr += rd[(w * 3 + 0) * 256 + tid];
            sz += rd[(w * 3 + 1) * 256 + tid];
            sn += rd[(w * 3 + 2) * 256 + tid];
        }
        const float r = sigmoidf_(e_r + sr + e_br);
        const float z = sigmoidf_(e_z + sz + e_bz);
        const float q = sn + e_bn;
        const float n = tanhf(e_n + r * q);
        const float h = (1.0f - z) * n + z * hp;
        __hip_atomic_store(p_h + (long)t * s_h, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
        if (STASH) {
            float* st = p_st + (long)t * s_st;
            st[0] = r; st[so] = z; st[2 * so] = n; st[3 * so] = q; st[4 * so] = hp;
        }
        hp = h;
    };
    fetch_ai(t0);
    int s = 0;
    if (t0 == t_first && nsteps > 0) {  // the sequence's first step (peeled: no recurrent term, nobody to wait for)
        f32x4 acc[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float e_r = nx_r, e_z = nx_z, e_n = nx_n;
        fetch_ai(nsteps > 1 ? t0 + dt : t0);
        finish(0, t0, acc, e_r, e_z, e_n);
        s = 1;
    }
    for (; s < nsteps; ++s) {
        const int t = t0 + s * dt;
        f32x4 acc[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float e_r = nx_r, e_z = nx_z, e_n = nx_n;
        // Pacing: the first polling trip goes out once this step's own stores are acknowledged -- about when the other
        // blocks' rows land too (they published at the same moment).  Issued straight behind the stores the trip comes back
        // stale and is repeated: three trips of 1 MB per group and step instead of one, in L2 bandwidth that the stores
        // compete for -- 5.0 instead of 3.7 us per step on a bidirectional S-LIBRI layer (the run-time-width kernel was paced
        // by accident: hipcc's vmcnt(0) in front of its polling loads).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        poll_until_fresh(t - dt);
        // the next step's operands (HBM): requested behind the poll -- in front of it they hold the poll's data back (one
        // in-order queue), at the end of the step they hold the NEXT poll's back
        fetch_ai(s + 1 < nsteps ? t + dt : t);  // (the last step re-reads its own row: unused)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < IPG; ++it)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const float4 w = wh[it][n];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, w.x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, w.y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, w.z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, w.w, acc[n], 0, 0, 0);
            }
        finish(s, t, acc, e_r, e_z, e_n);
    }
    if (P.stamp && threadIdx.x == 0) atomicMax(P.stamp + 1, (unsigned long long)wall_clock64());  // the LAST block out
}

// ------------------------------------------------------------------------------------- fused forward layer wavefront
// ONE launch runs the whole unidirectional stack over all T steps: the (layer, batch tile) groups sit on their own XCDs
// as in the persistent chunk kernel, but a layer's input projection W_ih h_{l-1}[t] is computed inside the kernel too,
// so the layers pipeline at STEP granularity (no chunks, no per-chunk GEMM launches, no fill / drain of whole chunks).
//   per step and block:  (a) input phase   (l >= 1) wait until layer l-1 has published step t, read its h rows ONCE
//                            (sc1 loads: the rows were written through to memory by another XCD and are not in this
//                            XCD's L2 before this first read), MFMA against the W_ih rows streamed from L2;
//                        (b) recurrent phase: flag-less poll of h_l[t-1] inside the XCD, MFMA against W_hh in LDS;
//                        (c) reduce, gates, publish h_l[t] with write-through stores.
//   (a) of step t needs nothing of the block's own recurrence, so it fills the store -> L2 -> load hop of (b).
// Cross-XCD readiness is an arrival counter per (layer, batch tile) in memory (agent-scope RMW, fire and forget): a
// block reports step t-1 at step t, after its own stores of step t-1 have been acknowledged (a wait that is already
// satisfied) and its step-t barrier -- the consumer layer therefore trails by one step more than it strictly must.
struct PFusedFwd {
    int L, B, H, T, nbt, ntile_u;
    int bt0, nbt_all;            // see PFwdJobs
    long rb, rt;
    const float* ai0;            // (T*B, 3H): layer 0's input projection incl. bias (one GEMM before the launch)
    const float* w_ih[kMaxJobs]; // l >= 1: (3H, H)
    const float* b_ih[kMaxJobs];
    const float* w_hh[kMaxJobs];
    const float* b_hh[kMaxJobs];
    float* h_out[kMaxJobs];      // (T, B, H), pre-filled with the sentinel
    float* h_drop[kMaxJobs];     // inter-layer dropout (nn.GRU(dropout=p), model.py:38): layer l < L-1 also writes
                                 // h_out[l] * mask here and layer l+1 reads THAT; null without dropout
    SaDrop drop;                 // mask of layer l's output: stream drop_stream0 + l, index (t B + b) H + u (dropout.h)
    unsigned drop_stream0;
    float* stash[kMaxJobs];      // rows x 5H or null
    unsigned* prog;              // [L][nbt] arrival counters, zeroed by the host
    unsigned* reg;
    unsigned reg_base;
    unsigned* err;
    int spin_limit, fault, prio; // see PFwdJobs
    unsigned long long* stamp;
    unsigned long long* timing;  // debug (SA_GRU_TIMING=1): per block 4 phase accumulators in 10 ns ticks, else null
    float* dump;                 // 256 x 256 floats nobody reads: where the stores of rows beyond the batch go
    int report_every;            // a layer's blocks report progress to the layer above every so many steps (power of two)
    // gru_fwd_planes_kernel: the exchange copy of a layer's output as three bf16 planes (see there), pre-filled with
    // kPlaneSentinel; hxd[l]: the same for the DROPPED output (what layer l + 1 reads under inter-layer dropout), or null
    char* hx[kMaxJobs];
    char* hxd[kMaxJobs];
};

// Round 5: what the comment above describes is now also what the ISA does.
//   * H = 64 IPG is a template parameter.  With a run-time width every k-group of every product sat under its own
//     uniform branch (`16 it < kslice`), and hipcc's wait-count insertion has to assume the path on which NONE of the
//     younger loads was issued: the kernel of rounds 1-4 waited vmcnt(0) at the top of every step (i.e. for the
//     acknowledgement of the step's own write-through stores) and again right after the lower layer's rows of step t+1
//     had been requested (a memory round trip to another XCD) -- the software pipeline existed in the source only.
//     (A second cause, found on the way: sa_raise's host-visible word was reached through a generic pointer, a
//     flat_atomic_or, and ONE possibly-pending FLAT instruction in a loop turns every vmcnt wait behind it into
//     vmcnt(0); common.h now casts to the global address space.)
//   * no branch around a vector-memory instruction inside the loop: t = 0 is peeled, the prefetch of the last step is
//     clamped, layer 0 and the upper layers run two separate loop bodies, rows beyond the batch store to a dump slot,
//     stash / dropout are template parameters.
//   * POLL_AT (eighths of the step's input MFMAs): the first polling trip for the own layer's h[t-1] is issued after
//     that share of the input product and examined after all of it -- its L2 round trip runs under the remaining input
//     MFMAs instead of behind them.  4 measured best at S-LIBRI (3 / 4 / 5 / 6 / 8: 2.21 / 2.22 / 2.30 / 2.39 / 2.60 ms
//     per stack forward); earlier and the trip comes back stale, later and its latency is exposed.
//   * layer 0's three input-projection values are requested during the PREVIOUS step, behind its poll (they came from
//     HBM at the top of their own step, in front of the poll, in the in-order vector-memory queue).
//   * progress is reported to the layer above every `report_every` steps (4), that many at a time.  THIS was the
//     cross-layer coupling rounds 3-4 could not find (they thinned out the READS of the counter; the cost is the 32
//     memory-side read-modify-writes per layer and step on a line that 128 waves of the layer above poll): with a
//     report every step the issue of the next vector-memory instruction of the reporting wave stalls ~0.8 us per step
//     at L = 4 (in-kernel clocks, tools/gru_fused_timing.py) -- 1 / 2 / 4 / 8 / 16 steps per report: 2.69 / 2.18 / 2.22 /
//     2.27 / 2.37 ms per stack forward (the consumer trails by up to that many steps more: (L - 1) reports of fill).
// Bit-identical to the kernel of rounds 1-4 (same products in the same order), 2.93 -> 2.22 ms per S-LIBRI stack forward.
template <int IPG, int POLL_AT, bool STASH, bool DROP, bool TIMED>
__global__ __launch_bounds__(256) void gru_fwd_fused_kernel(PFusedFwd P) {
    constexpr int H = 64 * IPG, NTU = H / 16, KS = 16 * IPG;
    constexpr int kPollIt = POLL_AT >= 8 ? IPG : (IPG * POLL_AT) / 8;
    extern __shared__ __attribute__((aligned(16))) float psm[];
    __shared__ int s_role[2];
    SA_PERSIST_EXCLUSIVE(P.prio);
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        s_role[0] = x;
        s_role[1] = (int)(__hip_atomic_fetch_add(P.reg + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.reg_base);
    }
    __syncthreads();
    if (s_role[1] < 0 || s_role[1] >= 32) {
        if (threadIdx.x == 0) sa_raise(P.err, 2u);
        return;
    }
    const int sub = s_role[1] / NTU, grp = s_role[0] * (32 / NTU) + sub;
    if (sub >= 32 / NTU) return;
    const int role_x = s_role[1] - sub * NTU, l = grp / P.nbt, role_y = grp - l * P.nbt + P.bt0;
    if (l >= P.L) return;
    if ((P.fault & 1) && role_x == 1 && role_y == 0 && l == 0) return;  // injected fault: a group one member short
    int budget = P.spin_limit;  // wave-uniform; 0 after the first timeout: the call is lost, drain quickly
    const int B = P.B, T = P.T;
    const bool stamper = P.stamp && threadIdx.x == 0 && role_x + role_y + l == 0;
    if (stamper) P.stamp[0] = wall_clock64();
    float* red = psm;             // [2][4 waves][4 sums][256]  (the weights live in registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int u0 = role_x * 16, b0 = role_y * 16;
    const int bi = tid >> 4, uj = tid & 15;
    const int b = b0 + bi, u = u0 + uj;
    const bool live = b < B;
    const float e_br = P.b_hh[l][u], e_bz = P.b_hh[l][H + u], e_bn = P.b_hh[l][2 * H + u];
    float* h_out = P.h_out[l];
    const long hs_t = (long)B * H;
    const int kbeg = wave * KS;
    const int brow = min(b0 + i, B - 1);
    __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)h_out, 0, 0x7fffffff, 0x00020000);
    unsigned* my_prog = P.prog + l * P.nbt_all + role_y;
    unsigned* errp = P.err;
    // per-thread store targets: base + t * stride (a row beyond the batch keeps hitting its dump slot)
    float* dump = P.dump + blockIdx.x * 256 + tid;
    float* p_h = live ? h_out + (long)b * H + u : dump;
    const long s_h = live ? hs_t : 0;
    float* p_hd = (DROP && live && P.h_drop[l]) ? P.h_drop[l] + (long)b * H + u : dump;
    const long s_hd = (DROP && live && P.h_drop[l]) ? hs_t : 0;
    float* p_st = (STASH && live) ? P.stash[l] + (long)b * P.rb * 5 * H + u : dump;
    const long s_st = (STASH && live) ? (long)P.rt * 5 * H : 0;
    const int so = (STASH && live) ? H : 0;  // distance between a row's stashed gates
    const SaDrop drop = P.drop;
    const unsigned drop_stream = P.drop_stream0 + (unsigned)l;
    const long drop_idx0 = (long)(live ? b : 0) * H + u;
    const int rep_tid = 0;
    const int exp_every = P.report_every;  // a power of two
    float hp = 0.f;
    // the W_hh fragments this lane feeds to its MFMAs: resident in registers for the whole launch
    float4 wh[IPG][3];
#pragma unroll
    for (int it = 0; it < IPG; ++it)
#pragma unroll
        for (int n = 0; n < 3; ++n)
            wh[it][n] = *reinterpret_cast<const float4*>(P.w_hh[l] + (long)(n * H + u0 + i) * H + kbeg + 16 * it + 4 * g);
    // TIMED: input / poll / recurrent MFMA / rest in 10 ns ticks, then shader-clock cycles and wall ticks of the whole loop
    // (their ratio is the clock the XCD actually ran at), polling trips beyond the first, and blocked waits for the lower layer
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool timed = TIMED && P.timing != nullptr && tid == 0;
    if (timed) { tprev = wall_clock64(); tacc[4] = clock64(); tacc[5] = tprev; }
#define SA_TICK(k) if (TIMED && timed) { const unsigned long long now = wall_clock64(); tacc[k] += now - tprev; tprev = now; }
    __syncthreads();

    f32x4v a[IPG];  // the gathered k-slice of the own layer's h[t-1]
    auto issue_poll = [&](int tprev_idx) {
        const int abase = (int)((((long)tprev_idx * B + brow) * H + kbeg + 4 * g) * 4);
#pragma unroll
        for (int it = 0; it < IPG; ++it)
            a[it] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hres, abase + 64 * it, 0, 16));
    };
    auto poll_until_fresh = [&](int tprev_idx, bool first_trip_issued) {  // flag-less hand-off inside the XCD
        for (int spins = 0;; ++spins) {
            if (!(first_trip_issued && spins == 0)) {
                asm volatile("" ::: "memory");  // every trip re-issues its loads
                issue_poll(tprev_idx);
            }
            bool stale = false;
#pragma unroll
            for (int it = 0; it < IPG; ++it) stale |= has_sentinel(a[it]);
            if (__builtin_amdgcn_ballot_w64(stale) == 0) break;
            if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
        }
    };
    auto recurrent = [&](f32x4 (&acc)[3], f32x4& acc_hn) {  // r and z share accumulators with the input product
#pragma unroll
        for (int it = 0; it < IPG; ++it)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const float4 w = wh[it][n];
                f32x4& dst = n == 2 ? acc_hn : acc[n];
                dst = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, w.x, dst, 0, 0, 0);
                dst = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, w.y, dst, 0, 0, 0);
                dst = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, w.z, dst, 0, 0, 0);
                dst = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, w.w, dst, 0, 0, 0);
            }
    };
    // reduce over the four waves, gates, publish: everything after the step's MFMAs.  `report`: tell the layer above
    // that the PREVIOUS step is out (every thread of the block has consumed loads it issued after those stores: they
    // are acknowledged -- vmcnt is one in-order queue)
    auto finish = [&](int t, const f32x4 (&acc)[3], const f32x4& acc_hn, float e_ai_r, float e_ai_z, float e_ai_n,
                      bool report) {
        float* rd = red + (t & 1) * 4096;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (g * 4 + r) * 16 + i;
            rd[(wave * 4 + 0) * 256 + o] = acc[0][r];
            rd[(wave * 4 + 1) * 256 + o] = acc[1][r];
            rd[(wave * 4 + 2) * 256 + o] = acc[2][r];
            rd[(wave * 4 + 3) * 256 + o] = acc_hn[r];
        }
        __syncthreads();
        if (exp_every == 1) {
            if (report && tid == rep_tid) __hip_atomic_fetch_add(my_prog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (report && tid == rep_tid && (t & (exp_every - 1)) == 0)
                __hip_atomic_fetch_add(my_prog, (unsigned)exp_every, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float sr = 0.f, sz = 0.f, sin_ = 0.f, shn = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sr += rd[(w * 4 + 0) * 256 + tid];
            sz += rd[(w * 4 + 1) * 256 + tid];
            sin_ += rd[(w * 4 + 2) * 256 + tid];
            shn += rd[(w * 4 + 3) * 256 + tid];
        }
        const float r = sigmoidf_(e_ai_r + sr + e_br);
        const float z = sigmoidf_(e_ai_z + sz + e_bz);
        const float q = shn + e_bn;
        const float n = tanhf(e_ai_n + sin_ + r * q);
        const float h = (1.0f - z) * n + z * hp;
        __hip_atomic_store(p_h + (long)t * s_h, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
        if (DROP)  // what the layer above reads, written through like h and AFTER it (the own layer's next step waits for h)
            __hip_atomic_store(p_hd + (long)t * s_hd,
                               h * sa_drop_factor(drop, drop_stream, (uint64_t)((long)t * hs_t + drop_idx0)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (STASH) {  // streaming stores: the stash must not push anything out of the XCD's L2
            float* st = p_st + (long)t * s_st;
            __builtin_nontemporal_store(r, st);
            __builtin_nontemporal_store(z, st + so);
            __builtin_nontemporal_store(n, st + 2 * so);
            __builtin_nontemporal_store(q, st + 3 * so);
            __builtin_nontemporal_store(hp, st + 4 * so);
        }
        hp = h;
    };

    if (l == 0) {
        // ---- layer 0: its input projection is one GEMM before the launch (ai0)
        const float* p_ai = P.ai0 + (long)(live ? b : 0) * P.rb * 3 * H + u;
        const long s_ai = (long)P.rt * 3 * H;
        float nx_r, nx_z, nx_n;  // the values of the step about to be processed, requested a step ago
        auto fetch_ai = [&](int tt) { const float* q = p_ai + (long)tt * s_ai; nx_r = q[0]; nx_z = q[H]; nx_n = q[2 * H]; };
        fetch_ai(0);
        if (T > 0) {  // t = 0 (peeled: no recurrent phase, nobody to wait for)
            f32x4 acc[3], acc_hn = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float e_r = nx_r, e_z = nx_z, e_n = nx_n;
            finish(0, acc, acc_hn, e_r, e_z, e_n, false);
            fetch_ai(T > 1 ? 1 : 0);
            SA_TICK(3)
        }
        for (int t = 1; t < T; ++t) {
            f32x4 acc[3], acc_hn = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float e_r = nx_r, e_z = nx_z, e_n = nx_n;
            SA_TICK(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // pacing of the first polling trip (see gru_fwd_chunk_kernel)
            poll_until_fresh(t - 1, false);
            SA_TICK(1)
            // the next step's values (HBM): requested behind the poll -- in front of it they would hold the poll's data
            // back (one in-order queue), at the end of the step they would hold the NEXT poll's back
            fetch_ai(t + 1 < T ? t + 1 : t);  // (the last step re-reads its own: unused)
            __builtin_amdgcn_sched_barrier(0);
            recurrent(acc, acc_hn);
            SA_TICK(2)
            finish(t, acc, acc_hn, e_r, e_z, e_n, true);
            SA_TICK(3)
        }
    } else {
        // ---- layers >= 1: the input projection W_ih h_{l-1}[t] is formed in the kernel, on rows fetched a step ago
        const float bi_r = P.b_ih[l][u], bi_z = P.b_ih[l][H + u], bi_n = P.b_ih[l][2 * H + u];
        const float* lower = (DROP && P.h_drop[l - 1]) ? P.h_drop[l - 1] : P.h_out[l - 1];
        __amdgpu_buffer_rsrc_t lres = __builtin_amdgcn_make_buffer_rsrc((void*)lower, 0, 0x7fffffff, 0x00020000);
        const unsigned* lower_prog = P.prog + (l - 1) * P.nbt_all + role_y;
        unsigned avail = 0;        // steps of the lower layer known to be published
        unsigned cnt_pending = 0;  // the arrival counter as requested a step ago
        float4 wn[IPG][3];         // W_ih fragments, resident like W_hh
#pragma unroll
        for (int it = 0; it < IPG; ++it)
#pragma unroll
            for (int n = 0; n < 3; ++n)
                wn[it][n] = *reinterpret_cast<const float4*>(P.w_ih[l] + (long)(n * H + u0 + i) * H + kbeg + 16 * it + 4 * g);
        f32x4v an[IPG];            // the lower layer's rows of the step about to be processed
        auto wait_lower = [&](int tt) {  // the lower layer has published step tt (every wave polls for itself)
            if (avail >= (unsigned)(tt + 1)) return;
            int spins = 0;
            unsigned c;
            while ((c = __hip_atomic_load(lower_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned)NTU * (unsigned)(tt + 1)) {
                if (++spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
            }
            avail = c / (unsigned)NTU;
        };
        auto issue_rows = [&](int tt) {  // from memory (another XCD wrote them through): slow, read once
            const int abase = (int)((((long)tt * B + brow) * H + kbeg + 4 * g) * 4);
#pragma unroll
            for (int it = 0; it < IPG; ++it)
                an[it] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(lres, abase + 64 * it, 0, 16 | 2));  // sc1 + nt
        };
        auto mfma_input = [&](f32x4 (&acc)[3], int lo, int hi) {
#pragma unroll
            for (int it = 0; it < IPG; ++it)
                if (it >= lo && it < hi)
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[it].x, wn[it][n].x, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[it].y, wn[it][n].y, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[it].z, wn[it][n].z, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[it].w, wn[it][n].w, acc[n], 0, 0, 0);
                    }
        };
        if (T > 0) {
            wait_lower(0);
            issue_rows(0);
            // t = 0 (peeled: no recurrent phase)
            f32x4 acc[3], acc_hn = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            mfma_input(acc, 0, IPG);
            SA_TICK(0)
            const int t1 = T > 1 ? 1 : 0;
            wait_lower(t1);
            issue_rows(t1);
            cnt_pending = __hip_atomic_load(lower_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            SA_TICK(2)
            finish(0, acc, acc_hn, bi_r, bi_z, bi_n, false);
            SA_TICK(3)
        }
        // Order of the vector-memory operations of a step (one in-order queue): [input MFMAs on rows fetched a step ago,
        // the first polling trip issued part-way through] -> [trip examined; re-polled until fresh] -> [the counter value
        // requested a step ago is consumed, the rows of step t+1 and the counter are requested: the slow loads go out
        // only now, behind the poll] -> [recurrent MFMAs, reduce, gates, publish].
        for (int t = 1; t < T; ++t) {
            f32x4 acc[3], acc_hn = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            mfma_input(acc, 0, kPollIt);
            if (kPollIt < IPG) {
                __builtin_amdgcn_sched_barrier(0);
                SA_TICK(0)
                issue_poll(t - 1);
                __builtin_amdgcn_sched_barrier(0);
                SA_TICK(6)
                mfma_input(acc, kPollIt, IPG);
                __builtin_amdgcn_sched_barrier(0);
                SA_TICK(7)
            } else {
                SA_TICK(0)
            }
            poll_until_fresh(t - 1, kPollIt < IPG);
            SA_TICK(1)
            {
                asm volatile("" : "+v"(cnt_pending) :: "memory");  // consumed HERE, not where it was loaded
                const unsigned got = cnt_pending / (unsigned)NTU;
                if (got > avail) avail = got;
                const int tn = t + 1 < T ? t + 1 : t;  // (the last step re-reads its own rows: unused)
                wait_lower(tn);  // almost always satisfied by the value requested a step ago
                issue_rows(tn);
                cnt_pending = __hip_atomic_load(lower_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_sched_barrier(0);  // requested BEFORE the recurrent MFMAs (hipcc sinks them below otherwise)
            }
            recurrent(acc, acc_hn);
            SA_TICK(2)
            finish(t, acc, acc_hn, bi_r, bi_z, bi_n, true);
            SA_TICK(3)
        }
    }
    if (TIMED && timed) {
        tacc[4] = clock64() - tacc[4]; tacc[5] = wall_clock64() - tacc[5];
        unsigned long long* o = P.timing + 8 * ((l * P.nbt + role_y - P.bt0) * NTU + role_x);
        for (int k = 0; k < 8; ++k) atomicAdd(&o[k], tacc[k]);
    }
#undef SA_TICK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        // reports so far: steps 1 .. T-1 one each (or, batched, exp_every at every multiple of exp_every below T); the
        // block ends having reported T steps
        unsigned done = T > 1 ? (unsigned)(T - 1) : 0u;
        if (exp_every > 1) done = T > 1 ? (unsigned)(((T - 1) / exp_every) * exp_every) : 0u;
        __hip_atomic_fetch_add(my_prog, (unsigned)T - done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (P.stamp && threadIdx.x == 0) atomicMax(P.stamp + 1, (unsigned long long)wall_clock64());  // the LAST block out
}

// ------------------------------------------------------------------- fused forward layer wavefront on bf16 planes (r6)
// gru_fwd_fused_kernel with both of a step's products on the bf16 MFMA.  Rounds 3-5 established that every cycle of the 96
// f32-input MFMAs of the recurrent product (1.43 us) is on the step's critical path, that the upper layers carry 96 more for
// their input product (2.9 us of matrix pipe in a 4.2 us step), and that splitting the freshly gathered fp32 rows into bf16
// pieces on the CONSUMER side (176 VALU instructions per lane and step, on the same path) eats what the faster MFMAs save
// (profiles/r03_recurrence_split_bf16_experiment.txt).  Here the PRODUCER splits: a thread owns one h value in a register
// when it is born; it pairs up with its neighbour lane (one DPP move), splits the pair exactly into three bf16 pieces
// (sa_split2, 11 instructions) and publishes the three packed pairs with two 4-byte stores into the layer's PLANES buffer
//     hx[t][batch tile][plane 3][k / 32][row 16][k % 32]   bf16  (1 KB per (plane, k / 32) tile: one load instruction of a wave)
// which is the A operand of v_mfma_f32_16x16x32_bf16 as it lies: lane (row i, k-group g) reads the 16 bytes k = 8 g .. 8 g + 7
// of row i.  Consumers -- the own layer's next step inside the XCD, the layer above from another XCD -- gather 12 x 16 B per
// lane instead of 8 (48 KB per block and step instead of 32) and run 72 sixteen-cycle MFMAs per product instead of 96
// thirty-two-cycle ones (the six piece products that carry an fp32 product down to 2^-26 of it, smallest first: the
// arithmetic of the split-bf16 GEMMs, gemm_f32.hip): 1152 instead of 3072 matrix-pipe cycles per product.  The weights'
// fragments are split once at launch (2 x 144 registers).  The data is its own flag as before: the planes are pre-filled
// with 0xffffffff (a pair of bf16 NaNs no split of a finite number produces), a dword store is atomic, and a consumer folds
// the 48 dwords of a trip with v_max3_u32 -- one compare per trip.
// What else changed against gru_fwd_fused_kernel, all on the reduce .. publish path that follows the MFMAs:
//   * thread (wave w, lane (i, g)) finishes element (row 4 g + w, unit i) -- the element whose partial sums sit in register w
//     of the SAME lane in every wave (16x16 C/D layout: row = 4 (lane >> 4) + reg, col = lane & 15).  The four waves' partial
//     sums cross LDS as 16-byte vectors {r, z, n_in, n_h}: 4 ds_write_b128 + 4 ds_read_b128 per thread instead of 16 + 16
//     four-byte accesses; neighbouring lanes hold neighbouring units, which is what the pair-wise publish wants;
//   * sigmoid and tanh on v_exp_f32 / v_rcp_f32 (1 ulp each; tanh x = 2 sigmoid(2 x) - 1: absolute error < 2e-7) instead of
//     expf / IEEE division / tanhf (~100 instructions on the critical path of every step);
//   * the fp32 h_out (what the classifier, the weight-gradient packs and the caller read) leaves with a plain store behind
//     the planes and needs no sentinel fill.
// NOT bit-identical to the step kernels any more (different products, different summation order, different transcendental
// rounding): tests/test_gpu_blocks.py holds the stack against an fp64 restatement with an error budget instead, and every
// comparison against the CPU restatements and the golden fixtures is unchanged.  Even IPG only (a wave's K slice is whole 32-k steps): H = 128, 256, 384, 512.
constexpr unsigned kPlaneSentinel = 0xffffffffu;
__host__ __device__ inline size_t hx_step_bytes(int H) { return (size_t)3 * (H / 32) * 1024; }  // per (t, batch tile)

__device__ __forceinline__ float sa_fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-SA_LOG2E * x));
}
// tanh |x| = (1 - e) / (1 + e), e = exp(-2 |x|); below 0.25 the odd series through x^9 (next term 9e-9 of x) -- the quotient
// form cancels there (the first version, 2 sigmoid(2 x) - 1, carried 1.2e-7 ABSOLUTE error at any x: fine for h, visible in
// the smallest parameter gradients of the seq2seq config).  Relative error < 3e-7 everywhere.
__device__ __forceinline__ float sa_fast_tanh(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(-2.0f * SA_LOG2E * ax);
    const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    const float small = ax * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.0f / 2835.0f, -17.0f / 315.0f), 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
    return copysignf(ax < 0.25f ? small : big, x);
}
template <int N>
__device__ __forceinline__ void sa_wait_vmcnt() {  // s_waitcnt vmcnt(N): all but the youngest N vector-memory operations done
    static_assert(N >= 0 && N <= 9, "extend the table");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}
__device__ __forceinline__ float sa_quad_swap1(float v) {  // lane i <-> lane i ^ 1 (DPP quad_perm [1, 0, 3, 2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));
}

template <int IPG, int POLL_AT, bool STASH, bool DROP, bool TIMED>
__global__ __launch_bounds__(256) void gru_fwd_planes_kernel(PFusedFwd P) {
    static_assert((IPG & 1) == 0, "a wave's K slice must be whole 32-k MFMA steps");
    constexpr int H = 64 * IPG, NTU = H / 16, KS = 16 * IPG, NK = IPG / 2, NKT = H / 32;
    constexpr int kPollIt = POLL_AT >= 8 ? NK : (NK * POLL_AT) / 8;
    constexpr int kStep = 3 * NKT * 1024;  // bytes of the planes per (t, batch tile)
    // vector-memory stores a step issues BEHIND its two publish stores (finish()): h_out, the stash, the dropped copy
    constexpr int kAfterPublish = 1 + (STASH ? 5 : 0) + (DROP ? 3 : 0);
    extern __shared__ __attribute__((aligned(16))) float psm[];
    __shared__ int s_role[2];
    SA_PERSIST_EXCLUSIVE(P.prio);
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        s_role[0] = x;
        s_role[1] = (int)(__hip_atomic_fetch_add(P.reg + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.reg_base);
    }
    __syncthreads();
    if (s_role[1] < 0 || s_role[1] >= 32) {
        if (threadIdx.x == 0) sa_raise(P.err, 2u);
        return;
    }
    const int sub = s_role[1] / NTU, grp = s_role[0] * (32 / NTU) + sub;
    if (sub >= 32 / NTU) return;
    const int role_x = s_role[1] - sub * NTU, l = grp / P.nbt, role_y = grp - l * P.nbt + P.bt0;
    if (l >= P.L) return;
    if ((P.fault & 1) && role_x == 1 && role_y == 0 && l == 0) return;  // injected fault: a group one member short
    int budget = P.spin_limit;  // wave-uniform; 0 after the first timeout: the call is lost, drain quickly
    const int B = P.B, T = P.T;
    const bool stamper = P.stamp && threadIdx.x == 0 && role_x + role_y + l == 0;
    if (stamper) P.stamp[0] = wall_clock64();
    float4* red = reinterpret_cast<float4*>(psm);  // [2][source wave 4][register = destination wave 4][lane 64] {r, z, n_in, n_h}
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int u0 = role_x * 16, b0 = role_y * 16;
    const int row = 4 * g + wave;  // the element this thread finishes: (row, unit i) of the block's 16 x 16 tile
    const int b = b0 + row, u = u0 + i;
    const bool live = b < B;
    const float e_br = P.b_hh[l][u], e_bz = P.b_hh[l][H + u], e_bn = P.b_hh[l][2 * H + u];
    const long hs_t = (long)B * H;
    const int kbeg = wave * KS;
    const int s_hx = P.nbt_all * kStep;  // bytes of a layer's planes per time step
    __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)P.hx[l], 0, 0x7fffffff, 0x00020000);
    const int gather_off = i * 64 + g * 16 + wave * NK * 1024;  // lane part of a gather address (+ plane / k-step / time: scalar)
    const int tile_off = role_y * kStep;
    unsigned* my_prog = P.prog + l * P.nbt_all + role_y;
    unsigned* errp = P.err;
    // per-thread store targets: base + t * stride (a row beyond the batch keeps hitting its dump slot)
    float* dump = P.dump + blockIdx.x * 256 + tid;
    float* p_h = live ? P.h_out[l] + (long)b * H + u : dump;
    const long s_h = live ? hs_t : 0;
    float* p_hd = (DROP && live && P.h_drop[l]) ? P.h_drop[l] + (long)b * H + u : dump;
    const long s_hd = (DROP && live && P.h_drop[l]) ? hs_t : 0;
    float* p_st = (STASH && live) ? P.stash[l] + (long)b * P.rb * 5 * H + u : dump;
    const long s_st = (STASH && live) ? (long)P.rt * 5 * H : 0;
    const int so = (STASH && live) ? H : 0;  // distance between a row's stashed gates
    // the publish: lanes 2 j and 2 j + 1 hold the same three packed pairs; store A carries plane 0 (even lane) and plane 1
    // (odd lane), store B plane 2 (both lanes: same address, same value)
    const bool odd = (i & 1) != 0;
    const int pub_col = ((role_x & 1) * 16 + (i & ~1)) * 2 + row * 64 + (role_x >> 1) * 1024 + tile_off;
    const int pub_a = pub_col + (odd ? NKT * 1024 : 0), pub_b = pub_col + 2 * NKT * 1024;
    char* hx_own = P.hx[l];
    // DROP: the dropped copy for the layer above; the top layer has none and re-publishes h into its own planes (the same
    // values at the same addresses: no branch around the stores inside the loop)
    const bool has_drop = DROP && P.hxd[l] != nullptr;
    char* hx_drop = has_drop ? P.hxd[l] : P.hx[l];
    const SaDrop drop = P.drop;
    const unsigned drop_stream = P.drop_stream0 + (unsigned)l;
    const long drop_idx0 = (long)(live ? b : 0) * H + u;
    const int rep_tid = 0;
    const int exp_every = P.report_every;  // a power of two
    float hp = 0.f;
    // Weight fragments: rows n H + u0 + i, this wave's K slice, split once.  The two sets are 288 registers; the first
    // `pin` k-steps of a set are pinned in the ACCUMULATOR half of the register file ("+a": an MFMA reads its A / B operands
    // from there directly) -- left alone hipcc keeps the fragments in the vector half's class, parks them in accumulator
    // registers as spills and copies ~130 of them back per step (v_accvgpr_read, on the MFMAs' own issue port).
    auto load_w = [&](const float* W, SaBf3 (&w)[NK][3], int pin) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const float* q = W + (long)(n * H + u0 + i) * H + kbeg + 32 * kk + 8 * g;
                w[kk][n] = sa_split8(*reinterpret_cast<const float4*>(q), *reinterpret_cast<const float4*>(q + 4));
                if (kk < pin) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+a"(w[kk][n].p[pl]));
                }
            }
    };
    SaBf3 wh[NK][3];
    load_w(P.w_hh[l], wh, NK / 2);
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool timed = TIMED && P.timing != nullptr && tid == 0;
    if (timed) { tprev = wall_clock64(); tacc[4] = clock64(); tacc[5] = tprev; }
#define SA_TICK(k) if (TIMED && timed) { const unsigned long long now = wall_clock64(); tacc[k] += now - tprev; tprev = now; }
    __syncthreads();

    sa_bf16x8 a[NK][3];  // the gathered k-slice of the own layer's h[t-1]: [k step][plane]
    auto issue_gather = [&](sa_bf16x8 (&dst)[NK][3], __amdgpu_buffer_rsrc_t res, int tt, int aux) {
        const int sbase = tt * s_hx + tile_off;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                dst[kk][pl] = __builtin_bit_cast(sa_bf16x8, aux == 16
                    ? __builtin_amdgcn_raw_buffer_load_b128(res, gather_off, sbase + (pl * NKT + kk) * 1024, 16)
                    : __builtin_amdgcn_raw_buffer_load_b128(res, gather_off, sbase + (pl * NKT + kk) * 1024, 16 | 2));
    };
    auto any_sentinel = [&](const sa_bf16x8 (&v)[NK][3]) {
        unsigned m = 0u;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const uint4 q = __builtin_bit_cast(uint4, v[kk][pl]);
                m = max(max(m, q.x), q.y);
                m = max(max(m, q.z), q.w);
            }
        return m == kPlaneSentinel;
    };
    auto poll_until_fresh = [&](int tprev_idx, bool first_trip_issued) {  // flag-less hand-off inside the XCD
        for (int spins = 0;; ++spins) {
            if (!(first_trip_issued && spins == 0)) {
                asm volatile("" ::: "memory");  // every trip re-issues its loads
                issue_gather(a, hres, tprev_idx, 16);
            }
            if (__builtin_amdgcn_ballot_w64(any_sentinel(a)) == 0) break;
            if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
        }
    };
    // one product of a step: 6 piece products x 3 gates per 32-k step, smallest pieces first, gates interleaved (consecutive
    // MFMAs never share an accumulator)
    auto x6 = [&](const sa_bf16x8 (&v)[NK][3], const SaBf3 (&w)[NK][3], f32x4& d0, f32x4& d1, f32x4& d2, int lo, int hi) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
            if (kk >= lo && kk < hi)
#pragma unroll
                for (int o = 0; o < 6; ++o) {
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[kk][SA_X6_PA(o)], w[kk][0].p[SA_X6_PB(o)], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[kk][SA_X6_PA(o)], w[kk][1].p[SA_X6_PB(o)], d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[kk][SA_X6_PA(o)], w[kk][2].p[SA_X6_PB(o)], d2, 0, 0, 0);
                }
    };
    auto publish = [&](char* base, int t, float v) {
        const float vn = sa_quad_swap1(v);
        unsigned p1, p2, p3;
        sa_split2(odd ? vn : v, odd ? v : vn, p1, p2, p3);
        char* q = base + (long)t * s_hx;
        __hip_atomic_store(reinterpret_cast<unsigned*>(q + pub_a), odd ? p2 : p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned*>(q + pub_b), p3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // reduce over the four waves, gates, publish: everything after the step's MFMAs (see gru_fwd_fused_kernel for `report`)
    auto finish = [&](int t, const f32x4 (&acc)[3], const f32x4& acc_hn, float e_ai_r, float e_ai_z, float e_ai_n,
                      bool report) {
        float4* rd = red + (t & 1) * 1024;
#pragma unroll
        for (int r = 0; r < 4; ++r) rd[(wave * 4 + r) * 64 + lane] = make_float4(acc[0][r], acc[1][r], acc[2][r], acc_hn[r]);
        __syncthreads();
        if (exp_every == 1) {
            if (report && tid == rep_tid) __hip_atomic_fetch_add(my_prog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (report && tid == rep_tid && (t & (exp_every - 1)) == 0)
                __hip_atomic_fetch_add(my_prog, (unsigned)exp_every, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float4 s0 = rd[(0 * 4 + wave) * 64 + lane], s1 = rd[(1 * 4 + wave) * 64 + lane];
        const float4 s2 = rd[(2 * 4 + wave) * 64 + lane], s3 = rd[(3 * 4 + wave) * 64 + lane];
        const float sr = ((s0.x + s1.x) + s2.x) + s3.x, sz = ((s0.y + s1.y) + s2.y) + s3.y;
        const float sin_ = ((s0.z + s1.z) + s2.z) + s3.z, shn = ((s0.w + s1.w) + s2.w) + s3.w;
        const float r = sa_fast_sigmoid(e_ai_r + sr + e_br);
        const float z = sa_fast_sigmoid(e_ai_z + sz + e_bz);
        const float q = shn + e_bn;
        const float n = sa_fast_tanh(e_ai_n + sin_ + r * q);
        const float h = live ? (1.0f - z) * n + z * hp : 0.f;  // rows beyond the batch publish zeros: a tile has no holes
        publish(hx_own, t, h);
        if (DROP) {  // what the layer above reads: published like h and AFTER it (the own layer's next step waits for h)
            const float hd = h * sa_drop_factor(drop, drop_stream, (uint64_t)((long)t * hs_t + drop_idx0));
            publish(hx_drop, t, has_drop ? hd : h);
            p_hd[(long)t * s_hd] = hd;
        }
        p_h[(long)t * s_h] = h;
        if (STASH) {  // streaming stores: the stash must not push anything out of the XCD's L2
            float* st = p_st + (long)t * s_st;
            __builtin_nontemporal_store(r, st);
            __builtin_nontemporal_store(z, st + so);
            __builtin_nontemporal_store(n, st + 2 * so);
            __builtin_nontemporal_store(q, st + 3 * so);
            __builtin_nontemporal_store(hp, st + 4 * so);
        }
        hp = h;
    };

    if (l == 0) {
        // ---- layer 0: its input projection is one GEMM before the launch (ai0)
        const float* p_ai = P.ai0 + (long)(live ? b : 0) * P.rb * 3 * H + u;
        const long s_ai = (long)P.rt * 3 * H;
        float nx_r, nx_z, nx_n;  // the values of the step about to be processed, requested a step ago
        auto fetch_ai = [&](int tt) { const float* q = p_ai + (long)tt * s_ai; nx_r = q[0]; nx_z = q[H]; nx_n = q[2 * H]; };
        fetch_ai(0);
        if (T > 0) {  // t = 0 (peeled: no recurrent phase, nobody to wait for)
            f32x4 acc[3], acc_hn = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float e_r = nx_r, e_z = nx_z, e_n = nx_n;
            finish(0, acc, acc_hn, e_r, e_z, e_n, false);
            fetch_ai(T > 1 ? 1 : 0);
            SA_TICK(3)
        }
        for (int t = 1; t < T; ++t) {
            f32x4 acc[3], acc_hn = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float e_r = nx_r, e_z = nx_z, e_n = nx_n;
            SA_TICK(0)
            // pacing of the first polling trip (see gru_fwd_chunk_kernel): it leaves once this step's own PUBLISH stores are
            // acknowledged -- not the stash / h_out stores behind them, whose acknowledgements come later and say nothing
            // about the neighbours' rows (r6: 1.67 - 1.74 -> 1.64 ms per S-LIBRI stack forward)
            sa_wait_vmcnt<kAfterPublish>();
            poll_until_fresh(t - 1, false);
            SA_TICK(1)
            fetch_ai(t + 1 < T ? t + 1 : t);  // behind the poll (one in-order queue); the last step re-reads its own: unused
            __builtin_amdgcn_sched_barrier(0);
            x6(a, wh, acc[0], acc[1], acc_hn, 0, NK);
            SA_TICK(2)
            finish(t, acc, acc_hn, e_r, e_z, e_n, true);
            SA_TICK(3)
        }
    } else {
        // ---- layers >= 1: the input projection W_ih h_{l-1}[t] is formed in the kernel, on planes fetched a step ago
        const float bi_r = P.b_ih[l][u], bi_z = P.b_ih[l][H + u], bi_n = P.b_ih[l][2 * H + u];
        char* lower = (DROP && P.hxd[l - 1]) ? P.hxd[l - 1] : P.hx[l - 1];
        __amdgpu_buffer_rsrc_t lres = __builtin_amdgcn_make_buffer_rsrc((void*)lower, 0, 0x7fffffff, 0x00020000);
        const unsigned* lower_prog = P.prog + (l - 1) * P.nbt_all + role_y;
        unsigned avail = 0;        // steps of the lower layer known to be published
        unsigned cnt_pending = 0;  // the arrival counter as requested a step ago
        SaBf3 wn[NK][3];           // W_ih fragments, resident like W_hh
        load_w(P.w_ih[l], wn, NK);
        sa_bf16x8 an[NK][3];       // the lower layer's planes of the step about to be processed
        auto wait_lower = [&](int tt) {  // the lower layer has published step tt (every wave polls for itself)
            if (avail >= (unsigned)(tt + 1)) return;
            int spins = 0;
            unsigned c;
            while ((c = __hip_atomic_load(lower_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned)NTU * (unsigned)(tt + 1)) {
                if (++spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
            }
            avail = c / (unsigned)NTU;
        };
        if (T > 0) {
            wait_lower(0);
            issue_gather(an, lres, 0, 18);
            f32x4 acc[3], acc_hn = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            x6(an, wn, acc[0], acc[1], acc[2], 0, NK);
            SA_TICK(0)
            const int t1 = T > 1 ? 1 : 0;
            wait_lower(t1);
            issue_gather(an, lres, t1, 18);
            cnt_pending = __hip_atomic_load(lower_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            SA_TICK(2)
            finish(0, acc, acc_hn, bi_r, bi_z, bi_n, false);
            SA_TICK(3)
        }
        for (int t = 1; t < T; ++t) {
            f32x4 acc[3], acc_hn = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            x6(an, wn, acc[0], acc[1], acc[2], 0, kPollIt);
            if (kPollIt < NK) {
                __builtin_amdgcn_sched_barrier(0);
                SA_TICK(0)
                issue_gather(a, hres, t - 1, 16);
                __builtin_amdgcn_sched_barrier(0);
                SA_TICK(6)
                x6(an, wn, acc[0], acc[1], acc[2], kPollIt, NK);
                __builtin_amdgcn_sched_barrier(0);
                SA_TICK(7)
            } else {
                SA_TICK(0)
            }
            // pacing as in layer 0: the first trip leaves once the previous step's own stores are acknowledged (with a narrow
            // layer the input product alone is too short a delay: a trip issued right behind the stores comes back stale)
            // (between scheduling barriers: the wait is no memory operation to the MFMAs, and hipcc hoisted it to the top of
            // the input product -- the whole product then ran BEHIND the acknowledgements: 1.67 -> 1.90 ms per stack forward)
            if (kPollIt >= NK) {
                __builtin_amdgcn_sched_barrier(0);
                sa_wait_vmcnt<kAfterPublish>();
                __builtin_amdgcn_sched_barrier(0);
            }
            poll_until_fresh(t - 1, kPollIt < NK);
            SA_TICK(1)
            {
                asm volatile("" : "+v"(cnt_pending) :: "memory");  // consumed HERE, not where it was loaded
                const unsigned got = cnt_pending / (unsigned)NTU;
                if (got > avail) avail = got;
                const int tn = t + 1 < T ? t + 1 : t;  // (the last step re-reads its own rows: unused)
                wait_lower(tn);  // almost always satisfied by the value requested a step ago
                issue_gather(an, lres, tn, 18);
                cnt_pending = __hip_atomic_load(lower_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_sched_barrier(0);  // requested BEFORE the recurrent MFMAs (hipcc sinks them below otherwise)
            }
            x6(a, wh, acc[0], acc[1], acc_hn, 0, NK);
            SA_TICK(2)
            finish(t, acc, acc_hn, bi_r, bi_z, bi_n, true);
            SA_TICK(3)
        }
    }
    if (TIMED && timed) {
        tacc[4] = clock64() - tacc[4]; tacc[5] = wall_clock64() - tacc[5];
        unsigned long long* o = P.timing + 8 * ((l * P.nbt + role_y - P.bt0) * NTU + role_x);
        for (int k = 0; k < 8; ++k) atomicAdd(&o[k], tacc[k]);
    }
#undef SA_TICK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned done = T > 1 ? (unsigned)(T - 1) : 0u;
        if (exp_every > 1) done = T > 1 ? (unsigned)(((T - 1) / exp_every) * exp_every) : 0u;
        __hip_atomic_fetch_add(my_prog, (unsigned)T - done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (P.stamp && threadIdx.x == 0) atomicMax(P.stamp + 1, (unsigned long long)wall_clock64());  // the LAST block out
}

// --------------------------------------------------------------------- persistent forward chunk on bf16 planes (r6)
// gru_fwd_chunk_kernel with the recurrent product on producer-written bf16 planes -- the two directions of a bidirectional
// layer (every shipped config: 4 x biGRU-256), whose input projections stay GEMMs.  Same exchange layout, publish, gather,
// reduce and gate arithmetic as gru_fwd_planes_kernel (see there); one planes buffer per direction, re-filled per layer.
template <int IPG, bool STASH>
__global__ __launch_bounds__(256) void gru_fwd_chunk_planes_kernel(PFwdJobs P) {
    static_assert((IPG & 1) == 0, "a wave's K slice must be whole 32-k MFMA steps");
    constexpr int H = 64 * IPG, NTU = H / 16, KS = 16 * IPG, NK = IPG / 2, NKT = H / 32;
    constexpr int kStep = 3 * NKT * 1024;
    extern __shared__ __attribute__((aligned(16))) float psm[];
    __shared__ int s_role[2];
    SA_PERSIST_EXCLUSIVE(P.prio);
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        s_role[0] = x;
        s_role[1] = (int)(__hip_atomic_fetch_add(P.reg + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.reg_base);
    }
    __syncthreads();
    if (s_role[1] < 0 || s_role[1] >= 32) {  // more than 32 workgroups landed on this XCD
        if (threadIdx.x == 0) sa_raise(P.err, 2u);
        return;
    }
    const int sub = s_role[1] / NTU, grp = s_role[0] * (32 / NTU) + sub;
    if (sub >= 32 / NTU) return;
    const int role_x = s_role[1] - sub * NTU, role_z = grp / P.nbt, role_y = grp - role_z * P.nbt + P.bt0;
    if (role_z >= P.n) return;
    if ((P.fault & 1) && role_x == 1 && role_y == 0 && role_z == 0) return;  // injected fault: a group one member short
    const PFwdJob& J = P.j[role_z];
    const int B = P.B;
    const bool stamper = P.stamp && threadIdx.x == 0 && role_x + role_y + role_z == 0;
    if (stamper) P.stamp[0] = wall_clock64();
    float4* red = reinterpret_cast<float4*>(psm);  // [2][source wave 4][register 4][lane 64] {r, z, n, -}
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int u0 = role_x * 16, b0 = role_y * 16;
    const int row = 4 * g + wave;
    const int b = b0 + row, u = u0 + i;
    const bool live = b < B;
    const float e_br = J.b_hh[u], e_bz = J.b_hh[H + u], e_bn = J.b_hh[2 * H + u];
    const int kbeg = wave * KS;
    int budget = P.spin_limit;
    unsigned* errp = P.err;
    const int t0 = J.t0, nsteps = J.nsteps, dt = J.dt, t_first = J.t_first;
    const int s_hx = P.nbt_all * kStep;
    __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)J.hx, 0, 0x7fffffff, 0x00020000);
    const int gather_off = i * 64 + g * 16 + wave * NK * 1024;
    const int tile_off = role_y * kStep;
    const float* p_ai = J.ai + (long)(live ? b : 0) * P.rb * 3 * H + u;
    const long s_ai = (long)P.rt * 3 * H;
    float* dump = P.dump + blockIdx.x * 256 + tid;
    float* p_h = live ? J.h_out + (long)b * J.hs_b + u : dump;
    const long s_h = live ? J.hs_t : 0;
    float* p_st = (STASH && live) ? J.stash + (long)b * P.rb * 5 * H + u : dump;
    const long s_st = (STASH && live) ? (long)P.rt * 5 * H : 0;
    const int so = (STASH && live) ? H : 0;
    const bool odd = (i & 1) != 0;
    const int pub_col = ((role_x & 1) * 16 + (i & ~1)) * 2 + row * 64 + (role_x >> 1) * 1024 + tile_off;
    const int pub_a = pub_col + (odd ? NKT * 1024 : 0), pub_b = pub_col + 2 * NKT * 1024;
    char* hx_own = J.hx;
    float hp = 0.f;
    if (live && t0 != t_first) hp = J.h_out[(long)b * J.hs_b + (long)(t0 - dt) * J.hs_t + u];
    SaBf3 wh[NK][3];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const float* q = J.w_hh + (long)(n * H + u0 + i) * H + kbeg + 32 * kk + 8 * g;
            wh[kk][n] = sa_split8(*reinterpret_cast<const float4*>(q), *reinterpret_cast<const float4*>(q + 4));
        }
    __syncthreads();

    sa_bf16x8 a[NK][3];
    auto poll_until_fresh = [&](int tprev) {
        const int sbase = tprev * s_hx + tile_off;
        for (int spins = 0;; ++spins) {
            asm volatile("" ::: "memory");  // every trip re-issues its loads
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[kk][pl] = __builtin_bit_cast(sa_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(hres, gather_off, sbase + (pl * NKT + kk) * 1024, 16));
            unsigned m = 0u;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const uint4 q = __builtin_bit_cast(uint4, a[kk][pl]);
                    m = max(max(m, q.x), q.y);
                    m = max(max(m, q.z), q.w);
                }
            if (__builtin_amdgcn_ballot_w64(m == kPlaneSentinel) == 0) break;
            if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
        }
    };
    float nx_r, nx_z, nx_n;
    auto fetch_ai = [&](int tt) { const float* q = p_ai + (long)tt * s_ai; nx_r = q[0]; nx_z = q[H]; nx_n = q[2 * H]; };
    auto finish = [&](int s, int t, const f32x4 (&acc)[3], float e_r, float e_z, float e_n) {
        float4* rd = red + (s & 1) * 1024;
#pragma unroll
        for (int r = 0; r < 4; ++r) rd[(wave * 4 + r) * 64 + lane] = make_float4(acc[0][r], acc[1][r], acc[2][r], 0.f);
        __syncthreads();
        const float4 s0 = rd[(0 * 4 + wave) * 64 + lane], s1 = rd[(1 * 4 + wave) * 64 + lane];
        const float4 s2 = rd[(2 * 4 + wave) * 64 + lane], s3 = rd[(3 * 4 + wave) * 64 + lane];
        const float sr = ((s0.x + s1.x) + s2.x) + s3.x, sz = ((s0.y + s1.y) + s2.y) + s3.y, sn = ((s0.z + s1.z) + s2.z) + s3.z;
        const float r = sa_fast_sigmoid(e_r + sr + e_br);
        const float z = sa_fast_sigmoid(e_z + sz + e_bz);
        const float q = sn + e_bn;
        const float n = sa_fast_tanh(e_n + r * q);
        const float h = live ? (1.0f - z) * n + z * hp : 0.f;  // rows beyond the batch publish zeros: a tile has no holes
        {
            const float vn = sa_quad_swap1(h);
            unsigned p1, p2, p3;
            sa_split2(odd ? vn : h, odd ? h : vn, p1, p2, p3);
            char* qx = hx_own + (long)t * s_hx;
            __hip_atomic_store(reinterpret_cast<unsigned*>(qx + pub_a), odd ? p2 : p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned*>(qx + pub_b), p3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        p_h[(long)t * s_h] = h;
        if (STASH) {
            float* st = p_st + (long)t * s_st;
            st[0] = r; st[so] = z; st[2 * so] = n; st[3 * so] = q; st[4 * so] = hp;
        }
        hp = h;
    };
    fetch_ai(t0);
    int s = 0;
    if (t0 == t_first && nsteps > 0) {  // the sequence's first step (peeled: no recurrent term, nobody to wait for)
        f32x4 acc[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float e_r = nx_r, e_z = nx_z, e_n = nx_n;
        fetch_ai(nsteps > 1 ? t0 + dt : t0);
        finish(0, t0, acc, e_r, e_z, e_n);
        s = 1;
    }
    for (; s < nsteps; ++s) {
        const int t = t0 + s * dt;
        f32x4 acc[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float e_r = nx_r, e_z = nx_z, e_n = nx_n;
        sa_wait_vmcnt<STASH ? 6 : 1>();  // pacing of the first polling trip: the step's own publish stores are acknowledged
        poll_until_fresh(t - dt);
        fetch_ai(s + 1 < nsteps ? t + dt : t);  // behind the poll; the last step re-reads its own row: unused
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int o = 0; o < 6; ++o)
#pragma unroll
                for (int n = 0; n < 3; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][SA_X6_PA(o)], wh[kk][n].p[SA_X6_PB(o)], acc[n], 0, 0, 0);
        finish(s, t, acc, e_r, e_z, e_n);
    }
    if (P.stamp && threadIdx.x == 0) atomicMax(P.stamp + 1, (unsigned long long)wall_clock64());  // the LAST block out
}

// Gate-gradient arithmetic shared by the step kernel and the persistent kernel.  Contraction is switched off inside:
// the two kernels inline this into different surroundings and hipcc would otherwise pick different mul+add -> fma
// fusions (observed: 1-ulp differences between the two paths); the explicit fmaf calls pin the fused ones.
__device__ __forceinline__ float gru_bwd_total_dh(float dh_out, float r0, float r1, float r2, float r3, float dh_next,
                                                  float z_next) {
#pragma clang fp contract(off)
    const float acc = ((r0 + r1) + r2) + r3;
    return __builtin_fmaf(dh_next, z_next, dh_out + acc);
}
__device__ __forceinline__ void gru_bwd_gates(float dh, float r, float z, float n, float q, float hp, float& dpr,
                                              float& dpz, float& dpn, float& dqn) {
#pragma clang fp contract(off)
    const float dn = dh * (1.0f - z);
    const float dz = dh * (hp - n);
    dpn = dn * (1.0f - n * n);
    dpr = dpn * q * r * (1.0f - r);
    dpz = dz * z * (1.0f - z);
    dqn = dpn * r;
}

// ----------------------------------------------------------------------------------------------------- backward step
struct BwdJob {
    const float* dh_out;  // gradient wrt h_out, [b * ds_b + t * ds_t + j]
    const float* stash;   // rows x 5H
    const float* w_hh_t;  // (H, 3H): W_hh transposed
    float* dai;           // rows x 3H
    float* dah;           // rows x 3H
    float* dh_ping;       // (B, H) running dh of the step being produced (write)
    const float* dh_pong; // (B, H) running dh of the previously produced step (read)
    long ds_b, ds_t;
    int t;                // step whose gate gradients this launch produces
    int t_next;           // step processed by the previous launch of this job (< 0: none)
};
struct BwdJobs {
    int n, B, H;
    int bt0, tile_rows;
    unsigned long long* stamp;
    long rb, rt;
    BwdJob j[kMaxJobs];
};

__global__ __launch_bounds__(256) void gru_bwd_step_kernel(BwdJobs P) {
    __shared__ float red[4][256];
    const BwdJob& J = P.j[blockIdx.z];
    const int H = P.H, B = P.B, H3 = 3 * P.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int u0 = blockIdx.x * 16, b0 = (blockIdx.y + P.bt0) * P.tile_rows;
    const bool have_next = J.t_next >= 0;
    const bool stamper = P.stamp && threadIdx.x == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0;
    if (stamper) P.stamp[0] = wall_clock64();

    const int bi = threadIdx.x >> 4, uj = threadIdx.x & 15;
    const int b = b0 + bi, u = u0 + uj;
    const bool live = b < B && u < H && bi < P.tile_rows;
    const long row = live ? (long)b * P.rb + (long)J.t * P.rt : 0;
    float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, q = 0.f, hp = 0.f, z_next = 0.f, dh_prev = 0.f;
    if (live) {  // epilogue operands first (see the forward kernel)
        dh = J.dh_out[(long)b * J.ds_b + (long)J.t * J.ds_t + u];
        const float* s = J.stash + row * 5 * H;
        r = s[u]; z = s[H + u]; n = s[2 * H + u]; q = s[3 * H + u]; hp = s[4 * H + u];
        if (have_next) {
            z_next = J.stash[((long)b * P.rb + (long)J.t_next * P.rt) * 5 * H + H + u];
            dh_prev = J.dh_pong[(long)b * H + u];
        }
    }

    if (have_next) {  // dh_t += dah_{t_next} W_hh   (K = 3H)
        const int kslice = ((H3 + 63) / 64) * 16;
        const int kbeg = wave * kslice, kend = min(H3, kbeg + kslice);
        const int brow = min(b0 + i, B - 1);
        const int urow = min(u0 + i, H - 1);
        const float* a_row = J.dah + ((long)brow * P.rb + (long)J.t_next * P.rt) * H3;
        const float* b_rows[1] = {J.w_hh_t + (long)urow * H3};
        f32x4 acc[1], acc2[1];
        acc[0] = acc2[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        smallm_mfma<1, 12, true>(a_row, b_rows, kbeg, kend, H3, acc, g, acc2);  // same order as the persistent kernel
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) red[wave][(g * 4 + rr) * 16 + i] = acc[0][rr] + acc2[0][rr];
    }
    __syncthreads();
    if (!live) { if (stamper) P.stamp[1] = wall_clock64(); return; }
    if (have_next)
        dh = gru_bwd_total_dh(dh, red[0][threadIdx.x], red[1][threadIdx.x], red[2][threadIdx.x], red[3][threadIdx.x],
                              dh_prev, z_next);
    J.dh_ping[(long)b * H + u] = dh;
    float dpr, dpz, dpn, dqn;
    gru_bwd_gates(dh, r, z, n, q, hp, dpr, dpz, dpn, dqn);
    float* di = J.dai + row * H3;
    float* dhh = J.dah + row * H3;
    di[u] = dpr; di[H + u] = dpz; di[2 * H + u] = dpn;
    dhh[u] = dpr; dhh[H + u] = dpz; dhh[2 * H + u] = dqn;
    if (stamper) P.stamp[1] = wall_clock64();
}

// ------------------------------------------------------------------------------------- persistent backward chunk
// The backward counterpart of gru_fwd_persist_kernel, XCD-local groups only (see PFwdJobs): one launch unwinds
// `nsteps` consecutive time steps (descending) of up to 8 layer-jobs.  A block keeps its 16 rows of W_hh^T (16 x 3H)
// in LDS and carries dh / z of the step it just produced in registers; what crosses CUs every step is the (B, 3H)
// gate-gradient row block dah[t+1], published with write-through stores and read back with sc1 loads out of the
// XCD's own L2.
struct PBwdJob {
    const float* dh_out;   // gradient wrt h_out, [b * ds_b + t * ds_t + j]
    const float* stash;    // rows x 5H
    const float* w_hh_t;   // (H, 3H)
    float* dai;            // rows x 3H
    float* dah;            // rows x 3H
    float* dh_state;       // (B, H): running dh carried across launches
    unsigned* counters;    // one per batch tile; monotonic over the whole stack call
    long ds_b, ds_t;
    const float* w_ih_t;   // gru_bwd_fused_kernel: (H, 3H) = this layer's W_ih transposed, or null (bottom layer)
    float* dx_out;         // gru_bwd_fused_kernel: d h_out of the layer below, [b * xs_b + t * xs_t + j], or null
    float* xch;            // gru_bwd_fused_kernel: the exchange buffer, dai tiled [t][batch tile][k / 16][b % 16][k % 16]
    float* dump;           // gru_bwd_fused_kernel: 256 x 256 floats nobody reads (where predicated-off stores go)
    long xs_b, xs_t;
    unsigned dx_drop_stream;  // gru_bwd_fused_kernel: mask stream of dx_out (= the dropped output of the layer below)
    char* gpk;             // gru_bwd_fused_kernel<.., PACKG>: the layer's 4H-row packed gate-gradient operand (gemm_f32.hip)
    float* gsum;           // gru_bwd_fused_kernel<.., PACKG>: its row sums per batch tile, [nbt_all][4H] (the bias gradients)
    char* kpk;             // gru_bwd_fused_kernel<.., PACKK>: dai as the packed A operand of the input-gradient product
                           // (rows = (t, b), k = (gate, unit): T * B rows x 3H)
    int t0, nsteps;        // first time index this launch unwinds, number of steps
    int dt, t_first;       // -1 for a forward-in-time chain (unwinds from T-1), +1 for a reverse chain; the very first index
    unsigned base;
};
struct PBwdJobs {
    int n, B, H, nbt, ntile_u, flagless;
    int bt0, nbt_all;  // see PFwdJobs
    long rb, rt;
    unsigned* reg;
    unsigned reg_base;
    unsigned long long* stamp;
    unsigned* err;
    int spin_limit, fault, prio;  // see PFwdJobs
    int packed;                   // gru_bwd_fused_kernel: 0 = publish with write-through stores, 1 = plain stores, 2 = plain
                                  // stores into a RING of kXRing time slots (default): a published tile is dead once every block
                                  // has published the NEXT step, so its owner re-arms it with the sentinel two steps later --
                                  // the exchange is 4 slots that live in the XCD's L2 instead of T slots that the host pre-fills
                                  // and HBM writes back (0.39 GB per S-LIBRI step)
    SaDrop drop;                  // gru_bwd_fused_kernel: inter-layer dropout -- d h_out[l-1] = mask * (dai[l] W_ih[l])
    int w_rowmajor;               // gru_bwd_fused_kernel: w_hh_t / w_ih_t point at the weights AS STORED ((3H, H) row-major) and
                                  // the block's fragments are read with a stride, once per launch -- no transpose launches
    long pk_kb;                   // gru_bwd_fused_kernel<.., PACKG>: k-tiles of the packed operands, T * B / 16
    int kpk_kb;                   // gru_bwd_fused_kernel<.., PACKK>: k-tiles per row block of the kpk operand (3H / 16, or
                                  // 6H / 16 when the two directions of a layer share one operand, side by side along k)
    unsigned long long* timing;   // debug (SA_GRU_TIMING=1): per block {poll+load, mfma, reduce+barrier, gates+publish} in
                                  // 10 ns ticks and the number of polling trips; else null
    PBwdJob j[kMaxJobs];
};

// POLL = 0: every polling trip re-reads the wave's whole k-slice (24 x 16 B per lane at H = 512) until it holds no
//           sentinel -- the data is its own flag, but a waiting block keeps its CU's vector-memory path saturated
//           (4 waves x 24 KB per trip), which starves a side-stream GEMM block sharing the CU.
// POLL = 1: light trips first -- the four lanes that share a batch row split the producers' 16-column pieces between
//           them (lane g reads float4 #g of the pieces it = g mod 4: a quarter of the bytes, every (row, piece)
//           still probed) -- then ONE full trip, checked in full (a torn piece just sends the wave back to polling).
// SLEEP: s_sleep between failed trips (64-cycle units): yields issue slots and memory bandwidth to the co-resident block.
__global__ __launch_bounds__(256) void gru_bwd_persist_kernel(PBwdJobs P) {
    constexpr int POLL = 0, SLEEP = 0;  // the light-trip / s_sleep polling variants of round 2 (measured 17 % slower) are retired
    extern __shared__ __attribute__((aligned(16))) float psm[];
    __shared__ int s_role[2];
    SA_PERSIST_EXCLUSIVE(P.prio);
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        s_role[0] = x;
        s_role[1] = (int)(__hip_atomic_fetch_add(P.reg + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.reg_base);
    }
    __syncthreads();
    if (s_role[1] < 0 || s_role[1] >= 32) {
        if (threadIdx.x == 0) sa_raise(P.err, 2u);
        return;
    }
    const int sub = s_role[1] / P.ntile_u, grp = s_role[0] * (32 / P.ntile_u) + sub;
    if (sub >= 32 / P.ntile_u) return;
    const int role_x = s_role[1] - sub * P.ntile_u, role_z = grp / P.nbt, role_y = grp - role_z * P.nbt + P.bt0;
    if (role_z >= P.n) return;
    if ((P.fault & 1) && role_x == 1 && role_y == 0 && role_z == 0) return;  // injected fault: a group one member short
    const PBwdJob& J = P.j[role_z];
    const int H = P.H, B = P.B, H3 = 3 * P.H;
    const bool stamper = P.stamp && threadIdx.x == 0 && role_x + role_y + role_z == 0;
    if (stamper) P.stamp[0] = wall_clock64();
    float* red = psm;  // [2][4][256]; the block's 16 rows of W_hh^T live in registers (see below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int u0 = role_x * 16, b0 = role_y * 16;
    const int bi = tid >> 4, uj = tid & 15;
    const int b = b0 + bi, u = u0 + uj;
    const bool live = b < B;
    float dh_run = 0.f, z_next = 0.f;
    if (live && J.t0 != J.t_first) {
        dh_run = J.dh_state[(long)b * H + u];
        z_next = J.stash[((long)b * P.rb + (long)(J.t0 - J.dt) * P.rt) * 5 * H + H + u];
    }
    unsigned* counter = J.counters + role_y;
    int budget = P.spin_limit;  // wave-uniform; 0 after the first timeout: the call is lost, drain quickly
    const int kslice = H3 / 4, kbeg = wave * kslice;  // 3H % 64 == 0 (checked by the host)
    const int brow = min(b0 + i, B - 1);
    __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void*)J.dah, 0, 0x7fffffff, 0x00020000);
    bool dead = false;
    // The W_hh^T fragments a lane feeds to its MFMAs are the same every step (row u0 + i, the wave's k-slice, the
    // lane's k-group: at most 24 x 4 floats at H = 512): loaded once, resident in the register file.
    float4 wr[24];
#pragma unroll
    for (int it = 0; it < 24; ++it) {
        const int k = kbeg + 16 * it + 4 * g;
        wr[it] = 16 * it < kslice ? *reinterpret_cast<const float4*>(J.w_hh_t + (long)(u0 + i) * H3 + k)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;
    const bool timed = P.timing != nullptr && tid == 0;
    if (timed) tprev = wall_clock64();
#define SA_TICK(k) if (timed) { const unsigned long long now = wall_clock64(); tacc[k] += now - tprev; tprev = now; }

    for (int s = 0; s < J.nsteps; ++s) {
        const int t = J.t0 + s * J.dt;
        const bool have_next = t != J.t_first;
        const long row = (long)(live ? b : 0) * P.rb + (long)t * P.rt;
        float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, q = 0.f, hp = 0.f;
        if (live) {
            dh = J.dh_out[(long)b * J.ds_b + (long)t * J.ds_t + u];
            const float* st = J.stash + row * 5 * H;
            r = st[u]; z = st[H + u]; n = st[2 * H + u]; q = st[3 * H + u]; hp = st[4 * H + u];
        }
        if (s > 0 && !P.flagless) {  // every unit tile of this (job, batch tile) must have published dah[t + 1]
            if (tid == 0 && !dead) {
                const unsigned need = J.base + (unsigned)P.ntile_u * (unsigned)s;
                int spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    if (++spins > 4 * P.spin_limit) { dead = true; sa_raise(P.err, 1u); break; }
                }
            }
            __syncthreads();
        }
        if (have_next) {  // dh_t += dah_{t+1} W_hh   (K = 3H), A rows = batch, B rows = this block's 16 units
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};  // even / odd k-groups
            const int abase = (int)((((long)brow * P.rb + (long)(t - J.dt) * P.rt) * H3) * 4);  // byte offset of the row
            for (int kk0 = kbeg; kk0 < kbeg + kslice; kk0 += 384) {  // H = 512: the whole k-slice in ONE round trip
                f32x4v a[24];
                for (int spins = 0;; ++spins) {
                    bool stale = false;
                    if (POLL == 1 && P.flagless) {  // light trips: 6 probes per lane
                        for (;; ++spins) {
                            f32x4v pr[6];
#pragma unroll
                            for (int j = 0; j < 6; ++j) {
                                const int k = kk0 + 16 * (4 * j + g) + 4 * g;
                                pr[j] = kk0 + 16 * (4 * j + g) < kbeg + kslice
                                            ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(dres, abase + 4 * k, 0, 16))
                                            : f32x4v{0.f, 0.f, 0.f, 0.f};
                            }
                            bool st = false;
#pragma unroll
                            for (int j = 0; j < 6; ++j) st |= has_sentinel(pr[j]);
                            if (__builtin_amdgcn_ballot_w64(st) == 0 || spins > budget) break;
                            if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
                        }
                    }
#pragma unroll
                    for (int it = 0; it < 24; ++it) {
                        const int k = kk0 + 16 * it + 4 * g;
                        a[it] = kk0 + 16 * it < kbeg + kslice
                                    ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(dres, abase + 4 * k, 0, 16))
                                    : f32x4v{0.f, 0.f, 0.f, 0.f};
                    }
                    if (!P.flagless) break;
#pragma unroll
                    for (int it = 0; it < 24; ++it) stale |= has_sentinel(a[it]);
                    if (timed) ++tacc[4];
                    if (__builtin_amdgcn_ballot_w64(stale) == 0) break;
                    if (spins > budget) { if (lane == 0) sa_raise(P.err, 1u); budget = 0; break; }
                    if (POLL == 0 && SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
                }
                SA_TICK(0)
#pragma unroll
                for (int it = 0; it < 24; ++it) {
                    if (kk0 + 16 * it < kbeg + kslice) {
                        const float4 w = wr[it];  // H <= 512: the k loop runs once, `it` indexes the resident fragments
                        f32x4& d = (it & 1) ? acc1 : acc;  // two chains: the 40-cycle dependent latency is hidden
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, w.x, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, w.y, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, w.z, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, w.w, d, 0, 0, 0);
                    }
                }
            }
            float* rd = red + (P.flagless ? (s & 1) * 1024 : 0);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) rd[wave * 256 + (g * 4 + rr) * 16 + i] = acc[rr] + acc1[rr];
            SA_TICK(1)
        }
        __syncthreads();
        SA_TICK(2)
        if (live) {
            const float* rd = red + (P.flagless ? (s & 1) * 1024 : 0);
            if (have_next)
                dh = gru_bwd_total_dh(dh, rd[tid], rd[256 + tid], rd[512 + tid], rd[768 + tid], dh_run, z_next);
            float dpr, dpz, dpn, dqn;
            gru_bwd_gates(dh, r, z, n, q, hp, dpr, dpz, dpn, dqn);
            float* di = J.dai + row * H3;
            float* dhh = J.dah + row * H3;
            di[u] = dpr; di[H + u] = dpz; di[2 * H + u] = dpn;
            __hip_atomic_store(dhh + u, dpr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // write-through
            __hip_atomic_store(dhh + H + u, dpz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dhh + 2 * H + u, dqn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dh_run = dh;
            z_next = z;
        }
        SA_TICK(3)
        if (P.flagless) continue;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#undef SA_TICK
    if (timed) {
        unsigned long long* o = P.timing + 5 * ((role_z * P.nbt + role_y - P.bt0) * P.ntile_u + role_x);
        for (int k = 0; k < 5; ++k) atomicAdd(&o[k], tacc[k]);
    }
    if (live) J.dh_state[(long)b * H + u] = dh_run;
    if (P.stamp && threadIdx.x == 0) atomicMax(P.stamp + 1, (unsigned long long)wall_clock64());  // the LAST block out
}

// --------------------------------------------------------------- persistent backward chunk with the input gradient fused
// gru_bwd_persist_kernel plus the product the layer wavefront used to pay one grouped GEMM launch (and its split-K
// reduce) per wave for:   d h_out[l-1][t] = dai[l][t] W_ih[l]   -- the gradient the layer below starts its step t from.
// The block that owns units u0 .. u0+15 of layer l has, every step, the WHOLE gate-gradient row block of its batch tile
// in registers (that is the exchange), so it can also form columns u0 .. u0+15 of that product: 16 more rows of weights
// (W_ih^T) resident in the register file, 24 more MFMAs per wave and step -- issued right after the step's own values
// have been published, i.e. while they travel to the other blocks, where the matrix pipe has nothing else to do.
// Two things change against gru_bwd_persist_kernel to make one gathered row serve both products:
//  * what is exchanged is dai = {dpr, dpz, dpn} (what W_ih multiplies); the recurrent product needs dqn = dpn * r, so a
//    wave multiplies its n-gate fragments by r_{t+1} out of the forward stash (the producer's own dqn, bit for bit);
//    dah is still written -- with plain stores -- for the W_hh weight gradient;
//  * the k-space is dealt out by gate: wave w takes columns w H/4 .. (w+1) H/4 of EACH gate (instead of a contiguous
//    quarter of 3H), so every wave needs the same H/4 r-values per batch row.
// The second product's partial sums cross the waves through their own double-buffered LDS slab and ride on the next
// step's barrier; the chunk's last row is gathered once more after the loop.  H = 64 IPG; flag-less hand-off only.
// FUSE = false: the same kernel without the second product (bidirectional layers, one-layer stacks, SA_GRU_FUSE_DX=0):
// tiled exchange, operands a step ahead, d h_out of the layer below left to a GEMM.
// DROP (with FUSE): inter-layer dropout -- the row of d h_out[l-1] is multiplied by the mask the forward pass applied to
// h_out[l-1] (recomputed from (seed, stream, index), dropout.h) before it is stored.  A template parameter, so that the
// kernel without dropout keeps its branch-free tail.
// PACKG (with FUSE, B a multiple of 16): the weight-gradient products run on split-bf16 PACKED operands (gemm_f32.hip), and
// the layout they want -- operand row = (gate, unit), 16 consecutive k = the 16 rows of a batch tile at one time step, as
// three bf16 planes -- is what a block holds at the end of a step, transposed.  Instead of the six row-major stores (dai,
// dah: 2 x 3H floats per row, re-read and re-laid-out by a pack launch afterwards: 0.26 ms per S-LIBRI step) the block
// stages {dpr, dpz, dpn, dqn} through LDS (5 KB per step parity; read back behind the NEXT step's barrier, so no
// barrier is added), wave w splits gate w -- each thread four values -- and stores 3 x 8 bytes into the layer's 4H-row
// operand; the bias gradients are per-thread running sums, folded over the batch tile after the last step.  dai of the
// bottom layer is still written row-major (the d x product reads it); nothing else is.
// PACKK (with PACKG, bidirectional layers): dai is ALSO written packed the other way round -- operand row = (t, b), 16
// consecutive k = the block's 16 units of one gate -- the A operand of the layer's input-gradient product
// d in = sum over directions of dai W_ih, which runs between this layer's recurrence and the next one's.  Waves 0 .. 2 take
// {dpr, dpz, dpn} out of the same LDS staging; no row-major copy of dai is left.
template <int IPG, bool FUSE, bool DROP = false, bool PACKG = false, bool PACKK = false, int SPEC = 4, bool EARLY_ON = true>
__global__ __launch_bounds__(256) void gru_bwd_fused_kernel(PBwdJobs P) {
    constexpr int NIT = 3 * IPG, H = 64 * IPG, H3 = 3 * H;
    // Three gates are exchanged per step.  Without the second product they are {dpr, dpz, dqn}, what the recurrent product
    // multiplies.  With it they are {dpr, dpz, dpn} (what W_ih multiplies) and a wave forms dqn = dpn * r itself from
    // r_{t+1} of the forward stash (RN): exchanging dqn as a FOURTH tile instead was measured 0.1 ms slower per step
    // (9.04 -> 9.12 ms: eight more loads in the gather, which is on the critical path, against eight row-major loads in the
    // tail, which is not), while dropping the r loads from the kernel without the product gained 0.5 ms on the
    // bidirectional step.
    constexpr bool RN = FUSE;
    constexpr int NG = 3, NITG = NG * IPG, XT = NG * (H / 16);
    constexpr bool HAND = SPEC > 0 && FUSE && PACKG && !PACKK;  // the tail scheduled by hand (see there)
    constexpr bool EARLY = HAND && EARLY_ON;
    extern __shared__ __attribute__((aligned(16))) float psm[];
    __shared__ int s_role[2];
    SA_PERSIST_EXCLUSIVE(P.prio);
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        s_role[0] = x;
        s_role[1] = (int)(__hip_atomic_fetch_add(P.reg + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.reg_base);
    }
    __syncthreads();
    if (s_role[1] < 0 || s_role[1] >= 32) {
        if (threadIdx.x == 0) sa_raise(P.err, 2u);
        return;
    }
    const int sub = s_role[1] / P.ntile_u, grp = s_role[0] * (32 / P.ntile_u) + sub;
    if (sub >= 32 / P.ntile_u) return;
    const int role_x = s_role[1] - sub * P.ntile_u, role_z = grp / P.nbt, role_y = grp - role_z * P.nbt + P.bt0;
    if (role_z >= P.n) return;
    if ((P.fault & 1) && role_x == 1 && role_y == 0 && role_z == 0) return;  // injected fault: a group one member short
    const PBwdJob& J = P.j[role_z];
    const int B = P.B;
    const bool stamper = P.stamp && threadIdx.x == 0 && role_x + role_y + role_z == 0;
    if (stamper) P.stamp[0] = wall_clock64();
    float* red = psm;          // [2][4][256]: the four waves' partial sums of the recurrent product, per step parity
    float* red2 = psm + 2048;  // [2][4][256]: the same for the input-gradient product
    float* pks = psm + 4096;   // PACKG: [2][4 gates][16 units][20]: a step's gate gradients, unit-major (16 used of 20)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int u0 = role_x * 16, b0 = role_y * 16;
    const int bi = tid >> 4, uj = tid & 15;
    const int b = b0 + bi, u = u0 + uj;
    const bool live = b < B;
    const bool fuse = FUSE && J.dx_out != nullptr;
    const bool bottom = !fuse;  // PACKG: the layer whose dai the caller's d x product reads
    int budget = P.spin_limit;  // wave-uniform; 0 after the first timeout: the call is lost, drain quickly
    const int kw = wave * (H / 4) + 4 * g;  // the lane's column offset inside a gate; fragment `it` adds 16 (it % IPG)
    __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void*)J.xch, 0, 0x7fffffff, 0x00020000);
    // Resident in the register file for the whole launch: the lane's fragments of rows u0 + i of W_hh^T and W_ih^T.
    float4 wr[NIT], wx[FUSE ? NIT : 1];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = (it / IPG) * H + kw + 16 * (it % IPG);
        if (P.w_rowmajor) {  // W[k + c][u0 + i], c = 0 .. 3: four strided loads (16 lanes x 4 B contiguous each), once per launch
            const float* wq = J.w_hh_t + (long)k * H + u0 + i;
            wr[it] = make_float4(wq[0], wq[H], wq[2 * H], wq[3 * H]);
            if constexpr (FUSE) {
                const float* xq = fuse ? J.w_ih_t + (long)k * H + u0 + i : wq;
                wx[it] = fuse ? make_float4(xq[0], xq[H], xq[2 * H], xq[3 * H]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
        wr[it] = *reinterpret_cast<const float4*>(J.w_hh_t + (long)(u0 + i) * H3 + k);
        if constexpr (FUSE)
            wx[it] = fuse ? *reinterpret_cast<const float4*>(J.w_ih_t + (long)(u0 + i) * H3 + k)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // Every address inside the loop is a per-thread base (formed here, once) plus t times a scalar stride.  The job's
    // fields are read out of the kernel-argument segment HERE: left alone, hipcc re-loads them inside the loop (an
    // s_load and a wait each -- about 1 us per step, measured) rather than keep them live; SA_KEEP makes a value
    // opaque, so the worst the register allocator can do is park it in a VGPR lane.
#define SA_KEEP(v) asm volatile("" : "+s"(v))
    const int bl = live ? b : 0;
    int t0 = J.t0, dt = J.dt, t_first = J.t_first, nsteps = J.nsteps, packed = P.packed;
    long s_dh = J.ds_t, s_st = (long)P.rt * 5 * H, s_d = (long)P.rt * H3, s_dx = J.xs_t;
    long s_x = (long)P.nbt_all * XT * 256;  // floats of the exchange buffer per time step
    unsigned* errp = P.err;
    SA_KEEP(t0); SA_KEEP(dt); SA_KEEP(t_first); SA_KEEP(nsteps); SA_KEEP(packed);
    SA_KEEP(s_dh); SA_KEEP(s_st); SA_KEEP(s_d); SA_KEEP(s_dx); SA_KEEP(s_x); SA_KEEP(errp);
#undef SA_KEEP
    const float* p_dh = J.dh_out + (long)bl * J.ds_b + u;
    const float* p_st = J.stash + (long)bl * P.rb * 5 * H + u;
    const float* p_rn = J.stash + (long)min(b0 + i, B - 1) * P.rb * 5 * H + kw;  // RN: r of row i, this wave's columns
    float* p_xs = J.xch + ((long)role_y * XT + role_x) * 256 + tid;
    float* p_di = J.dai + (long)bl * P.rb * H3 + u;
    float* p_dhh = J.dah + (long)bl * P.rb * H3 + u;
    float* p_dx = fuse ? J.dx_out + (long)b * J.xs_b + u : nullptr;
    // tile (gate, wave, j) = 16 batch rows x 16 columns, 1 KB contiguous: one load instruction of a wave covers it
    const int a0 = (int)((((long)role_y * XT + wave * IPG) * 256 + i * 16 + 4 * g) * 4);
    __syncthreads();

    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;  // SA_GRU_TIMING=1: {gather, mfma, barrier .. publish, tail, trips}
    const bool timed = P.timing != nullptr && tid == 0;
    if (timed) tprev = wall_clock64();
#define SA_TICK(k) if (timed) { const unsigned long long now = wall_clock64(); tacc[k] += now - tprev; tprev = now; }
    f32x4v a[NITG];  // the gathered row block: rows = the batch tile, this wave's fragments of every exchanged gate
    const bool ring = packed >= 2;
    auto gather_issue = [&](int trow) {
        const int abase = a0 + (ring ? (trow & (kXRing - 1)) : trow) * (int)(s_x * 4);
#pragma unroll
        for (int it = 0; it < NITG; ++it)
            a[it] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(
                                                   dres, abase + 1024 * (it % IPG), (it / IPG) * 4 * IPG * 1024, 16));
    };
    // returns once no fragment holds the sentinel (or the call is lost); `issued`: the first trip is already in flight
    auto gather = [&](int trow, bool issued = false) {
        for (int spins = 0;; ++spins) {
            if (!(issued && spins == 0)) {
                asm volatile("" ::: "memory");  // every trip re-issues its loads (they are loop-invariant to the compiler)
                gather_issue(trow);
            }
            bool stale = false;
#pragma unroll
            for (int it = 0; it < NITG; ++it) stale |= has_sentinel(a[it]);
            if (timed) ++tacc[4];
            if (__builtin_amdgcn_ballot_w64(stale) == 0) break;
            if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
        }
    };
    auto second = [&](int par) {  // a[] W_ih -> this wave's partial sums of d h_out[l-1] (columns u0 .. u0+15)
        // four chains (one per k within a fragment): wherever the scheduler cuts the sequence to slip a memory
        // instruction in, neighbouring MFMAs stay independent
        f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const float4 w = wx[FUSE ? it : 0];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, w.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, w.y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, w.z, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, w.w, c3, 0, 0, 0);
        }
        float* rd = red2 + par * 1024;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) rd[wave * 256 + (g * 4 + rr) * 16 + i] = (c0[rr] + c1[rr]) + (c2[rr] + c3[rr]);
    };
    // Stores that do not apply (a row beyond the batch, no product pending) still execute, aimed at a per-thread dump
    // slot: a BRANCH around a vector-memory instruction makes hipcc wait vmcnt(0) at the join, i.e. here for the
    // store's own acknowledgement -- 1.2 us per step when the d h_out store sat under `if (pending && live)`.
    float* p_dump = J.dump + blockIdx.x * 256 + tid;
    // PACKG: wave w packs gate w; thread (unit pu = lane / 4, quarter pq = lane % 4) takes batch rows 4 pq .. 4 pq + 3 of
    // the tile = k 4 pq .. + 3 of the 16-k tile (t B + b0) / 16 of operand row w H + u0 + pu
    char* p_pk = nullptr;
    long s_pk = 0;
    const float* pk_src = nullptr;
    float gs0 = 0.f, gs1 = 0.f, gs2 = 0.f, gs3 = 0.f;
    if constexpr (PACKG) {
        const int pu = lane >> 2, pq = lane & 3;
        const int prow = wave * H + u0 + pu, prl = prow & 127;
        p_pk = J.gpk + ((size_t)(prow >> 7) * P.pk_kb + role_y) * 12288 + prl * 32 + ((((pq >> 1) ^ (prl >> 3)) & 1) << 4) +
               (pq & 1) * 8;
        s_pk = (long)(B / 16) * 12288;
        pk_src = pks + wave * 320 + pu * 20 + 4 * pq;
    }
    // PACKK: wave w < 3 packs gate w; thread (batch row kr = lane / 4, quarter pq = lane % 4) takes units 4 pq .. 4 pq + 3
    const int kr = lane >> 2, kq = lane & 3;
    const float* kk_src = PACKK ? pks + (wave < 3 ? wave : 0) * 320 + (4 * kq) * 20 + kr : nullptr;
    const size_t kk_tile = PACKK ? ((size_t)(wave < 3 ? wave : 0) * (H / 16) + role_x) * 12288 + (kq & 1) * 8 : 0;
    auto pack_row_k = [&](int par, bool on, int trow) {
        const float* q = kk_src + par * 1280;
        unsigned a0, a1, b0_, b1, c0, c1;
        sa_split2(q[0], q[20], a0, b0_, c0);
        sa_split2(q[40], q[60], a1, b1, c1);
        const int m = (trow < 0 ? 0 : trow) * B + b0 + kr, rl = m & 127;
        const bool go = on && wave < 3;
        char* dst = go ? J.kpk + (size_t)(m >> 7) * P.kpk_kb * 12288 + kk_tile + rl * 32 + ((((kq >> 1) ^ (rl >> 3)) & 1) << 4)
                       : reinterpret_cast<char*>(J.dump + ((blockIdx.x * 256 + tid) & ~1));
        const long pl = go ? 4096 : 0;
        *reinterpret_cast<uint2*>(dst) = make_uint2(a0, a1);
        *reinterpret_cast<uint2*>(dst + pl) = make_uint2(b0_, b1);
        *reinterpret_cast<uint2*>(dst + 2 * pl) = make_uint2(c0, c1);
    };
    auto pack_row = [&](int par, bool on, int trow) {  // the step staged in pks[par]: split, store (behind a barrier)
        const float4 v = *reinterpret_cast<const float4*>(pk_src + par * 1280);
        unsigned a0, a1, b0_, b1, c0, c1;
        sa_split2(v.x, v.y, a0, b0_, c0);
        sa_split2(v.z, v.w, a1, b1, c1);
        char* dst = on ? p_pk + (long)trow * s_pk : reinterpret_cast<char*>(J.dump + ((blockIdx.x * 256 + tid) & ~1));
        const long pl = on ? 4096 : 0;
        *reinterpret_cast<uint2*>(dst) = make_uint2(a0, a1);
        *reinterpret_cast<uint2*>(dst + pl) = make_uint2(b0_, b1);
        *reinterpret_cast<uint2*>(dst + 2 * pl) = make_uint2(c0, c1);
    };
    const SaDrop drop = P.drop;
    const unsigned dx_stream = J.dx_drop_stream;
    const long dx_idx0 = (long)bl * H + u;  // mask index of element (trow, b, u) of the layer below: (trow B + b) H + u
    auto flush2 = [&](int par, bool on, int trow) {  // after a barrier: add the four waves' parts, store the row's tile
        const float* rd = red2 + par * 1024;
        float* dst = on && live ? p_dx + (long)trow * s_dx : p_dump;
        float v = ((rd[tid] + rd[256 + tid]) + rd[512 + tid]) + rd[768 + tid];
        // the layer below fed this layer its DROPPED output (nn.GRU(dropout=p)): the same mask routes the gradient.
        // Off the critical path (the layer below trails by a few steps).
        if constexpr (DROP) v *= sa_drop_factor(drop, dx_stream, (uint64_t)((long)(trow < 0 ? 0 : trow) * B * H + dx_idx0));
        // write-through: in one-launch mode the layer below (another XCD) is waiting for exactly this value
        __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // A step's operands out of memory -- its stashed gates, its d h_out, and r of the row the gather brings -- are
    // fetched during the PREVIOUS step, right after that step's values went out: the vector-memory queue returns in
    // order, so a load still on its way to HBM when the gather is issued would hold the gather's data back.
    // Unconditional loads on clamped indices (a branch around a load costs a vmcnt(0) at the join).
    float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, q = 0.f, hp = 0.f;
    f32x4v rn[RN ? IPG : 1];
    auto fetch = [&](int tt) {
        const int to = tt;
        // d h_out may be a row the layer ABOVE forms in this very launch (one-launch mode, another XCD): agent scope
        dh = __hip_atomic_load(p_dh + (long)to * s_dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float* st = p_st + (long)to * s_st;
        r = st[0]; z = st[H]; n = st[2 * H]; q = st[3 * H]; hp = st[4 * H];
        if constexpr (RN) {
            const int tr = tt != t_first ? tt - dt : tt;  // the first step of the sequence gathers nothing
            const float* rrow = p_rn + (long)tr * s_st;
#pragma unroll
            for (int j = 0; j < IPG; ++j) rn[j] = *reinterpret_cast<const f32x4v*>(rrow + 16 * j);
        }
    };
    float dh_run = 0.f, z_next = 0.f;
    if (live && t0 != t_first) {
        dh_run = J.dh_state[(long)b * H + u];
        z_next = p_st[(long)(t0 - dt) * s_st + H];
    }
    fetch(t0);
    int pend_t = -1;  // time index of the input-gradient row whose partial sums wait in red2[(s - 1) & 1]
    for (int s = 0; s < nsteps; ++s) {
        const int t = t0 + s * dt;
        const bool have_next = t != t_first;
        if (have_next) {  // dh_t += dah_{t+1} W_hh   (K = 3H), A rows = batch, B rows = this block's 16 units
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};  // even / odd fragments
            if constexpr (SPEC > 0) {
                // (r6) The product runs on the fragments AS THEY LAND, and the trip's loads are issued BETWEEN its MFMAs.  A wave
                // issues in order, and handing a 1 KB load to the CU's one address path (64 B / clock, four waves contending)
                // holds it for ~35 ns: 24 of them in a row were 0.8 us during which its matrix pipe stood still before the first
                // fragment was even waited for (probe: issue .. first fragment consumed 1.1 - 1.3 us with an EMPTY queue in front).
                // Now SPEC loads go out up front and one more behind the first MFMA of every fragment; the queue returns in
                // order, so the wait in front of fragment `it` is vmcnt(SPEC - 1).  Whether the trip was complete is known at
                // its end -- a stale one (polling trips per step 1.00 - 1.02) throws its sums away and goes again.  Scheduling
                // barriers pin the order: left alone hipcc hoists every sentinel compare, and with them the wait for the LAST
                // fragment, to the top.  2.66 -> 2.29 ms per S-LIBRI launch (profiles/r06_backward_experiments.txt).
                const int tslot = ring ? ((t - dt) & (kXRing - 1)) : (t - dt);
                const int abase = a0 + tslot * (int)(s_x * 4);
                auto load1 = [&](int it) {
                    a[it] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(
                                                           dres, abase + 1024 * (it % IPG), (it / IPG) * 4 * IPG * 1024, 16));
                    __builtin_amdgcn_sched_barrier(0);
                };
                f32x4 acc2, acc3;  // four chains, one per k of a fragment: neighbouring MFMAs never share a sum
                auto trip = [&](auto PROLOGUE) -> bool {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (decltype(PROLOGUE)::value) {
#pragma unroll
                        for (int it = 0; it < SPEC && it < NIT; ++it) load1(it);
                    }
                    acc = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc; acc2 = acc; acc3 = acc;
                    // every element XOR the sentinel, folded by unsigned minimum (4 v_xor + 2 v_min3 per fragment, in the shadow
                    // of its MFMAs): zero at the end <=> the trip brought a sentinel.  Independent of the MFMAs' results: the
                    // branch behind the last fragment does not wait for the matrix pipe to drain.
                    unsigned fold = ~0u;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        f32x4v f = a[it];
                        {
                            const u32x4v q = __builtin_bit_cast(u32x4v, f) ^ kSentinel;
                            fold = min(min(fold, q.x), q.y);
                            fold = min(min(fold, q.z), q.w);
                        }
                        if (RN && it >= 2 * IPG) f = f * rn[RN ? it - 2 * IPG : 0];  // dqn = dpn * r
                        const float4 w = wr[it];
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, w.x, acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (it + SPEC < NIT) load1(it + SPEC);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, w.y, acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, w.z, acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, w.w, acc3, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc = acc + acc2; acc1 = acc1 + acc3;
                    return fold == 0u;  // some element equals the sentinel
                };
                // EARLY (with the hand-scheduled tail): the first SPEC loads of this trip went out behind the LAST quads of the
                // previous step's tail -- ~0.3 us before its end, ~1.9 us after the publish they look for (the hop is ~1.2) -- so
                // the first fragments are there when the product starts; only a retry issues them here.
                bool prologue = !(EARLY && s > 0);
                for (int spins = 0;; ++spins) {
                    asm volatile("" ::: "memory");
                    const bool stale = prologue ? trip(std::true_type{}) : trip(std::false_type{});
                    prologue = true;
                    if (timed) ++tacc[4];
                    if (__builtin_amdgcn_ballot_w64(stale) == 0) break;
                    if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
                }
                SA_TICK(0)
            } else {
                gather(t - dt);
                SA_TICK(0)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    f32x4v f = a[it];
                    if (RN && it >= 2 * IPG) f = f * rn[RN ? it - 2 * IPG : 0];  // dqn = dpn * r
                    const float4 w = wr[it];
                    f32x4& d = (it & 1) ? acc1 : acc;  // two chains: the 40-cycle dependent latency is hidden
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, w.x, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, w.y, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, w.z, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, w.w, d, 0, 0, 0);
                }
            }
            float* rd = red + (s & 1) * 1024;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) rd[wave * 256 + (g * 4 + rr) * 16 + i] = acc[rr] + acc1[rr];
            SA_TICK(1)
        }
        __syncthreads();
        if constexpr (FUSE || PACKK) {
            // (PACKK, bidirectional layers: the layer above's input-gradient products may still be streaming rows in on the
            // side stream, ends first -- same hand-off)
            // one-launch mode: the row of d h_out fetched a step ago may not have been there yet (the host pre-fills
            // the lower layers' d h_out with the sentinel; the layer above stores it two steps after it ran the same
            // time index, so a lower layer settles a few steps behind the one above and this loop seldom turns)
            for (int spins = 0; __builtin_amdgcn_ballot_w64(live && __builtin_bit_cast(unsigned, dh) == kSentinel) != 0; ++spins) {
                if (__builtin_bit_cast(unsigned, dh) == kSentinel)
                    dh = __hip_atomic_load(p_dh + (long)t * s_dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
            }
        }
        float dpr = 0.f, dpz = 0.f, dpn = 0.f, dqn = 0.f;
        if (live) {
            const float* rd = red + (s & 1) * 1024;
            if (have_next)
                dh = gru_bwd_total_dh(dh, rd[tid], rd[256 + tid], rd[512 + tid], rd[768 + tid], dh_run, z_next);
            gru_bwd_gates(dh, r, z, n, q, hp, dpr, dpz, dpn, dqn);
        }
        {
            // The exchange: this block's three 16 x 16 tiles (rows beyond the batch publish zeros -- a tile has no
            // holes a reader could wait on).  Consecutive threads, consecutive addresses: 1 KB per store instruction.
            float* xp = p_xs + (long)(ring ? (t & (kXRing - 1)) : t) * s_x;
            const float third = FUSE ? dpn : dqn;
            // plain stores: the XCD's own L2 is where the group meets, nothing needs to reach memory (a write-through
            // variant behind `if (!packed)` was dead code since round 3 -- and its join cost a vmcnt(0) here)
            xp[0] = dpr; xp[(H / 16) * 256] = dpz; xp[2 * (H / 16) * 256] = third;
        }
        {
            // ring: this step's gather found every block's tile of time t - dt, so every block had finished gathering time
            // t - 2 dt before it published that: this block's tile of t - 2 dt is dead.  Re-armed here, it is visible before
            // this block publishes again (the next gather's loads return behind this store: one in-order queue), i.e. two
            // publishes before the slot's next use.  (No branch around the stores: the idle case hits the dump slot.)
            const bool dead = ring && have_next && (t - dt) != t_first;
            float* rp = dead ? p_xs + (long)((t - 2 * dt) & (kXRing - 1)) * s_x : p_dump;
            const int q1 = dead ? (H / 16) * 256 : 0, q2 = dead ? 2 * (H / 16) * 256 : 0;
            const float sv = __builtin_bit_cast(float, kSentinel);
            rp[0] = sv; rp[q1] = sv; rp[q2] = sv;
        }
        SA_TICK(2)
        // ---- everything the other blocks wait for is out.  The rest of the step is one branch-free scheduling region:
        // the previous row's input gradient leaves, the row-major copies leave, the next step's operands are
        // requested -- 21 vector-memory instructions per wave whose address processing (~0.8 us per step when issued
        // back to back, measured) is dealt out between the 96 MFMAs of the second product instead.  The product runs
        // unconditionally (bottom layer: zero weights; first step: no row -- the result goes to the dump slot).
        if constexpr (HAND) {
            // (r6) The tail by hand (FUSE, PACKG, not PACKK): the 3 IPG quads of second-product MFMAs with the step's 13 + IPG
            // vector-memory instructions dealt out evenly behind them, one scheduling region each, the LOADS first -- d h_out
            // of the layer above (another XCD wrote it through: the longest trip), the stash row, the r rows -- the stores
            // behind them: the queue returns in order, so whatever is still on its way when the next gather goes out holds
            // that gather's data back.  (hipcc's group barriers dealt out 8 of the 21 and issued the rest in a burst.)
            float* ps = pks + (s & 1) * 1280 + uj * 20 + bi;
            ps[0] = dpr; ps[320] = dpz; ps[640] = dpn; ps[960] = dqn;
            gs0 += dpr; gs1 += dpz; gs2 += dpn; gs3 += dqn;
            dh_run = dh;
            z_next = z;
            const int tn = s + 1 < nsteps ? t + dt : t;
            f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
            auto quad = [&](int it) {
                const float4 w = wx[FUSE ? it : 0];
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, w.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, w.y, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, w.z, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, w.w, c3, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            const float* st = p_st + (long)tn * s_st;
            const float* rrow = p_rn + (long)(tn != t_first ? tn - dt : tn) * s_st;
            constexpr int NRN = RN ? IPG : 0, NOPS = 13 + NRN;
            // what the stores carry, formed in the region in front of the first of them
            float fv = 0.f;
            float* fdst = nullptr;
            unsigned pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0, pc0 = 0, pc1 = 0;
            char* pdst = nullptr;
            long ppl = 0;
            float* di = (live && bottom) ? p_di + (long)t * s_d : p_dump;  // the d x product of layer 0 reads dai row-major
            const int g1 = (live && bottom) ? H : 0, g2 = (live && bottom) ? 2 * H : 0;
            auto op = [&](int k) {  // k is a constant after unrolling
                if (k == 0) dh = __hip_atomic_load(p_dh + (long)tn * s_dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (k == 1) r = st[0];
                else if (k == 2) z = st[H];
                else if (k == 3) n = st[2 * H];
                else if (k == 4) q = st[3 * H];
                else if (k == 5) hp = st[4 * H];
                else if (k < 6 + NRN) rn[RN ? k - 6 : 0] = *reinterpret_cast<const f32x4v*>(rrow + 16 * (k - 6));
                else if (k == 6 + NRN) {  // the previous row's d h_out (flush2)
                    const float* rd = red2 + ((s - 1) & 1) * 1024;
                    fdst = pend_t >= 0 && live ? p_dx + (long)pend_t * s_dx : p_dump;
                    fv = ((rd[tid] + rd[256 + tid]) + rd[512 + tid]) + rd[768 + tid];
                    if constexpr (DROP) fv *= sa_drop_factor(drop, dx_stream, (uint64_t)((long)(pend_t < 0 ? 0 : pend_t) * B * H + dx_idx0));
                    __hip_atomic_store(fdst, fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (k == 7 + NRN) {  // the previous row's packed gate operand (pack_row)
                    const float4 v = *reinterpret_cast<const float4*>(pk_src + ((s - 1) & 1) * 1280);
                    sa_split2(v.x, v.y, pa0, pb0, pc0);
                    sa_split2(v.z, v.w, pa1, pb1, pc1);
                    const bool on = s > 0;
                    pdst = on ? p_pk + (long)(t - dt) * s_pk : reinterpret_cast<char*>(J.dump + ((blockIdx.x * 256 + tid) & ~1));
                    ppl = on ? 4096 : 0;
                    *reinterpret_cast<uint2*>(pdst) = make_uint2(pa0, pa1);
                } else if (k == 8 + NRN) *reinterpret_cast<uint2*>(pdst + ppl) = make_uint2(pb0, pb1);
                else if (k == 9 + NRN) *reinterpret_cast<uint2*>(pdst + 2 * ppl) = make_uint2(pc0, pc1);
                else if (k == 10 + NRN) di[0] = dpr;
                else if (k == 11 + NRN) di[g1] = dpz;
                else if (k == 12 + NRN) di[g2] = dpn;
                __builtin_amdgcn_sched_barrier(0);
            };
            // EARLY: the next trip's first loads (time t: the row this step has just published), into fragments the product is done with
            const int abase_next = a0 + (ring ? (t & (kXRing - 1)) : t) * (int)(s_x * 4);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                quad(it);
#pragma unroll
                for (int k = 0; k < NOPS; ++k)
                    if (k * NIT / NOPS == it) op(k);  // op k behind quad floor(k NIT / NOPS): evenly, in order
                if constexpr (EARLY) {
                    constexpr int NE = SPEC < NIT ? SPEC : NIT;
                    if (it >= NIT - NE) {
                        const int e = it - (NIT - NE);
                        a[e] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(
                                                              dres, abase_next + 1024 * (e % IPG), (e / IPG) * 4 * IPG * 1024, 16));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            float* rd2 = red2 + (s & 1) * 1024;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) rd2[wave * 256 + (g * 4 + rr) * 16 + i] = (c0[rr] + c1[rr]) + (c2[rr] + c3[rr]);
            pend_t = fuse && have_next && s > 0 ? t - dt : -1;
            SA_TICK(3)
            continue;
        }
        if constexpr (FUSE) flush2((s - 1) & 1, pend_t >= 0, pend_t);
        if constexpr (PACKG) {  // the packed operand the weight-gradient products read
            float* ps = pks + (s & 1) * 1280 + uj * 20 + bi;
            ps[0] = dpr; ps[320] = dpz; ps[640] = dpn; ps[960] = dqn;
            gs0 += dpr; gs1 += dpz; gs2 += dpn; gs3 += dqn;
            pack_row((s - 1) & 1, s > 0, t - dt);
            if constexpr (PACKK) {
                pack_row_k((s - 1) & 1, s > 0, t - dt);
            } else {
                float* di = (live && bottom) ? p_di + (long)t * s_d : p_dump;  // the d x product of layer 0 reads dai row-major
                const int g1 = (live && bottom) ? H : 0, g2 = (live && bottom) ? 2 * H : 0;
                di[0] = dpr; di[g1] = dpz; di[g2] = dpn;
            }
            dh_run = dh;
            z_next = z;
        } else {  // the row-major copies the weight-gradient products read
            float* di = live ? p_di + (long)t * s_d : p_dump;
            float* dhh = live ? p_dhh + (long)t * s_d : p_dump;
            const int g1 = live ? H : 0, g2 = live ? 2 * H : 0;
            di[0] = dpr; di[g1] = dpz; di[g2] = dpn;
            dhh[0] = dpr; dhh[g1] = dpz; dhh[g2] = dqn;
            dh_run = dh;
            z_next = z;
        }
        fetch(s + 1 < nsteps ? t + dt : t);  // the next step's operands (the last step re-reads its own: unused)
        if constexpr (FUSE) {
            second(s & 1);
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMA
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // 1 vector-memory instruction
            }
            pend_t = fuse && have_next && s > 0 ? t - dt : -1;  // s = 0: that row belongs to the previous launch
        }
        SA_TICK(3)
    }
#undef SA_TICK
    if (timed) {
        unsigned long long* o = P.timing + 5 * ((role_z * P.nbt + role_y - P.bt0) * P.ntile_u + role_x);
        for (int k = 0; k < 5; ++k) atomicAdd(&o[k], tacc[k]);
    }
    if (FUSE && fuse) {  // the chunk's last row, published by the step that just ended
        const int tl = t0 + (nsteps - 1) * dt;
        gather(tl);
        second(nsteps & 1);
        __syncthreads();
        flush2((nsteps - 1) & 1, pend_t >= 0, pend_t);
        flush2(nsteps & 1, true, tl);
    } else if (PACKG) {
        __syncthreads();
    }
    if constexpr (PACKG) {  // the last step's values are staged (a barrier has passed on either branch above)
        pack_row((nsteps - 1) & 1, nsteps > 0, t0 + (nsteps - 1) * dt);
        if constexpr (PACKK) pack_row_k((nsteps - 1) & 1, nsteps > 0, t0 + (nsteps - 1) * dt);
        __syncthreads();
        // bias gradients: this batch tile's row sums, unit-major in LDS, 16 rows folded by the first 64 threads
        float* ps = pks + uj * 20 + bi;
        ps[0] = gs0; ps[320] = gs1; ps[640] = gs2; ps[960] = gs3;
        __syncthreads();
        if (tid < 64) {
            const float* q = pks + (tid >> 4) * 320 + (tid & 15) * 20;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += q[k];
            J.gsum[(long)role_y * 4 * H + (tid >> 4) * H + u0 + (tid & 15)] = t;
        }
    }
    if (live) J.dh_state[(long)b * H + u] = dh_run;
    if (P.stamp && threadIdx.x == 0) atomicMax(P.stamp + 1, (unsigned long long)wall_clock64());  // the LAST block out
}

// ----------------------------------------------- (r6) the one-launch backward recurrence on 16-byte elements {dpr, dpz, dqn, dpn}
// gru_bwd_fused_kernel<8, FUSE, DROP, PACKG> (H = 512) with another exchange layout.  That kernel publishes three 1 KB tiles per
// step (three 4-byte stores per thread, three more to re-arm) and reads the r rows of the forward stash to form dqn = dpn r
// (eight 16-byte loads per lane and step).  Here a thread publishes its element's FOUR gate gradients -- its own dqn among them --
// with ONE 16-byte store, unit-major: element (unit k, batch row i) of a batch tile at ((k 16) + i) 16 bytes.  A wave's fragment is
// then 1 KB contiguous again -- lane (i, g) holds the four gates of (row i, unit 4 f + g) -- and a 16x16x4 MFMA takes ONE gate of a
// fragment: lane (i, g) supplies A[i][g] = that gate of unit 4 f + g, the weights' lane (n, g) holds W[gate][unit 4 f + g][n].
// Per wave and step: 32 fragments (the same 128 KB per block as 24 + 8 r rows), 3 MFMAs per fragment and product, one publish and
// one re-arm store instead of three each, no r rows, no multiply.  The pipelined trip, the hand-scheduled tail and the early loads
// of the other kernel carry over (see there).  Sums over k in another order than that kernel's: H = 512 with PACKG only.
template <bool DROP, int SPEC = 4>
__global__ __launch_bounds__(256) void gru_bwd_fused16_kernel(PBwdJobs P) {
    constexpr int H = 512, H3 = 3 * H, NF = H / 16;  // NF fragments of 4 units per wave (H / 4 units per wave)
    extern __shared__ __attribute__((aligned(16))) float psm[];
    __shared__ int s_role[2];
    SA_PERSIST_EXCLUSIVE(P.prio);
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        s_role[0] = x;
        s_role[1] = (int)(__hip_atomic_fetch_add(P.reg + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.reg_base);
    }
    __syncthreads();
    if (s_role[1] < 0 || s_role[1] >= 32) {
        if (threadIdx.x == 0) sa_raise(P.err, 2u);
        return;
    }
    const int sub = s_role[1] / P.ntile_u, grp = s_role[0] * (32 / P.ntile_u) + sub;
    if (sub >= 32 / P.ntile_u) return;
    const int role_x = s_role[1] - sub * P.ntile_u, role_z = grp / P.nbt, role_y = grp - role_z * P.nbt + P.bt0;
    if (role_z >= P.n) return;
    if ((P.fault & 1) && role_x == 1 && role_y == 0 && role_z == 0) return;  // injected fault: a group one member short
    const PBwdJob& J = P.j[role_z];
    const int B = P.B;
    if (P.stamp && threadIdx.x == 0 && role_x + role_y + role_z == 0) P.stamp[0] = wall_clock64();
    float* red = psm;          // [2][4][256]: the four waves' partial sums of the recurrent product, per step parity
    float* red2 = psm + 2048;  // [2][4][256]: the same for the input-gradient product
    float* pks = psm + 4096;   // [2][4 gates][16 units][20]: a step's gate gradients, unit-major (16 used of 20)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int u0 = role_x * 16, b0 = role_y * 16;
    const int bi = tid >> 4, uj = tid & 15;
    const int b = b0 + bi, u = u0 + uj;
    const bool live = b < B;
    const bool fuse = J.dx_out != nullptr;
    const bool bottom = !fuse;
    int budget = P.spin_limit;
    __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void*)J.xch, 0, 0x7fffffff, 0x00020000);
    // Resident for the whole launch: W[gate][unit][u0 + i] of both matrices for the lane's unit of every fragment.
    float wr[NF][3], wx[NF][3];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int k = wave * (H / 4) + 4 * f + g;  // the lane's unit of fragment f
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            if (P.w_rowmajor) {
                wr[f][n] = J.w_hh_t[(long)(n * H + k) * H + u0 + i];
                wx[f][n] = fuse ? J.w_ih_t[(long)(n * H + k) * H + u0 + i] : 0.f;
            } else {
                wr[f][n] = J.w_hh_t[(long)(u0 + i) * H3 + n * H + k];
                wx[f][n] = fuse ? J.w_ih_t[(long)(u0 + i) * H3 + n * H + k] : 0.f;
            }
        }
    }
#define SA_KEEP(v) asm volatile("" : "+s"(v))
    const int bl = live ? b : 0;
    int t0 = J.t0, dt = J.dt, t_first = J.t_first, nsteps = J.nsteps;
    long s_dh = J.ds_t, s_st = (long)P.rt * 5 * H, s_d = (long)P.rt * H3, s_dx = J.xs_t;
    long s_x = (long)P.nbt_all * H * 64;  // floats of the exchange ring per time slot: nbt_all x H units x 16 rows x 4 gates
    unsigned* errp = P.err;
    SA_KEEP(t0); SA_KEEP(dt); SA_KEEP(t_first); SA_KEEP(nsteps);
    SA_KEEP(s_dh); SA_KEEP(s_st); SA_KEEP(s_d); SA_KEEP(s_dx); SA_KEEP(s_x); SA_KEEP(errp);
#undef SA_KEEP
    const float* p_dh = J.dh_out + (long)bl * J.ds_b + u;
    const float* p_st = J.stash + (long)bl * P.rb * 5 * H + u;
    float* p_dump = J.dump + blockIdx.x * 256 + (tid & ~3);  // (16-byte dump slots: a quad of lanes shares one)
    float* p_xs = J.xch + (((long)role_y * H + u) * 16 + bi) * 4;
    float* p_di = J.dai + (long)bl * P.rb * H3 + u;
    float* p_dx = fuse ? J.dx_out + (long)bl * J.xs_b + u : nullptr;
    const int a0 = (int)(((((long)role_y * H + wave * (H / 4)) * 16) * 4 + lane * 4) * 4);  // bytes: the wave's first fragment, the lane's element
    __syncthreads();

    f32x4v a[NF];
    auto slot_of = [&](int trow) { return trow & (kXRing - 1); };
    auto load_frag = [&](int f, int abase) {
        a[f] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(dres, abase, f * 1024, 16));
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x4 ar, az, an;  // one chain per gate: neighbouring MFMAs never share a sum
    auto trip = [&](int trow, auto PROLOGUE) -> bool {  // the recurrent product on the fragments as they land; true: stale
        const int abase = a0 + slot_of(trow) * (int)(s_x * 4);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(PROLOGUE)::value) {
#pragma unroll
            for (int f = 0; f < SPEC; ++f) load_frag(f, abase);
        }
        ar = f32x4{0.f, 0.f, 0.f, 0.f}; az = ar; an = ar;
        unsigned fold = ~0u;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const f32x4v v = a[f];
            {
                const u32x4v q = __builtin_bit_cast(u32x4v, v) ^ kSentinel;
                fold = min(min(fold, q.x), q.y);
                fold = min(min(fold, q.z), q.w);
            }
            ar = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, wr[f][0], ar, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (f + SPEC < NF) load_frag(f + SPEC, abase);
            az = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, wr[f][1], az, 0, 0, 0);
            an = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, wr[f][2], an, 0, 0, 0);  // (.z = dqn, the producer's own)
            __builtin_amdgcn_sched_barrier(0);
        }
        return fold == 0u;
    };
    f32x4 cr, cz, cn;
    auto triple = [&](int f) {  // the second product's share of fragment f (.w = dpn)
        cr = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].x, wx[f][0], cr, 0, 0, 0);
        cz = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].y, wx[f][1], cz, 0, 0, 0);
        cn = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f].w, wx[f][2], cn, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto second_done = [&](int par) {
        float* rd2 = red2 + par * 1024;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) rd2[wave * 256 + (g * 4 + rr) * 16 + i] = (cr[rr] + cz[rr]) + cn[rr];
    };
    // the packed gate operand: wave w packs gate w; thread (unit pu, quarter pq) takes batch rows 4 pq .. 4 pq + 3 of the tile
    const int pu = lane >> 2, pq = lane & 3;
    const int prow = wave * H + u0 + pu, prl = prow & 127;
    char* p_pk = J.gpk + ((size_t)(prow >> 7) * P.pk_kb + role_y) * 12288 + prl * 32 + ((((pq >> 1) ^ (prl >> 3)) & 1) << 4) + (pq & 1) * 8;
    const long s_pk = (long)(B / 16) * 12288;
    const float* pk_src = pks + wave * 320 + pu * 20 + 4 * pq;
    char* pk_dump = reinterpret_cast<char*>(J.dump + ((blockIdx.x * 256 + tid) & ~1));
    float gs0 = 0.f, gs1 = 0.f, gs2 = 0.f, gs3 = 0.f;
    const SaDrop drop = P.drop;
    const unsigned dx_stream = J.dx_drop_stream;
    const long dx_idx0 = (long)bl * H + u;
    auto flush_row = [&](int par, int trow) {  // add the four waves' parts of a row of d h_out[l-1], store it write-through
        const float* rd = red2 + par * 1024;
        float* dst = trow >= 0 && live ? p_dx + (long)trow * s_dx : J.dump + blockIdx.x * 256 + tid;
        float v = ((rd[tid] + rd[256 + tid]) + rd[512 + tid]) + rd[768 + tid];
        if constexpr (DROP) v *= sa_drop_factor(drop, dx_stream, (uint64_t)((long)(trow < 0 ? 0 : trow) * B * H + dx_idx0));
        __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, q = 0.f, hp = 0.f;
    float dh_run = 0.f, z_next = 0.f;
    if (live && t0 != t_first) {
        dh_run = J.dh_state[(long)b * H + u];
        z_next = p_st[(long)(t0 - dt) * s_st + H];
    }
    {   // the first step's operands
        dh = __hip_atomic_load(p_dh + (long)t0 * s_dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float* st = p_st + (long)t0 * s_st;
        r = st[0]; z = st[H]; n = st[2 * H]; q = st[3 * H]; hp = st[4 * H];
    }
    int pend_t = -1;
    for (int s = 0; s < nsteps; ++s) {
        const int t = t0 + s * dt;
        const bool have_next = t != t_first;
        if (have_next) {  // dh_t += dah_{t+1} W_hh
            bool prologue = !(s > 0);
            for (int spins = 0;; ++spins) {
                asm volatile("" ::: "memory");
                const bool stale = prologue ? trip(t - dt, std::true_type{}) : trip(t - dt, std::false_type{});
                prologue = true;
                if (__builtin_amdgcn_ballot_w64(stale) == 0) break;
                if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
            }
            float* rd = red + (s & 1) * 1024;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) rd[wave * 256 + (g * 4 + rr) * 16 + i] = (ar[rr] + az[rr]) + an[rr];
        }
        __syncthreads();
        for (int spins = 0; __builtin_amdgcn_ballot_w64(live && __builtin_bit_cast(unsigned, dh) == kSentinel) != 0; ++spins) {
            if (__builtin_bit_cast(unsigned, dh) == kSentinel)
                dh = __hip_atomic_load(p_dh + (long)t * s_dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
        }
        float dpr = 0.f, dpz = 0.f, dpn = 0.f, dqn = 0.f;
        if (live) {
            const float* rd = red + (s & 1) * 1024;
            if (have_next) dh = gru_bwd_total_dh(dh, rd[tid], rd[256 + tid], rd[512 + tid], rd[768 + tid], dh_run, z_next);
            gru_bwd_gates(dh, r, z, n, q, hp, dpr, dpz, dpn, dqn);
        }
        // the exchange: ONE 16-byte element (rows beyond the batch publish zeros -- an element has no holes a reader could wait on)
        *reinterpret_cast<float4*>(p_xs + (long)slot_of(t) * s_x) = make_float4(dpr, dpz, dqn, dpn);
        {   // re-arm the element two steps back (see gru_bwd_fused_kernel); the idle case hits the dump slot
            const bool dead = have_next && (t - dt) != t_first;
            float* rp = dead ? p_xs + (long)slot_of(t - 2 * dt) * s_x : p_dump;
            const float sv = __builtin_bit_cast(float, kSentinel);
            *reinterpret_cast<float4*>(rp) = make_float4(sv, sv, sv, sv);
        }
        // ---- the tail: the second product's 32 triples, the step's other 13 vector-memory instructions dealt out between them (the
        // loads first), the next trip's first loads behind the last triples
        {
            float* ps = pks + (s & 1) * 1280 + uj * 20 + bi;
            ps[0] = dpr; ps[320] = dpz; ps[640] = dpn; ps[960] = dqn;
        }
        gs0 += dpr; gs1 += dpz; gs2 += dpn; gs3 += dqn;
        dh_run = dh;
        z_next = z;
        const int tn = s + 1 < nsteps ? t + dt : t;
        cr = f32x4{0.f, 0.f, 0.f, 0.f}; cz = cr; cn = cr;
        const float* st = p_st + (long)tn * s_st;
        float* di = (live && bottom) ? p_di + (long)t * s_d : J.dump + blockIdx.x * 256 + tid;  // the d x product of layer 0 reads dai row-major
        const int g1 = (live && bottom) ? H : 0, g2 = (live && bottom) ? 2 * H : 0;
        const int abase_next = a0 + slot_of(t) * (int)(s_x * 4);
        unsigned pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0, pc0 = 0, pc1 = 0;
        char* pdst = pk_dump;
        long ppl = 0;
        constexpr int NOPS = 13;
        auto op = [&](int k) {  // k is a constant after unrolling
            if (k == 0) dh = __hip_atomic_load(p_dh + (long)tn * s_dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (k == 1) r = st[0];
            else if (k == 2) z = st[H];
            else if (k == 3) n = st[2 * H];
            else if (k == 4) q = st[3 * H];
            else if (k == 5) hp = st[4 * H];
            else if (k == 6) flush_row((s - 1) & 1, pend_t);
            else if (k == 7) {  // the previous row's packed gate operand
                const float4 v = *reinterpret_cast<const float4*>(pk_src + ((s - 1) & 1) * 1280);
                sa_split2(v.x, v.y, pa0, pb0, pc0);
                sa_split2(v.z, v.w, pa1, pb1, pc1);
                const bool on = s > 0;
                pdst = on ? p_pk + (long)(t - dt) * s_pk : pk_dump;
                ppl = on ? 4096 : 0;
                *reinterpret_cast<uint2*>(pdst) = make_uint2(pa0, pa1);
            } else if (k == 8) *reinterpret_cast<uint2*>(pdst + ppl) = make_uint2(pb0, pb1);
            else if (k == 9) *reinterpret_cast<uint2*>(pdst + 2 * ppl) = make_uint2(pc0, pc1);
            else if (k == 10) di[0] = dpr;
            else if (k == 11) di[g1] = dpz;
            else if (k == 12) di[g2] = dpn;
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            triple(f);
#pragma unroll
            for (int k = 0; k < NOPS; ++k)
                if (k * NF / NOPS == f) op(k);
            if (f >= NF - SPEC) load_frag(f - (NF - SPEC), abase_next);
        }
        second_done(s & 1);
        pend_t = fuse && have_next && s > 0 ? t - dt : -1;
    }
    const int tl = t0 + (nsteps - 1) * dt;
    if (fuse) {  // the chunk's last row, published by the step that just ended
        for (int spins = 0;; ++spins) {
            asm volatile("" ::: "memory");
            const bool stale = trip(tl, std::true_type{});
            if (__builtin_amdgcn_ballot_w64(stale) == 0) break;
            if (spins > budget) { if (lane == 0) sa_raise(errp, 1u); budget = 0; break; }
        }
        cr = f32x4{0.f, 0.f, 0.f, 0.f}; cz = cr; cn = cr;
#pragma unroll
        for (int f = 0; f < NF; ++f) triple(f);
        second_done(nsteps & 1);
        __syncthreads();
        flush_row((nsteps - 1) & 1, pend_t);
        flush_row(nsteps & 1, tl);
    } else {
        __syncthreads();
    }
    if (nsteps > 0) {  // the last step's staged gates (a barrier has passed on either branch above)
        const float4 v = *reinterpret_cast<const float4*>(pk_src + ((nsteps - 1) & 1) * 1280);
        unsigned pa0, pa1, pb0, pb1, pc0, pc1;
        sa_split2(v.x, v.y, pa0, pb0, pc0);
        sa_split2(v.z, v.w, pa1, pb1, pc1);
        char* pdst = p_pk + (long)tl * s_pk;
        *reinterpret_cast<uint2*>(pdst) = make_uint2(pa0, pa1);
        *reinterpret_cast<uint2*>(pdst + 4096) = make_uint2(pb0, pb1);
        *reinterpret_cast<uint2*>(pdst + 8192) = make_uint2(pc0, pc1);
    }
    __syncthreads();
    {   // bias gradients: this batch tile's row sums, unit-major in LDS, 16 rows folded by the first 64 threads
        float* ps = pks + uj * 20 + bi;
        ps[0] = gs0; ps[320] = gs1; ps[640] = gs2; ps[960] = gs3;
    }
    __syncthreads();
    if (tid < 64) {
        const float* qq = pks + (tid >> 4) * 320 + (tid & 15) * 20;
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tsum += qq[k];
        J.gsum[(long)role_y * 4 * H + (tid >> 4) * H + u0 + (tid & 15)] = tsum;
    }
    if (live) J.dh_state[(long)b * H + u] = dh_run;
    if (P.stamp && threadIdx.x == 0) atomicMax(P.stamp + 1, (unsigned long long)wall_clock64());  // the LAST block out
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

// up to 16 same-shape transposes in ONE launch (blockIdx.z selects the matrix): the bidirectional backward pass transposes
// W_hh of every (layer, direction) -- eight 5 - 7 us launches per TIMIT step were 1 % of the step (r6)
struct TransposeList { const float* in[16]; float* out[16]; };
__global__ __launch_bounds__(256) void transpose_many_kernel(TransposeList l, int R, int C) {
    __shared__ float tile[32][33];
    const float* __restrict__ in = l.in[blockIdx.z];
    float* __restrict__ out = l.out[blockIdx.z];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < C) tile[k][tx] = in[(long)(r0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && r0 + tx < R) out[(long)(c0 + k) * R + r0 + tx] = tile[tx][k];
}

// Column sums in two deterministic stages: grid (N/64, R) blocks each reduce a row range of 64 columns into
// part[r][n]; the second kernel adds the R partials in order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, long lda, int M, int N,
                                                             int rows_per, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s0 = 0.f, s1 = 0.f;
    if (n < N) {
        int m = m0 + ty;
        for (; m + 4 < m1; m += 8) {  // two independent accumulators: loads in flight
            s0 += a[(long)m * lda + n];
            s1 += a[(long)(m + 4) * lda + n];
        }
        if (m < m1) s0 += a[(long)m * lda + n];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && n < N) part[(long)blockIdx.y * N + n] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// 16 lanes per column add the R partials in a fixed tree (deterministic), 16 columns per block
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int R, int N,
                                                           float* __restrict__ out, int accumulate) {
    const int n = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int j = threadIdx.x & 15;
    float s = 0.f;
    if (n < N)
        for (int r = j; r < R; r += 16) s += part[(long)r * N + n];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (n < N && j == 0) out[n] = accumulate ? out[n] + s : s;
}

__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ a, long lda,
                                                       const float* __restrict__ b, long ldb, float* __restrict__ y,
                                                       long ldy, int rows, int cols) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * cols) return;
    const int r = (int)(idx / cols), c = (int)(idx % cols);
    y[(long)r * ldy + c] = a[(long)r * lda + c] + b[(long)r * ldb + c];
}

int colsum_parts(int M) {
    int r = (M + 127) / 128;
    return r > 256 ? 256 : (r < 1 ? 1 : r);
}

bool aligned4(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// ------------------------------------------------------------------------------------------------------------ host
// One direction of one layer, batch-major (B, T, .) arrays: the building-block entry points.
extern "C" ctcStatus_t sa_gru_fwd(const float* ai, const float* w_hh, const float* b_hh, float* h_out, long hs_b,
                                  long hs_t, float* stash, int B, int T, int H, int reverse, void* stream_) {
    SA_CLEAR_ERR();
    if (!ai || !w_hh || !b_hh || !h_out || B <= 0 || T <= 0 || H <= 0) return CTC_STATUS_INVALID_VALUE;
    if ((H & 3) || (hs_b & 3) || (hs_t & 3) || !aligned4(h_out) || !aligned4(w_hh))
        return CTC_STATUS_INVALID_VALUE;  // 16-byte fragment loads need 4-float alignment
    hipStream_t stream = (hipStream_t)stream_;
    FwdJobs P;
    P.n = 1; P.B = B; P.H = H; P.rb = T; P.rt = 1; P.bt0 = 0; P.tile_rows = 16; P.stamp = nullptr;
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
    SA_CLEAR_ERR();
    (void)h_out;  // h_{t-1} is read from the stash
    if (!dh_out || !stash || !w_hh || !dai || !dah || !workspace || B <= 0 || T <= 0 || H <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if ((H & 3) || !aligned4(dah) || workspace_bytes < sa_gru_bwd_workspace_bytes(B, T, H))
        return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    float* dh0 = (float*)workspace;
    float* dh1 = dh0 + (size_t)B * H;
    float* w_t = (float*)((char*)workspace + sa_align_up((size_t)2 * B * H * sizeof(float), 256));
    hipLaunchKernelGGL(transpose_kernel, dim3((H + 31) / 32, (3 * H + 31) / 32), dim3(256), 0, stream, w_hh, w_t,
                       3 * H, H);
    BwdJobs P;
    P.n = 1; P.B = B; P.H = H; P.rb = T; P.rt = 1; P.bt0 = 0; P.tile_rows = 16; P.stamp = nullptr;
    BwdJob& J = P.j[0];
    J.dh_out = dh_out; J.stash = stash; J.w_hh_t = w_t; J.dai = dai; J.dah = dah;
    J.ds_b = hs_b; J.ds_t = hs_t;
    dim3 grid((H + 15) / 16, (B + 15) / 16, 1);
    for (int s = 0; s < T; ++s) {
        // forward-in-time layers are unwound from t = T-1 down to 0; reverse layers from t = 0 up to T-1
        J.t = reverse ? s : T - 1 - s;
        J.t_next = s == 0 ? -1 : (reverse ? J.t - 1 : J.t + 1);
        J.dh_ping = (s & 1) ? dh1 : dh0;
        J.dh_pong = (s & 1) ? dh0 : dh1;
        hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(256), 0, stream, P);
    }
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

// ---- side stream and overlap switches (used by both stack passes) ----
struct SideStream {
    static constexpr int kEvents = 64;
    hipStream_t s = nullptr;
    hipEvent_t ev[kEvents];
    int next = 0;
    bool ready = false;
    bool init() {
        if (ready) return true;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
        for (int i = 0; i < kEvents; ++i)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return false;
        ready = true;
        return true;
    }
    // `waiter` will not start work queued after this call before everything queued on `signaller` so far is done
    bool order(hipStream_t signaller, hipStream_t waiter) {
        hipEvent_t e = ev[next];
        next = (next + 1) % kEvents;
        return hipEventRecord(e, signaller) == hipSuccess && hipStreamWaitEvent(waiter, e, 0) == hipSuccess;
    }
};
// One per DEVICE (a process normally drives one GPU -- one rank per GPU, speech_amd/dist.py -- but nothing here assumes
// it): streams, events and the health word belong to the device that was current when they were created.
constexpr int kMaxDevices = 16;
static int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
    return d;
}
SideStream g_side_dev[kMaxDevices];
#define g_side (g_side_dev[current_device()])

// Holds the side stream back until the persistent recurrence launch issued at the same moment on the caller's stream has
// been DISPATCHED: enqueued in front of a batch of XCD-filtered GEMM launches.  The recurrence launch's exit-at-once
// workgroups for the idle XCDs need a free slot there, and once GEMM blocks (hundreds of us each) fill those CUs the
// in-order dispatcher keeps the whole recurrence launch waiting (measured: 0.5 ms per layer).
// Rounds 2-5 guessed the moment with a 40 us spin on the wall clock (side_delay_kernel(4000)) -- an ordering by elapsed
// ticks, fragile across clocks and boxes.  Every workgroup of a recurrence launch registers itself
// in the sync page before anything else (reg[XCC id] += 1: how it learns its role), so "dispatched" is a fact that can be
// read: the launch with index k is in once the eight counters sum to 256 (k + 1).  The wait is bounded (~0.3 ms: if the
// recurrence launch never comes -- the persistent path switched off between the two enqueues -- the products just run).
__global__ void side_wait_dispatched_kernel(const unsigned* __restrict__ reg, unsigned want_total) {
    for (int polls = 0; polls < 256; ++polls) {
        unsigned v[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) v[x] = __hip_atomic_load(reg + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned sum = 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) sum += v[x];
        if (sum >= want_total) return;
        __builtin_amdgcn_s_sleep(32);
    }
}

// uni = the layer wavefront of a unidirectional stack (all 8 XCDs busy: the GEMM blocks share CUs with recurrence
// blocks); bidirectional layers keep half the chip idle.  Measured at S-LIBRI (profiles/r02_overlap_*): beside a GEMM
// block a recurrence launch takes 230-310 us instead of 169 (its MFMAs queue behind the GEMM's 64-cycle ones, its
// exchange loads behind the GEMM's tile loads) and the GEMM runs at a third of its speed -- 15.4 ms per step against
// 12.1 -- so unidirectional stacks run their weight gradients behind the recurrence (that side-stream path is retired).
// Bidirectional layers leave XCDs idle (a layer's groups sit on XCDs 0 .. u-1), and XCD-FILTERED side GEMMs keep off the
// busy ones.  Three things had to hold before that paid (profiles/r02_bidirectional_overlap_trace.txt):
//   * one tile per block, not a persistent tile loop: the dispatcher walks a grid in order, so long-lived GEMM blocks on
//     the idle XCDs keep the NEXT recurrence launch from starting (its exit-at-once blocks for those XCDs find no room);
//   * the recurrence launch must be dispatched BEFORE the GEMM blocks arrive (side_wait_dispatched_kernel);
//   * the filtered GEMM must fit BESIDE a recurrence block (<= 232 registers: the one-stage kernel), or its own surplus
//     blocks on the busy XCDs -- and the launch's completion -- wait for the recurrence to end.
// With all three: bidirectional S-LIBRI 36.3 -> 30.9 ms per step (backward stack 22.2 -> 17.7 ms: the 2.9 ms of weight-
// gradient products of a layer run entirely inside the next layer's 2.96 ms recurrence).  ON for bidirectional stacks;
// SA_GRU_OVERLAP=0 switches it off (A/B runs).
bool overlap_enabled() { return true; }


// A whole-sequence product (input projection / input gradient: tall M, short K) with room for the packed split-bf16
// copies of its operands, never split along K (these calls had no workspace before the packed path existed: the
// summation order of the f32 kernel stays what the bit-identity tests of the recurrence kernels were written against).
// A whole-layer product fills the chip by its output tiles alone at the BASELINE sizes (S-LIBRI: 125 x 12 tiles) and is
// never split along K.  At the sizes the reference SHIPS (examples/*/ctc_config.json: batch 8 - 16, T' ~ 150: 9 x 4 tiles on
// 256 CUs) it is a 36-block launch whose blocks walk the whole reduction alone -- 57 + 90 us for the two input-gradient
// products of a TIMIT layer; there the library's split-K rule (deterministic fold) decides (r6).
static int whole_no_split(int M, int N, int nprob) {
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * nprob;
    return tiles >= 192 ? 1 : 0;
}
static ctcStatus_t gemm_whole(int trans_a, int trans_b, int M, int N, int K, const float* A, long lda, const float* B,
                              long ldb, float beta, float* C, long ldc, const float* bias, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
    SaGemmOpts o;
    o.no_split = whole_no_split(M, N, 1); o.colsum = nullptr; o.xcc_mask = 0; o.tile_counter = nullptr;
    return sa_gemm_f32_group_impl(1, trans_a, trans_b, M, N, K, 1.f, &A, lda, &B, ldb, beta, &C, ldc, &bias, nullptr, ws,
                                  ws_bytes, stream, &o);
}

// ------------------------------------------------------------------------------------------------- the layer stack
// A step launch is a dependent-latency chain that leaves the chip mostly idle (profiles/r01_pmc_gru_step_kernels.txt:
// >60 % of wave cycles in s_waitcnt), so the batch is cut into independent groups of rows ("chains"), each advanced on
// its own HIP stream: their step kernels overlap on the GPU.  Stream 0 is the caller's (it also runs the GEMMs); the
// chains join it once per wavefront wave through events.
constexpr int kMaxChains = 8;

struct Chains {
    int n = 1;
    int tile_rows = 16;
    hipStream_t s[kMaxChains];
    int bt0[kMaxChains], nbt[kMaxChains];
    hipEvent_t ev_main = nullptr, ev_sub[kMaxChains];
    bool ok = true;

    Chains(hipStream_t main, void* const* aux, int n_aux, int B) {
        s[0] = main;
        int want = 1 + (aux ? n_aux : 0);
        if (want > kMaxChains) want = kMaxChains;
        tile_rows = 16;
        while (tile_rows > 4 && (B + tile_rows - 1) / tile_rows < want) tile_rows >>= 1;
        const int tiles = (B + tile_rows - 1) / tile_rows;
        n = want < tiles ? want : tiles;
        for (int c = 1; c < n; ++c) s[c] = (hipStream_t)aux[c - 1];
        for (int c = 0, t = 0; c < n; ++c) {
            const int cnt = tiles / n + (c < tiles % n ? 1 : 0);
            bt0[c] = t; nbt[c] = cnt; t += cnt;
        }
        for (int c = 0; c < kMaxChains; ++c) ev_sub[c] = nullptr;
        if (n > 1) {
            ok = hipEventCreateWithFlags(&ev_main, hipEventDisableTiming) == hipSuccess;
            for (int c = 1; c < n && ok; ++c) ok = hipEventCreateWithFlags(&ev_sub[c], hipEventDisableTiming) == hipSuccess;
        }
    }
    ~Chains() {
        if (ev_main) (void)hipEventDestroy(ev_main);
        for (int c = 0; c < kMaxChains; ++c)
            if (ev_sub[c]) (void)hipEventDestroy(ev_sub[c]);
    }
    void fork() {  // side chains wait for everything enqueued on the main stream so far
        if (n < 2) return;
        (void)hipEventRecord(ev_main, s[0]);
        for (int c = 1; c < n; ++c) (void)hipStreamWaitEvent(s[c], ev_main, 0);
    }
    void join() {  // the main stream waits for the side chains
        for (int c = 1; c < n; ++c) {
            (void)hipEventRecord(ev_sub[c], s[c]);
            (void)hipStreamWaitEvent(s[0], ev_sub[c], 0);
        }
    }
};
// Time-major arrays throughout: x (T, B, I0); h_out[l] (T, B, D*H); ai / stash / dai / dah [l*D+d] (T, B, .).
static size_t stack_ai_bytes(int B, int T, int H) { return sa_align_up((size_t)T * B * 3 * H * sizeof(float), 256); }

// ---- opt-in launch profiler (bench.py's live roofline measurement).  When enabled, thread 0 of block (0,0,0) of
// every step launch of the stack entry points writes the 100 MHz wall clock at kernel entry and exit into a device
// ring -- in the XCD-local persistent / fused kernels, whose blocks finish at different times (a layer-0 block of the
// fused forward is done 0.35 ms before the top layer's), the exit word is the atomic MAX over every block's exit, so the
// duration is the launch's, not one block's (the ring is zeroed whenever it is re-armed);
// sa_gru_profile_read() copies the ring back and averages (a) entry-to-exit = kernel time and
// (b) entry-to-entry of consecutive full-width launches = the per-launch interval including the dispatch gap.
// HIP events around single launches perturb the stream by several microseconds each, device stamps do not.
// This is the library's only process-global state and is off by default.
namespace {
struct StepProfiler {
    static constexpr int kRing = 1 << 14;
    bool on = false;
    unsigned long long* ring[2] = {nullptr, nullptr};   // device: [kRing][2] per kind
    unsigned char* full[2] = {nullptr, nullptr};        // host: launch was full width (L jobs)
    long count[2] = {0, 0};
    int steps[2] = {1, 1};      // time steps covered by each launch recorded since the last read (1: step kernels)
    // flags: bit 0 = full width (L jobs), bit 1 = directly follows another step launch (no GEMM in between)
    unsigned long long* slot(int kind, bool full_width, bool follows_step, int nsteps = 1) {
        if (!on || count[kind] >= kRing) return nullptr;
        full[kind][count[kind]] = (full_width ? 1 : 0) | (follows_step ? 2 : 0);
        if (full_width) steps[kind] = nsteps;
        return ring[kind] + 2 * (count[kind]++);
    }
};

// Outcome of the XCD-local persistent kernels (the library's other piece of process-global state).
//   * `dev` is a STICKY device error word: a kernel whose placement assumption or hand-off fails ORs its code in, and
//     nothing but sa_gru_persist_reset() ever clears it -- a failure cannot be overwritten by a later, healthy launch.
//   * the optimiser gates on it ON THE DEVICE: sa_gru_health_flag() turns the word into a float the caller appends to
//     its gradient message, sa_clip_sgd_step() skips the update when that float is non-zero (and, summed by the
//     gradient all-reduce, when ANY rank's is).  No host round trip sits between a failure and the update it must stop.
//   * the host learns about it without a sync and without a copy: the failing block also ORs its code into a word of
//     mapped host memory (sa_raise); every stack call looks at that word, sa_gru_persist_status() first waits for the
//     device to drain.
struct PersistHealth {
    unsigned* dev = nullptr;            // [0] the sticky word, [1] spare, [2 .. 3] the device address of `host` (sa_raise, common.h)
    volatile unsigned* host = nullptr;  // one word of mapped, coherent host memory: failing blocks OR their code in
    unsigned code = 0;      // OR of every failure code seen since the last reset
    bool disabled = false;  // a failure was seen at some point: the persistent path stays off for the process
    bool ready = false;
    bool init() {
        if (ready) return true;
        if (hipMalloc((void**)&dev, 4 * sizeof(unsigned)) != hipSuccess) { dev = nullptr; return false; }
        if (hipMemset(dev, 0, 4 * sizeof(unsigned)) != hipSuccess) return false;
        void* h = nullptr;
        if (hipHostMalloc(&h, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return false;
        host = (volatile unsigned*)h;
        host[0] = 0;
        void* hd = nullptr;
        if (hipHostGetDevicePointer(&hd, h, 0) != hipSuccess) return false;
        if (hipMemcpy(dev + 2, &hd, sizeof(void*), hipMemcpyHostToDevice) != hipSuccess) return false;
        ready = true;
        return true;
    }
    // (rounds 1-3 queued a copy of the word here; the kernels report to the host word themselves now)
    void submit(hipStream_t) { (void)init(); }
    // 0 = fine (or nothing has been reported yet); wait: everything queued on the device so far has run
    unsigned poll(bool wait) {
        if (!ready) return 0;
        if (wait) (void)hipDeviceSynchronize();
        const unsigned c = host[0];
        if (c) { code |= c; disabled = true; }
        return code;
    }
    void reset() {
        if (!ready) return;
        (void)poll(true);
        (void)hipMemset(dev, 0, 2 * sizeof(unsigned));
        host[0] = 0;
        code = 0;  // `disabled` stays
    }
};
PersistHealth g_health_dev[kMaxDevices];  // per device: see g_side_dev
#define g_health (g_health_dev[current_device()])
StepProfiler g_prof;                       // bench.py's opt-in profiler: one device at a time (configure() binds it)
// Serialises the stack entry points per device: they enqueue on library-owned objects (side stream, event rings, the
// health ring).  Launches are asynchronous, so the lock is held for microseconds of host time; two threads driving the
// SAME device through sa_gru_stack_* are safe, two devices never contend.
std::mutex g_stack_mutex[kMaxDevices];
}  // namespace

extern "C" void sa_gru_profile_configure(int enable) {
    if (enable && !g_prof.ring[0]) {
        for (int k = 0; k < 2; ++k) {
            if (hipMalloc((void**)&g_prof.ring[k], sizeof(unsigned long long) * 2 * StepProfiler::kRing) != hipSuccess)
                return;
            g_prof.full[k] = (unsigned char*)malloc(StepProfiler::kRing);
        }
    }
    g_prof.on = enable != 0 && g_prof.ring[0] && g_prof.ring[1];
    g_prof.count[0] = g_prof.count[1] = 0;
    if (g_prof.on)   // exit words are atomic maxima: start from zero
        for (int k = 0; k < 2; ++k) (void)hipMemset(g_prof.ring[k], 0, sizeof(unsigned long long) * 2 * StepProfiler::kRing);
}

// kind 0 = forward step kernel, 1 = backward.  Returns the number of launches averaged.
extern "C" int sa_gru_profile_read(int kind, float* avg_interval_us, float* avg_kernel_us) {
    if (kind < 0 || kind > 1 || !avg_interval_us || !avg_kernel_us || !g_prof.ring[kind]) return 0;
    const long n = g_prof.count[kind];
    g_prof.count[kind] = 0;
    if (n < 2) return 0;
    unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * 2 * n);
    if (hipMemcpy(h, g_prof.ring[kind], sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) {
        free(h);
        return 0;
    }
    (void)hipMemset(g_prof.ring[kind], 0, sizeof(unsigned long long) * 2 * n);  // the slots are handed out again
    double ti = 0.0, tk = 0.0;
    long ni = 0, nk = 0;
    for (long i = 0; i < n; ++i) {
        if (!(g_prof.full[kind][i] & 1)) continue;
        tk += (double)(h[2 * i + 1] - h[2 * i]); ++nk;
        if (i + 1 < n && g_prof.full[kind][i + 1] == 3) { ti += (double)(h[2 * i + 2] - h[2 * i]); ++ni; }
    }
    free(h);
    g_prof.steps[kind] = g_prof.steps[kind] > 0 ? g_prof.steps[kind] : 1;
    *avg_kernel_us = nk ? (float)(tk / nk * 0.01) : 0.f;      // 100 MHz ticks -> us
    *avg_interval_us = ni ? (float)(ti / ni * 0.01) : 0.f;
    return (int)nk;
}

extern "C" int sa_gru_profile_steps_per_launch(int kind) { return kind == 0 || kind == 1 ? g_prof.steps[kind] : 0; }

extern "C" int sa_gru_persist_status(void) { return (int)g_health.poll(true); }

extern "C" int sa_gru_persist_reset(void) {
    const unsigned c = g_health.poll(true);
    g_health.reset();
    return (int)c;
}

namespace {
__global__ void health_flag_kernel(const unsigned* __restrict__ err, float* __restrict__ out) {
    out[0] = (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1.0f : 0.0f;
}
}  // namespace

extern "C" ctcStatus_t sa_gru_health_flag(float* d_flag, void* stream_) {
    SA_CLEAR_ERR();
    if (!d_flag) return CTC_STATUS_INVALID_VALUE;
    if (!g_health.init()) return CTC_STATUS_MEMOPS_FAILED;
    hipLaunchKernelGGL(health_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, (const unsigned*)g_health.dev, d_flag);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

constexpr size_t kSyncBytes = 32768;  // hand-off counters of the persistent kernels (+ an error word)
constexpr size_t kFwdDumpBytes = (size_t)256 * 256 * sizeof(float);  // gru_fwd_fused_kernel: where predicated-off stores go

static int device_cus() {
    static int cus_of[kMaxDevices] = {0};
    int& cus = cus_of[current_device()];
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            cus = n;
        if (cus <= 0) cus = 1;
    }
    return cus;
}

// (Round 1's chip-wide persistent groups -- SA_GRU_PERSIST=1: one sync group spanning XCDs, an in-kernel step of ~10 us
// against a 9 us kernel boundary -- are retired; the finding is in DESIGN.md 3.3.)
static int persist_mode() {  // 2: XCD-local groups (default where eligible); 0: off (SA_GRU_PERSIST=0, or after a failure)
    const long e = sa_opt(SA_OPT_GRU_PERSIST);  // -1 auto; 0 off; 2 arrival counters; 3 flag-less (= auto)
    const int m = e < 0 ? 2 : ((e == 2 || e == 3) ? 2 : 0);
    return (m == 2 && g_health.disabled) ? 0 : m;
}
static bool flagless_mode() {  // the flag-less (sentinel) hand-off is the default; SA_GRU_PERSIST=2 keeps the counters
    const long e = sa_opt(SA_OPT_GRU_PERSIST);
    return e < 0 || e == 3;
}
static int spin_limit() {
    const long v = sa_opt(SA_OPT_GRU_SPIN_LIMIT);
    return v > 0 ? (int)v : (1 << 20);
}
static int fault_injection() {  // tests only: SA_GRU_FAULT=1 makes one workgroup of every persistent launch leave early
    return sa_opt(SA_OPT_GRU_FAULT) == 1 ? 1 : 0;
}
static bool sentinel_fill(float* p, size_t n, hipStream_t stream) {
    return hipMemsetD32Async((hipDeviceptr_t)p, (int)kSentinel, n, stream) == hipSuccess;
}
// The pre-fills of one stack call as ONE launch (round 4): the one-launch forward issues five fills (the layers' h_out with the
// sentinel, the sync words with 0), the one-launch backward eight -- 35 + 45 us of 5 - 8 us launches for 130 + 98 MB; one
// kernel writes the same bytes at the same rate without the seven launch tails in between.
struct FillList { unsigned* p[8]; unsigned long long n[8]; unsigned v[8]; int count; };
__global__ __launch_bounds__(256) void fill_many_kernel(FillList f) {
    const size_t stride = (size_t)gridDim.x * 256, first = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < f.count; ++k) {
        const unsigned v = f.v[k];
        const uint4 v4 = make_uint4(v, v, v, v);
        uint4* d4 = reinterpret_cast<uint4*>(f.p[k]);   // 16-byte aligned (workspace offsets, torch allocations)
        const size_t n4 = f.n[k] >> 2;
        for (size_t i = first; i < n4; i += stride) d4[i] = v4;
        if (first < (f.n[k] & 3)) f.p[k][4 * n4 + first] = v;
    }
}
struct FillBatch {
    FillList f;
    hipStream_t stream;
    bool batched, ok = true;
    FillBatch(hipStream_t s, bool on) : stream(s), batched(on) { f.count = 0; }
    void add(void* p, size_t nwords, unsigned value) {
        if (!batched || ((uintptr_t)p & 15) != 0) {
            ok = ok && hipMemsetD32Async((hipDeviceptr_t)p, (int)value, nwords, stream) == hipSuccess;
            return;
        }
        if (f.count == 8) flush();
        f.p[f.count] = (unsigned*)p; f.n[f.count] = nwords; f.v[f.count] = value; ++f.count;
    }
    void flush() {
        if (f.count == 0) return;
        hipLaunchKernelGGL(fill_many_kernel, dim3(2048), dim3(256), 0, stream, f);
        ok = ok && hipGetLastError() == hipSuccess;
        f.count = 0;
    }
};

// shapes the XCD-local persistent kernels take: 8 XCDs x 32 CUs; a sync group = one (concurrent job, batch tile) with
// H / 16 = 32, 16 or 8 unit tiles, so an XCD hosts 1, 2 or 4 groups.  jobs = layers in flight (unidirectional layer
// wavefront) or the 2 directions of one bidirectional layer.
constexpr int kSyncErr = 200, kSyncReg = 201;  // word offsets in the sync page (counters occupy [0, 200))
constexpr int kSyncTiles = 210, kSyncTileWords = 46;  // tile counters of XCD-filtered GEMM launches: words [210, 256)
static bool xcd_shape_ok(int jobs, int B, int H) {
    const int ntile_u = H / 16;
    // H: a multiple of 64 (a wave's k-slice H / 4 in whole 16-wide fragments) from 128 to 512 (the weight fragments of a
    // block's 16 units live in its register file)
    if (persist_mode() != 2 || H < 128 || H > 512 || (H % 64) || device_cus() != 256 || !g_health.init()) return false;
    (void)B;
    return jobs <= 8 * (32 / ntile_u);  // at least one batch tile per pass (see tiles_per_pass)
}
// batch tiles one persistent launch can host next to `jobs` concurrent jobs: 8 XCDs x (32 / ntile_u) groups
// (r6) groups of 16 units an XCD's 32 CUs host.  Only meaningful for a shape xcd_shape_ok() takes, but the bidirectional
// passes evaluate it BEFORE they know that: H < 16 (H / 16 = 0) and H > 512 (32 / 33 = 0 as a divisor further down) were an
// integer division by zero on the host -- a bidirectional stack of width 4, 8 or 528+ killed the process with SIGFPE instead
// of taking the step kernels (found by tools/s2s_shape_sweep.py).
static int groups_per_xcd(int H) {
    const int nt = H / 16;
    return nt >= 1 && nt <= 32 ? 32 / nt : 1;
}
static int tiles_per_pass(int jobs, int H) { return (8 * groups_per_xcd(H)) / (jobs > 0 ? jobs : 1); }
// XCD-local kernels are one-per-CU through their register reservation (SA_PERSIST_EXCLUSIVE), so they ask for the LDS
// they use and nothing more (an XCD-filtered side-stream GEMM block fits beside them)
static size_t xcd_lds(size_t need) { return need; }
typedef void (*BwdPersistFn)(PBwdJobs);
static BwdPersistFn bwd_persist_fn() { return gru_bwd_persist_kernel; }
static int fwd_chunks(int T) {  // bidirectional forward: time chunks per layer for the projection / recurrence overlap
    const int v = (int)sa_opt(SA_OPT_GRU_FWD_CHUNKS);
    // measured (S-LIBRI T' = 498 / TIMIT T' = 144, ms per step): 1 chunk 30.9 / 6.26, 2: 29.5 / 6.15, 4: 29.5 / 6.27,
    // 6: 29.7 / 6.55, 8: 32.0 / 6.78 -- a chunk launch costs a ramp and an event
    return v > 0 ? (v > 15 ? 15 : v) : (T >= 256 ? 4 : 2);
}
static bool fuse_dx_enabled() {  // SA_GRU_FUSE_DX=0: the per-wave grouped GEMM computes d h_out of the lower layers
    return sa_opt(SA_OPT_GRU_FUSE_DX) != 0;
}
static bool fill_batch_enabled() { return true; }  // (one hipMemsetD32Async per buffer was rounds 1-3)
static bool xring_enabled() { return true; }  // (the backward exchange as T pre-filled time slots was rounds 2-3)
static bool tiled_enabled() {  // SA_GRU_TILED=0: the round-1 recurrence kernels (row-major exchange; bit-identical to the step kernels)
    return sa_opt(SA_OPT_GRU_TILED) != 0;
}
static BwdPersistFn bwd_fused_fn(int H, bool fuse, bool drop = false, bool packg = false, bool allow16 = false) {
    if (!tiled_enabled()) return nullptr;
    if (packg && !fuse) {  // bidirectional layers: gate operands and the input-gradient operand
        if (H == 512) return gru_bwd_fused_kernel<8, false, false, true, true>;
        if (H == 256) return gru_bwd_fused_kernel<4, false, false, true, true>;
        return nullptr;
    }
    if (packg && fuse) {
        // (r6) every instance gathers pipelined (SPEC = 4 loads in flight; with PACKG the tail is scheduled by hand and carries the next
        // trip's first loads); gru.exp bit 4 = the round-5 form of the S-LIBRI instance, bit 5 = without the early loads: A/B
        // (r6) H = 512: the exchange on 16-byte elements {dpr, dpz, dqn, dpn} (gru_bwd_fused16_kernel); gru.exp bit 7 = three tiles: A/B
        if (H == 512 && allow16 && !(sa_opt(SA_OPT_GRU_EXP) & 128)) return drop ? gru_bwd_fused16_kernel<true> : gru_bwd_fused16_kernel<false>;
        if (H == 512 && !drop && (sa_opt(SA_OPT_GRU_EXP) & 16)) return gru_bwd_fused_kernel<8, true, false, true, false, 0>;
        if (H == 512 && !drop && (sa_opt(SA_OPT_GRU_EXP) & 32)) return gru_bwd_fused_kernel<8, true, false, true, false, 4, false>;
        if (H == 512) return drop ? gru_bwd_fused_kernel<8, true, true, true> : gru_bwd_fused_kernel<8, true, false, true>;
        if (H == 256) return drop ? gru_bwd_fused_kernel<4, true, true, true> : gru_bwd_fused_kernel<4, true, false, true>;
        if (H == 128) return drop ? gru_bwd_fused_kernel<2, true, true, true> : gru_bwd_fused_kernel<2, true, false, true>;
    }
#define SA_BWD_FUSED(IPG) \
    if (H == 64 * IPG) return fuse ? (drop ? gru_bwd_fused_kernel<IPG, true, true> : gru_bwd_fused_kernel<IPG, true>) : gru_bwd_fused_kernel<IPG, false>;
    SA_BWD_FUSED(8) SA_BWD_FUSED(4) SA_BWD_FUSED(7) SA_BWD_FUSED(6) SA_BWD_FUSED(5) SA_BWD_FUSED(3)
#undef SA_BWD_FUSED
    return nullptr;
}
// gru_bwd_fused16_kernel exchanges FOUR values per element: its ring holds 4 H floats per batch row and slot
static bool bwd_fused_is16(BwdPersistFn fn) { return fn == gru_bwd_fused16_kernel<true> || fn == gru_bwd_fused16_kernel<false>; }
// the backward kernel has a form that packs the weight gradients' operands itself (PACKG / PACKK) for these widths
static bool packg_available(int H, bool fuse) { return H == 512 || H == 256 || (fuse && H == 128); }
typedef void (*FusedFwdFn)(PFusedFwd);
// gru_fwd_fused_kernel<IPG, POLL_AT, STASH, DROP, TIMED> for H = 64 IPG (null: no such instance)
template <int IPG, int POLL_AT>
static FusedFwdFn fused_fwd_pick(bool stash, bool drop, bool timed) {
    if (timed && !drop)  // (the phase clocks exist without dropout only: with it the call runs un-instrumented)
        return stash ? gru_fwd_fused_kernel<IPG, POLL_AT, true, false, true> : gru_fwd_fused_kernel<IPG, POLL_AT, false, false, true>;
    if (stash) return drop ? gru_fwd_fused_kernel<IPG, POLL_AT, true, true, false> : gru_fwd_fused_kernel<IPG, POLL_AT, true, false, false>;
    return drop ? gru_fwd_fused_kernel<IPG, POLL_AT, false, true, false> : gru_fwd_fused_kernel<IPG, POLL_AT, false, false, false>;
}
static FusedFwdFn fused_fwd_fn(int H, bool stash, bool drop, bool timed) {
    switch (H / 64) {
        case 8: return fused_fwd_pick<8, 4>(stash, drop, timed);
        case 7: return fused_fwd_pick<7, 4>(stash, drop, timed);
        case 6: return fused_fwd_pick<6, 4>(stash, drop, timed);
        case 5: return fused_fwd_pick<5, 4>(stash, drop, timed);
        case 4: return fused_fwd_pick<4, 4>(stash, drop, timed);
        case 3: return fused_fwd_pick<3, 4>(stash, drop, timed);
        case 2: return fused_fwd_pick<2, 4>(stash, drop, timed);
    }
    return nullptr;
}
// gru_fwd_planes_kernel<IPG, POLL_AT, STASH, DROP, TIMED> (even IPG only; null: no such instance)
template <int IPG, int POLL_AT>
static FusedFwdFn planes_fwd_pick(bool stash, bool drop, bool timed) {
    if (timed && !drop)
        return stash ? gru_fwd_planes_kernel<IPG, POLL_AT, true, false, true> : gru_fwd_planes_kernel<IPG, POLL_AT, false, false, true>;
    if (stash) return drop ? gru_fwd_planes_kernel<IPG, POLL_AT, true, true, false> : gru_fwd_planes_kernel<IPG, POLL_AT, true, false, false>;
    return drop ? gru_fwd_planes_kernel<IPG, POLL_AT, false, true, false> : gru_fwd_planes_kernel<IPG, POLL_AT, false, false, false>;
}
static bool planes_fwd_shape(int H) { return H == 128 || H == 256 || H == 384 || H == 512; }
static FusedFwdFn planes_fwd_fn(int H, bool stash, bool drop, bool timed) {
    if (H == 512) {  // A/B (profiles/r06_forward_planes_experiments.txt): the first polling trip inside the input product
        switch (sa_opt(SA_OPT_GRU_EXP) & 7) {
            case 1: return planes_fwd_pick<8, 4>(stash, drop, timed);
            case 2: return planes_fwd_pick<8, 6>(stash, drop, timed);
        }
    }
    switch (H / 64) {
        case 8: return planes_fwd_pick<8, 8>(stash, drop, timed);
        case 6: return planes_fwd_pick<6, 8>(stash, drop, timed);
        case 4: return planes_fwd_pick<4, 8>(stash, drop, timed);
        case 2: return planes_fwd_pick<2, 8>(stash, drop, timed);
    }
    return nullptr;
}
// bytes of ONE layer's bf16-planes exchange buffer (gru_fwd_planes_kernel); 0: the shape does not run that kernel
static int planes_fwd_buffers(int L, int D) { return D == 1 ? 2 * L - 1 : 2; }  // bidirectional: one per direction, per layer in turn
static size_t planes_fwd_bytes(int L, int D, int B, int T, int H) {
    if (!planes_fwd_shape(H) || L > kMaxJobs) return 0;
    const size_t n = (size_t)T * ((B + 15) / 16) * hx_step_bytes(H);
    return n < 0x7fffffffull ? sa_align_up(n, 256) : 0;
}
typedef void (*FwdChunkFn)(PFwdJobs);
// gru_fwd_chunk_kernel<IPG, STASH> for H = 64 IPG (the flag-less XCD-local persistent forward; null: no such instance)
static FwdChunkFn fwd_chunk_fn(int H, bool stash) {
    switch (H / 64) {
#define SA_FWD_CHUNK(I_) case I_: return stash ? gru_fwd_chunk_kernel<I_, true> : gru_fwd_chunk_kernel<I_, false>;
        SA_FWD_CHUNK(8) SA_FWD_CHUNK(7) SA_FWD_CHUNK(6) SA_FWD_CHUNK(5) SA_FWD_CHUNK(4) SA_FWD_CHUNK(3) SA_FWD_CHUNK(2)
#undef SA_FWD_CHUNK
    }
    return nullptr;
}
static FwdChunkFn fwd_chunk_planes_fn(int H, bool stash) {
    switch (H / 64) {
#define SA_FWD_CHUNKP(I_) case I_: return stash ? gru_fwd_chunk_planes_kernel<I_, true> : gru_fwd_chunk_planes_kernel<I_, false>;
        SA_FWD_CHUNKP(8) SA_FWD_CHUNKP(6) SA_FWD_CHUNKP(4) SA_FWD_CHUNKP(2)
#undef SA_FWD_CHUNKP
    }
    return nullptr;
}
static int persist_prio() { return 1; }  // the recurrence waves issue at raised priority (s_setprio 3)

static int clamp_chunk(int chunk, int T) {
    // measured on MI355X at S-LIBRI, whole train step: step kernels 16 -> 19.6 ms, 32 -> 19.9, 8 -> 20.5;
    // persistent chunk kernels (a wave of the layer wavefront costs one launch + one grouped GEMM whatever its
    // length) 8 -> 18.3, 16 -> 16.4, 24 -> 16.0, 28 -> 15.7, 32 -> 15.8, 40 -> 15.7, 48 -> 15.8, 64 -> 16.5
    if (chunk <= 0) chunk = 16;
    return chunk > T ? T : chunk;
}

// Default chunk of the persistent path: among 24 .. 40 steps, the length whose grouped projection GEMM
// ((L-1) problems of (chunk * B) x 3H, 128 x 128 tiles) fills whole rounds of the device's CUs best -- at S-LIBRI
// 28 steps = 252 tiles on 256 CUs (15.7 ms / step) against 32 steps = 288 tiles (15.8 ms).  Ties go to the longer chunk.
static int persistent_chunk(int L, int B, int T, int H) {
    if (L < 2) return T < 32 ? T : 32;
    const int cus = device_cus();
    int best = 32;
    double best_eff = -1.0;
    for (int c = 24; c <= 40; ++c) {
        const long tiles = (long)((c * B + 127) / 128) * ((3 * H + 127) / 128) * (L - 1);
        const long rounds = (tiles + cus - 1) / cus;
        const double eff = (double)tiles / (double)(rounds * cus);
        if (eff >= best_eff) { best_eff = eff; best = c; }
    }
    return best > T ? T : best;
}
static size_t stack_gemm_ws(int L, int B, int T, int H, int chunk, bool fwd) {
    // workspace of the grouped per-chunk projections.  A launch carries 1 .. L-1 problems (fill / drain of the layer
    // wavefront, a ragged last chunk on its own) and the split-K factor -- hence the bytes -- is NOT monotone in the
    // problem count, so every count is priced (a split-bf16 product without room for its operands is an error
    // now, not a silent fall-back to the exact kernel).
    const int c = clamp_chunk(chunk, T);
    size_t w = 0;
    for (int np = 1; np <= (L > 1 ? L - 1 : 1); ++np) {
        const size_t v = fwd ? sa_gemm_group_workspace_bytes(np, c * B, 3 * H, H)
                             : sa_gemm_group_workspace_bytes(np, c * B, H, 3 * H);
        if (v > w) w = v;
    }
    return sa_align_up(w, 256);
}
// ... over every chunk length a call can use (the default is 24 .. 40 steps, callers may pass 1 .. 64, the last chunk of
// a sequence is whatever is left)
static size_t stack_gemm_ws_max(int L, int B, int T, int H, bool fwd) {
    size_t gw = 0;
    for (int c = 1; c <= 64; ++c) { const size_t w = stack_gemm_ws(L, B, T, H, c, fwd); if (w > gw) gw = w; }
    return gw;
}

extern "C" size_t sa_gru_stack_fwd_workspace_bytes(int L, int D, int B, int T, int H, int I0) {
    if (L <= 0 || D <= 0 || B <= 0 || T <= 0 || H <= 0 || I0 <= 0) return 0;
    size_t gw = stack_gemm_ws_max(L, B, T, H, true);
    // the whole-layer input projections (layer 0 of a unidirectional stack; every layer of a bidirectional one, both
    // directions grouped): the packed split-bf16 copies of their operands live here (gemm_f32.hip)
    const int Imax = I0 > D * H ? I0 : D * H;
    const size_t pw = sa_align_up(sa_gemm_group_workspace_bytes(D, T * B, 3 * H, D == 1 ? I0 : Imax), 256);
    if (pw > gw) gw = pw;
    // the planes of every layer's output and of the L - 1 dropped outputs (gru_fwd_planes_kernel)
    const size_t hxw = (size_t)planes_fwd_buffers(L, D) * planes_fwd_bytes(L, D, B, T, H);
    return (size_t)L * D * stack_ai_bytes(B, T, H) + gw + hxw + kFwdDumpBytes + kSyncBytes;
}

namespace {
// Inter-layer dropout of the stack (nn.GRU(dropout=p), /root/reference/speech/models/model.py:38): layer l < L-1 hands
// h_out[l] * mask to layer l+1; mask stream stream0 + l, index = the element's offset in the (T, B, D*H) array.
// h_drop[l] (l < L-1) are caller-owned (T, B, D*H) buffers for the dropped copies: the next layer's input projection,
// and its dW_ih product in the backward pass, read them.  The one-launch fused kernels apply the mask themselves;
// every other path applies it with an element-wise launch per layer (bidirectional) or per wavefront chunk.
struct DropCtx {
    SaDrop drop;
    unsigned stream0;
    float* const* h_drop;
    bool on() const { return drop.on(); }
};

ctcStatus_t stack_fwd_impl(const float* x, int I0, const float* const* w_ih, const float* const* b_ih,
                           const float* const* w_hh, const float* const* b_hh, float* const* h_out,
                           float* const* stash, int L, int D, int B, int T, int H, int chunk,
                           void* workspace, size_t workspace_bytes, void* stream_,
                           void* const* aux_streams, int n_aux, const DropCtx& dc) {
    SA_CLEAR_ERR();
    std::lock_guard<std::mutex> device_lock(g_stack_mutex[current_device()]);
    if (!x || !w_ih || !b_ih || !w_hh || !b_hh || !h_out || !workspace) return CTC_STATUS_INVALID_VALUE;
    const bool drop_on = dc.on() && L > 1;
    if (drop_on) {
        if (!dc.h_drop) return CTC_STATUS_INVALID_VALUE;
        for (int l = 0; l + 1 < L; ++l) if (!dc.h_drop[l]) return CTC_STATUS_INVALID_VALUE;
    }
    // layer l's input as layer l sees it (l >= 1)
    auto lower_of = [&](int l) -> const float* { return drop_on ? dc.h_drop[l - 1] : h_out[l - 1]; };
    if (L <= 0 || L > kMaxJobs || (D != 1 && D != 2) || B <= 0 || T <= 0 || H <= 0 || (H & 3) || I0 <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if (workspace_bytes < sa_gru_stack_fwd_workspace_bytes(L, D, B, T, H, I0)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    (void)g_health.poll(false);  // a failure seen by now switches the persistent path off (persist_mode() == 0)
    if (chunk <= 0 && n_aux <= 0 && D == 1 && xcd_shape_ok(L, B, H)) chunk = persistent_chunk(L, B, T, H);
    chunk = clamp_chunk(chunk > 64 ? 64 : chunk, T);
    auto ai_of = [&](int l, int d) { return (float*)((char*)workspace + (size_t)(l * D + d) * stack_ai_bytes(B, T, H)); };
    char* gws = (char*)workspace + (size_t)L * D * stack_ai_bytes(B, T, H);
    const size_t hx_each = planes_fwd_bytes(L, D, B, T, H);
    const size_t gws_bytes = workspace_bytes - (size_t)L * D * stack_ai_bytes(B, T, H) - (size_t)planes_fwd_buffers(L, D) * hx_each -
                             kFwdDumpBytes - kSyncBytes;
    char* hx_base = (char*)workspace + workspace_bytes - kSyncBytes - kFwdDumpBytes - (size_t)planes_fwd_buffers(L, D) * hx_each;
    unsigned* sync = (unsigned*)((char*)workspace + workspace_bytes - kSyncBytes);
    const long DH = (long)D * H;
    dim3 grid((H + 15) / 16, (B + 15) / 16, 1);
    ctcStatus_t st;

    auto make_job = [&](int l, int d, int t, int t_prev) {
        FwdJob J;
        J.ai = ai_of(l, d); J.w_hh = w_hh[l * D + d]; J.b_hh = b_hh[l * D + d];
        J.h_out = h_out[l] + (long)d * H; J.stash = stash ? stash[l * D + d] : nullptr;
        J.hs_b = DH; J.hs_t = (long)B * DH; J.t = t; J.t_prev = t_prev;
        return J;
    };
    FwdJobs P;
    P.B = B; P.H = H; P.rb = 1; P.rt = B; P.bt0 = 0; P.tile_rows = 16; P.stamp = nullptr;

    if (D == 2) {  // bidirectional: a layer needs both directions of the layer below -> layers in sequence,
                   // the two directions of a layer share every launch
        const int bi_nbt = (B + 15) / 16, bi_tpp = tiles_per_pass(2, H);
        unsigned bi_launches = 0;
        const size_t bi_lds = xcd_lds((size_t)2 * 4 * 4 * 256 * sizeof(float));  // the reduction scratch only (16-byte vectors in the planes kernel)
        const bool bi_xcd = n_aux <= 0 && xcd_shape_ok(2, B, H) && bi_lds <= 160 * 1024 && L * 2 * bi_nbt <= kSyncErr &&
                            (long)T * B * DH * 4 < 0x7fffffffL;
        if (bi_xcd) {
            if (hipMemsetAsync(sync, 0, 1024, stream) != hipSuccess) return CTC_STATUS_MEMOPS_FAILED;
            if (hipFuncSetAttribute((const void*)gru_fwd_persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bi_lds) != hipSuccess)
                return CTC_STATUS_EXECUTION_FAILED;
        }
        // Overlap (default on, SA_GRU_OVERLAP=0 off): a layer's two directions x batch tiles hold bi_used of the 8 XCDs.  The
        // layer's input projections are cut into kFwdChunks time chunks per direction in the order the recurrences consume
        // them (first steps of the forward direction, last steps of the reverse one); chunk 0 runs on the whole chip, the
        // others on the side stream, XCD-filtered to the idle XCDs, BESIDE the recurrence launches of the earlier chunks
        // (the same mechanism as the backward pass's weight gradients: one tile per block, the <= 170-register kernel, a
        // delay kernel so that the recurrence launch is dispatched first).
        const int per_xcd = groups_per_xcd(H);
        const int bi_used = (2 * min(bi_tpp, bi_nbt) + per_xcd - 1) / per_xcd;
        const unsigned bi_mask = bi_used >= 8 ? 0u : (0xffu & ~((1u << bi_used) - 1u));
        const int nck = fwd_chunks(T);
        const bool fside = bi_xcd && bi_mask && nck > 1 && T >= 16 * nck && bi_nbt <= bi_tpp && overlap_enabled() &&
                           L * (nck - 1) <= kSyncTileWords && g_side.init();
        const int S = (T + nck - 1) / nck;
        int side_launch = 0;
        for (int l = 0; l < L; ++l) {
            const float* in = l == 0 ? x : lower_of(l);
            const int I = l == 0 ? I0 : 2 * H;
            auto finish_layer = [&]() -> ctcStatus_t {  // the layer is complete on `stream`: its dropped copy for layer l+1
                if (!drop_on || l + 1 >= L) return CTC_STATUS_SUCCESS;
                return sa_dropout_apply_impl(h_out[l], dc.h_drop[l], (size_t)T * B * DH, 0, dc.drop, dc.stream0 + l, stream);
            };
            auto project = [&](int c, hipStream_t on, unsigned mask) {  // both directions' rows of time chunk c
                const int n = min(S, T - c * S);
                const float* gA[2]; const float* gB[2]; float* gC[2]; const float* gbias[2];
                for (int d = 0; d < 2; ++d) {
                    const long r0 = (long)(d ? T - c * S - n : c * S) * B;
                    gA[d] = in + r0 * I; gB[d] = w_ih[l * 2 + d]; gC[d] = ai_of(l, d) + r0 * 3 * H; gbias[d] = b_ih[l * 2 + d];
                }
                SaGemmOpts o;
                o.no_split = mask ? 1 : whole_no_split(n * B, 3 * H, 2); o.colsum = nullptr; o.xcc_mask = mask;
                o.tile_counter = mask ? sync + kSyncTiles + side_launch++ : nullptr;
                o.err_word = g_health.dev;  // a filtered launch that did not cover its tiles stops the step's update
                return sa_gemm_f32_group_impl(2, 0, 1, n * B, 3 * H, I, 1.f, gA, I, gB, I, 0.f, gC, 3 * H, gbias, nullptr,
                                              gws, gws_bytes, on, &o);  // (side chunks run between the main stream's products)
            };
            hipEvent_t ready[16];
            if (fside) {
                st = project(0, stream, 0u);
                if (st != CTC_STATUS_SUCCESS) return st;
                if (!g_side.order(stream, g_side.s)) return CTC_STATUS_EXECUTION_FAILED;  // the layer's input is complete
                hipLaunchKernelGGL(side_wait_dispatched_kernel, dim3(1), dim3(1), 0, g_side.s, (const unsigned*)(sync + kSyncReg),
                                   256u * (bi_launches + 1u));  // the layer's first recurrence launch (enqueued below) is in
                for (int c = 1; c < nck && c * S < T; ++c) {
                    st = project(c, g_side.s, bi_mask);
                    if (st != CTC_STATUS_SUCCESS) return st;
                    ready[c] = g_side.ev[g_side.next];
                    g_side.next = (g_side.next + 1) % SideStream::kEvents;
                    if (hipEventRecord(ready[c], g_side.s) != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;
                }
            } else {
                for (int d = 0; d < 2; ++d) {
                    st = gemm_whole(0, 1, T * B, 3 * H, I, in, I, w_ih[l * 2 + d], I, 0.f, ai_of(l, d), 3 * H, b_ih[l * 2 + d],
                                    gws, gws_bytes, stream);
                    if (st != CTC_STATUS_SUCCESS) return st;
                }
            }
            if (bi_xcd) {  // persistent launches run both directions of the layer: all T steps, or chunk by chunk
                PFwdJobs Q;
                Q.B = B; Q.H = H; Q.rb = 1; Q.rt = B; Q.err = g_health.dev; Q.spin_limit = spin_limit(); Q.fault = fault_injection(); Q.prio = persist_prio(); Q.timing = nullptr;
                Q.xcd_mode = 1; Q.nbt_all = bi_nbt; Q.ntile_u = H / 16; Q.reg = sync + kSyncReg;
                Q.flagless = flagless_mode() ? 1 : 0;
                // the bf16-planes form of the chunk kernel (default where it exists): the exchange is one planes buffer per
                // direction, h_out needs no sentinel
                const bool bi_planes = Q.flagless && hx_each > 0 && sa_opt(SA_OPT_GRU_FWD_PLANES) != 0;
                if (bi_planes) {
                    if (hipMemsetD32Async((hipDeviceptr_t)hx_base, (int)kPlaneSentinel, 2 * hx_each / 4, stream) != hipSuccess)
                        return CTC_STATUS_MEMOPS_FAILED;
                } else if (Q.flagless && !sentinel_fill(h_out[l], (size_t)T * B * DH, stream)) return CTC_STATUS_MEMOPS_FAILED;
                Q.stamp = nullptr; Q.n = 2;
                const int nlaunch = fside ? (T + S - 1) / S : 1;
                for (int c = 0; c < nlaunch; ++c) {
                    const int s0 = fside ? c * S : 0, n = fside ? min(S, T - c * S) : T;
                    if (c > 0 && hipStreamWaitEvent(stream, ready[c], 0) != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;
                    for (int d = 0; d < 2; ++d) {
                        PFwdJob& J = Q.j[d];
                        J.ai = ai_of(l, d); J.w_hh = w_hh[l * 2 + d]; J.b_hh = b_hh[l * 2 + d];
                        J.h_out = h_out[l] + (long)d * H; J.stash = stash ? stash[l * 2 + d] : nullptr;
                        J.counters = sync + (l * 2 + d) * bi_nbt;
                        J.hs_b = DH; J.hs_t = (long)B * DH; J.nsteps = n; J.base = (unsigned)(H / 16) * (unsigned)s0;
                        J.dt = d ? -1 : 1; J.t_first = d ? T - 1 : 0; J.t0 = d ? T - 1 - s0 : s0;
                        J.hx = bi_planes ? hx_base + (size_t)d * hx_each : nullptr;
                    }
                    for (int bt0 = 0; bt0 < bi_nbt; bt0 += bi_tpp) {  // passes over the batch tiles
                        Q.bt0 = bt0; Q.nbt = min(bi_tpp, bi_nbt - bt0);
                        Q.reg_base = bi_launches++ * 32u;
                        Q.dump = (float*)((char*)workspace + workspace_bytes - kSyncBytes - kFwdDumpBytes);
                        FwdChunkFn cfn = bi_planes ? fwd_chunk_planes_fn(H, Q.j[0].stash != nullptr)
                                         : Q.flagless ? fwd_chunk_fn(H, Q.j[0].stash != nullptr) : nullptr;  // (arrival counters: option gru.persist = 2)
                        if (cfn) hipLaunchKernelGGL(cfn, dim3(256), dim3(256), bi_lds, stream, Q);
                        else hipLaunchKernelGGL(gru_fwd_persist_kernel<true>, dim3(256), dim3(256), bi_lds, stream, Q);
                    }
                }
                st = finish_layer();
                if (st != CTC_STATUS_SUCCESS) return st;
                continue;
            }
            P.n = 2; grid.z = 2;
            for (int s = 0; s < T; ++s) {
                P.j[0] = make_job(l, 0, s, s == 0 ? -1 : s - 1);
                P.j[1] = make_job(l, 1, T - 1 - s, s == 0 ? -1 : T - s);
                hipLaunchKernelGGL(gru_fwd_step_kernel, grid, dim3(256), 0, stream, P);
            }
            st = finish_layer();
            if (st != CTC_STATUS_SUCCESS) return st;
        }
        SA_CHECK_LAUNCH();
        if (bi_xcd) g_health.submit(stream);
        return CTC_STATUS_SUCCESS;
    }

    // unidirectional: chunked layer wavefront.  Layer 0's input projection for all t up front (enqueued below, once the
    // path is known: the planes pre-fill may go to the side stream in front of it).
    Chains ch(stream, aux_streams, n_aux, B);
    if (!ch.ok) return CTC_STATUS_EXECUTION_FAILED;
    P.tile_rows = ch.tile_rows;
    const int nch = (T + chunk - 1) / chunk;
    // persistent chunk kernel: needs every block co-resident (one per CU) and its W_hh slice in LDS
    const int nbt = (B + 15) / 16, ntile_u = H / 16;
    size_t plds = 0;
    const bool persist_common = persist_mode() != 0 && g_health.init() && ch.n == 1 && (H % 64) == 0 &&
                                L * nbt <= kSyncErr && (long)T * B * H * 4 < 0x7fffffffL;
    // XCD-local groups: 8 XCDs x 32 CUs, 32 / ntile_u (layer, batch tile) groups per XCD; larger batches run in passes over
    // their batch tiles (tiles_per_pass), so the chip-wide "every block co-resident" bound does not apply to them
    const bool xcd = persist_common && xcd_shape_ok(L, B, H);
    const bool persist = xcd;
    if (xcd) plds = xcd_lds((size_t)2 * 4 * 3 * 256 * sizeof(float));  // WREG: the reduction scratch only
    unsigned persist_launches = 0;
    const bool flagless = xcd && flagless_mode();
    const size_t flds = xcd_lds((size_t)2 * 4 * 4 * 256 * sizeof(float));
    const bool fused_fwd = sa_opt(SA_OPT_GRU_FUSED) != 0 && flagless && flds <= 160 * 1024 && L * nbt <= kSyncErr;
    // the bf16-planes form of the one-launch kernel (default where it exists): its exchange is the planes buffers, h_out
    // needs no sentinel
    const bool planes = fused_fwd && sa_opt(SA_OPT_GRU_FWD_PLANES) != 0 && hx_each > 0;
    // (r6) the planes pre-fill (196 MB of stores at S-LIBRI) goes to the library's side stream, BESIDE the operand packs and
    // the layer-0 projection, instead of between the projection and the recurrence: the recurrence launch then starts on an
    // L2 that is not full of dirty sentinel lines (its own time 1.53 -> 1.46 ms) at the price of slower packs beside the fill;
    // net -0.03 ms per step in three one-box pairs (profiles/r06_forward_planes_experiments.txt; gru.exp bit 10 = in line: A/B)
    const bool side_fill = planes && !(sa_opt(SA_OPT_GRU_EXP) & 1024) && g_side.init();
    if (side_fill) {
        if (!g_side.order(stream, g_side.s)) return CTC_STATUS_EXECUTION_FAILED;  // (the previous call's kernels read these buffers)
        FillBatch pf(g_side.s, true);
        pf.add(hx_base, (size_t)(drop_on ? 2 * L - 1 : L) * hx_each / 4, kPlaneSentinel);
        pf.flush();
        if (!pf.ok) return CTC_STATUS_MEMOPS_FAILED;
    }
    st = gemm_whole(0, 1, T * B, 3 * H, I0, x, I0, w_ih[0], I0, 0.f, ai_of(0, 0), 3 * H, b_ih[0], gws, gws_bytes, stream);
    if (st != CTC_STATUS_SUCCESS) return st;
    if (side_fill && !g_side.order(g_side.s, stream)) return CTC_STATUS_EXECUTION_FAILED;
    FillBatch fills(stream, fused_fwd && fill_batch_enabled());
    if (planes && !side_fill) {
        fills.add(hx_base, (size_t)(drop_on ? 2 * L - 1 : L) * hx_each / 4, kPlaneSentinel);  // (the buffers are contiguous)
    } else if (planes) {
    } else if (flagless) {
        for (int l = 0; l < L; ++l) fills.add(h_out[l], (size_t)T * B * H, kSentinel);
    }
    if (!fused_fwd) fills.flush();
    if (!fills.ok) return CTC_STATUS_MEMOPS_FAILED;
    {   // the whole stack as ONE launch with in-kernel input projections (gru_fwd_fused_kernel); SA_GRU_FUSED=0: off
        if (fused_fwd) {
            fills.add(sync, 1024 / 4, 0u);
            fills.flush();
            if (!fills.ok) return CTC_STATUS_MEMOPS_FAILED;
            const bool timing_on = sa_opt(SA_OPT_GRU_TIMING) != 0;
            FusedFwdFn fused_fn = planes ? planes_fwd_fn(H, stash != nullptr, drop_on, timing_on)
                                         : fused_fwd_fn(H, stash != nullptr, drop_on, timing_on);
            if (!fused_fn) return CTC_STATUS_INVALID_VALUE;  // (xcd_shape_ok admits H = 128 .. 512 in steps of 64 only)
            if (hipFuncSetAttribute((const void*)fused_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)flds) != hipSuccess)
                return CTC_STATUS_EXECUTION_FAILED;
            PFusedFwd Q;
            Q.L = L; Q.B = B; Q.H = H; Q.T = T; Q.nbt_all = nbt; Q.ntile_u = ntile_u; Q.rb = 1; Q.rt = B;
            Q.ai0 = ai_of(0, 0); Q.prog = sync; Q.reg = sync + kSyncReg; Q.err = g_health.dev; Q.spin_limit = spin_limit(); Q.fault = fault_injection(); Q.prio = persist_prio();
            Q.stamp = g_prof.slot(0, true, false, T);
            Q.timing = timing_on && !drop_on ? (unsigned long long*)(sync + 256) : nullptr;
            if (Q.timing && hipMemsetAsync(sync + 256, 0, 256 * 8 * 8, stream) != hipSuccess) return CTC_STATUS_MEMOPS_FAILED;  // (r4 kernel: 4 words per block)
            Q.dump = (float*)((char*)workspace + workspace_bytes - kSyncBytes - kFwdDumpBytes);
            { const long r = sa_opt(SA_OPT_GRU_FWD_REPORT); Q.report_every = (r >= 1 && r <= 64 && (r & (r - 1)) == 0) ? (int)r : 8; }
            Q.drop = dc.drop; Q.drop_stream0 = dc.stream0;
            for (int l = 0; l < kMaxJobs; ++l) { Q.h_drop[l] = nullptr; Q.hx[l] = nullptr; Q.hxd[l] = nullptr; }
            for (int l = 0; planes && l < L; ++l) {
                Q.hx[l] = hx_base + (size_t)l * hx_each;
                Q.hxd[l] = drop_on && l + 1 < L ? hx_base + (size_t)(L + l) * hx_each : nullptr;
            }
            for (int l = 0; l < L; ++l) {
                Q.w_ih[l] = w_ih[l]; Q.b_ih[l] = b_ih[l]; Q.w_hh[l] = w_hh[l]; Q.b_hh[l] = b_hh[l];
                Q.h_out[l] = h_out[l]; Q.stash[l] = stash ? stash[l] : nullptr;
                Q.h_drop[l] = drop_on && l + 1 < L ? dc.h_drop[l] : nullptr;
            }
            const int tpp = tiles_per_pass(L, H);
            unsigned launches = 0;
            for (int bt0 = 0; bt0 < nbt; bt0 += tpp) {  // passes over the batch tiles (one pass up to B = 32 at L = 4)
                Q.bt0 = bt0; Q.nbt = min(tpp, nbt - bt0);
                Q.reg_base = launches++ * 32u;
                if (bt0 > 0) Q.stamp = nullptr;
                hipLaunchKernelGGL(fused_fn, dim3(256), dim3(256), flds, stream, Q);
            }
            SA_CHECK_LAUNCH();
            g_health.submit(stream);
            return CTC_STATUS_SUCCESS;
        }
    }
    if (persist) {
        if (hipMemsetAsync(sync, 0, kSyncBytes, stream) != hipSuccess) return CTC_STATUS_MEMOPS_FAILED;
        if (hipFuncSetAttribute((const void*)gru_fwd_persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)plds) != hipSuccess)
            return CTC_STATUS_EXECUTION_FAILED;
    }
    for (int w = 0; w < nch + L - 1; ++w) {
        // input projections of the chunk each upper layer is about to process (its lower layer finished it last
        // wave): full chunks of all layers go out as ONE grouped GEMM launch, a ragged last chunk on its own
        {
            const float* gA[kMaxJobs]; const float* gB[kMaxJobs]; float* gC[kMaxJobs]; const float* gb[kMaxJobs];
            int ng = 0;
            for (int l = 1; l < L; ++l) {
                const int c = w - l;
                if (c < 0 || c >= nch) continue;
                const int t0 = c * chunk, t1 = min(T, t0 + chunk);
                if (drop_on) {  // the chunk's rows of the layer below, dropped (it finished them last wave)
                    st = sa_dropout_apply_impl(h_out[l - 1] + (long)t0 * B * H, dc.h_drop[l - 1] + (long)t0 * B * H,
                                               (size_t)(t1 - t0) * B * H, (size_t)t0 * B * H, dc.drop, dc.stream0 + l - 1, stream);
                    if (st != CTC_STATUS_SUCCESS) return st;
                }
                const float* Ap = lower_of(l) + (long)t0 * B * H;
                float* Cp = ai_of(l, 0) + (long)t0 * B * 3 * H;
                if (t1 - t0 == chunk) {
                    gA[ng] = Ap; gB[ng] = w_ih[l]; gC[ng] = Cp; gb[ng] = b_ih[l]; ++ng;
                } else {
                    st = sa_gemm_f32_impl(0, 1, (t1 - t0) * B, 3 * H, H, 1.f, Ap, H, w_ih[l], H, 0.f, Cp, 3 * H,
                                          b_ih[l], nullptr, gws, gws_bytes, stream);
                    if (st != CTC_STATUS_SUCCESS) return st;
                }
            }
            if (ng > 0) {
                st = sa_gemm_f32_group_impl(ng, 0, 1, chunk * B, 3 * H, H, 1.f, gA, H, gB, H, 0.f, gC, 3 * H, gb,
                                            nullptr, gws, gws_bytes, stream);
                if (st != CTC_STATUS_SUCCESS) return st;
            }
        }
        if (persist) {  // ONE launch runs the whole chunk of every active layer
            PFwdJobs Q;
            Q.B = B; Q.H = H; Q.rb = 1; Q.rt = B; Q.err = g_health.dev; Q.spin_limit = spin_limit(); Q.fault = fault_injection(); Q.prio = persist_prio();
            Q.xcd_mode = xcd ? 1 : 0; Q.nbt = nbt; Q.nbt_all = nbt; Q.bt0 = 0; Q.ntile_u = ntile_u; Q.reg = sync + kSyncReg;
            Q.flagless = flagless ? 1 : 0;
            Q.stamp = nullptr;
            Q.timing = sa_opt(SA_OPT_GRU_TIMING) ? (unsigned long long*)(sync + 256) : nullptr;  // 3 KB of the sync page
            int n = 0;
            for (int l = 0; l < L; ++l) {
                const int c = w - l;
                if (c < 0 || c >= nch) continue;
                PFwdJob& J = Q.j[n++];
                J.ai = ai_of(l, 0); J.w_hh = w_hh[l]; J.b_hh = b_hh[l]; J.h_out = h_out[l];
                J.stash = stash ? stash[l] : nullptr; J.counters = sync + l * nbt;
                J.hs_b = H; J.hs_t = (long)B * H; J.t0 = c * chunk; J.nsteps = min(chunk, T - J.t0);
                J.dt = 1; J.t_first = 0; J.hx = nullptr;
                J.base = (unsigned)ntile_u * (unsigned)J.t0;
            }
            Q.n = n;
            if (xcd) {
                Q.stamp = g_prof.slot(0, n == L, false, chunk);
                const int tpp = tiles_per_pass(L, H);
                for (int bt0 = 0; bt0 < nbt; bt0 += tpp) {  // passes over the batch tiles
                    Q.bt0 = bt0; Q.nbt = min(tpp, nbt - bt0);
                    Q.reg_base = persist_launches++ * 32u;
                    if (bt0 > 0) Q.stamp = nullptr;
                    Q.dump = (float*)((char*)workspace + workspace_bytes - kSyncBytes - kFwdDumpBytes);
                    FwdChunkFn cfn = (Q.flagless && !Q.timing) ? fwd_chunk_fn(H, stash != nullptr) : nullptr;
                    if (cfn) hipLaunchKernelGGL(cfn, dim3(256), dim3(256), plds, stream, Q);
                    else hipLaunchKernelGGL(gru_fwd_persist_kernel<true>, dim3(256), dim3(256), plds, stream, Q);
                }
            }
            continue;
        }
        ch.fork();
        for (int s = 0; s < chunk; ++s) {
            int n = 0;
            for (int l = 0; l < L; ++l) {
                const int c = w - l;
                if (c < 0 || c >= nch) continue;
                const int t = c * chunk + s;
                if (t >= T) continue;
                P.j[n++] = make_job(l, 0, t, t == 0 ? -1 : t - 1);
            }
            if (n == 0) continue;
            P.n = n; grid.z = n;
            for (int k = 0; k < ch.n; ++k) {
                P.bt0 = ch.bt0[k]; grid.y = ch.nbt[k];
                P.stamp = k == 0 ? g_prof.slot(0, n == L, s > 0) : nullptr;
                hipLaunchKernelGGL(gru_fwd_step_kernel, grid, dim3(256), 0, ch.s[k], P);
            }
        }
        ch.join();
    }
    SA_CHECK_LAUNCH();
    if (xcd) g_health.submit(stream);
    return CTC_STATUS_SUCCESS;
}
}  // namespace

extern "C" ctcStatus_t sa_gru_stack_fwd(const float* x, int I0, const float* const* w_ih, const float* const* b_ih,
                                        const float* const* w_hh, const float* const* b_hh, float* const* h_out,
                                        float* const* stash, int L, int D, int B, int T, int H, int chunk,
                                        void* workspace, size_t workspace_bytes, void* stream_,
                                        void* const* aux_streams, int n_aux) {
    const DropCtx dc{sa_drop_make(0.f, 0ull), 0u, nullptr};
    return stack_fwd_impl(x, I0, w_ih, b_ih, w_hh, b_hh, h_out, stash, L, D, B, T, H, chunk, workspace, workspace_bytes,
                          stream_, aux_streams, n_aux, dc);
}

extern "C" ctcStatus_t sa_gru_stack_fwd_dropout(const float* x, int I0, const float* const* w_ih,
                                                const float* const* b_ih, const float* const* w_hh,
                                                const float* const* b_hh, float* const* h_out, float* const* h_drop,
                                                float* const* stash, int L, int D, int B, int T, int H, int chunk,
                                                void* workspace, size_t workspace_bytes, float p,
                                                unsigned long long seed, unsigned int mask_stream0, void* stream_) {
    if (!sa_drop_valid(p)) return CTC_STATUS_INVALID_VALUE;
    const DropCtx dc{sa_drop_make(p, seed), mask_stream0, h_drop};
    return stack_fwd_impl(x, I0, w_ih, b_ih, w_hh, b_hh, h_out, stash, L, D, B, T, H, chunk, workspace, workspace_bytes,
                          stream_, nullptr, 0, dc);
}

// The weight gradients of a unidirectional stack on SHARED packed operands (WGradIssuer::issue_shared): every matrix is
// split into its bf16 planes once -- the gate gradients of a layer as ONE 4H-row operand [dpr, dpz, dpn | dqn] (dai's
// three gates, then dah's third: dah's first two ARE dai's), which dW_ih reads as rows [0, 3H) and dW_hh as rows [0, 2H) +
// [3H, 4H).  0 = the shape does not take this path.
static bool shared_pack_enabled() {
    return sa_opt(SA_OPT_GRU_SHARED_PACK) != 0;
}
// (round 5) h_prev of a layer IS its output one time step earlier (zeros at t = 0), and without dropout the layer above
// reads that same output: with B a multiple of 16 a time step is B / 16 whole k-tiles, so ONE packed copy of h_out[l] --
// `lead` = B / 16 zero tiles in front of it in every row block -- serves dW_hh[l] (read at its base: the output shifted by one
// step behind zeros) and dW_ih[l+1] (read `lead` tiles in): L matrices to pack instead of 2 L - 1 (0.05 ms of pack launches
// per S-LIBRI step), the same values in the same k order, so the products are bit for bit what separate copies gave.  With
// dropout the layer above read the DROPPED output: those L - 1 copies are still packed (same tile stride, one launch).
struct SharedPackLayout { size_t g_each, h_each, x_bytes, cs_bytes, g_off, hp_off, lo_off, x_off, cs_off, sk_off, total; int parts;
                          int lead; };  // lead > 0: h_each counts ceil(K / 16) + lead tiles per row block
static bool shared_pack_layout(int L, int D, int B, int T, int H, int I0, SharedPackLayout& y) {
    const long K = (long)T * B;
    if (D != 1 || (H % 128) || K > 0x7fffffffL || 2 * L - 1 > 8 || !shared_pack_enabled()) return false;
    if (!sa_pk_enabled(3 * H, H, (int)K, L) || !sa_pk_enabled(3 * H, I0, (int)K, 1)) return false;
    y.lead = (B % 16) == 0 && K + B <= 0x7fffffffL ? B / 16 : 0;
    y.g_each = sa_pk_operand_bytes(4 * H, (int)K);
    y.h_each = sa_pk_operand_bytes(H, (int)(K + (y.lead ? B : 0)));
    y.x_bytes = sa_pk_operand_bytes(I0, (int)K);
    y.parts = sa_pk_rowsum_parts((int)K);
    y.cs_bytes = sa_align_up((size_t)L * y.parts * 4 * H * sizeof(float), 256);
    y.g_off = 0;
    y.hp_off = y.g_off + (size_t)L * y.g_each;
    y.lo_off = y.hp_off + (size_t)L * y.h_each;
    y.x_off = y.lo_off + (size_t)(L - 1) * y.h_each;
    y.cs_off = y.x_off + y.x_bytes;
    y.sk_off = y.cs_off + y.cs_bytes;
    const size_t a = sa_gemm_pk_group_workspace_bytes(2 * L - 1, 3 * H, H, (int)K);
    const size_t b = sa_gemm_pk_group_workspace_bytes(1, 3 * H, I0, (int)K);
    y.total = y.sk_off + sa_align_up(a > b ? a : b, 256);
    return true;
}
// Bidirectional stacks, layer by layer (WGradIssuer::issue_shared_bi): the backward recurrence kernel of a layer writes the
// two directions' 4H-row gate operands and their row sums (gru_bwd_fused_kernel<.., PACKG>) into the slot of the layer's
// PARITY -- the layer above's slot is still being read by its weight-gradient products on the side stream.  h_prev of the
// two directions and the layer's input (2H or I0 rows, shared by both directions' dW_ih) are packed by launches.
struct SharedPackLayoutBi { size_t g_each, h_each, lo_bytes, cs_each, slot, hp_off, lo_off, sk_off, total;
                            size_t k_each, w_each, kslot_off, w_off, skx_off, skx_bytes; };
static bool shared_pack_layout_bi(int L, int D, int B, int T, int H, int I0, SharedPackLayoutBi& y) {
    const long K = (long)T * B;
    const int Imax = I0 > 2 * H ? I0 : 2 * H;
    if (D != 2 || (H % 128) || (B % 16) || K > 0x7fffffffL || !shared_pack_enabled()) return false;
    if (!sa_pk_enabled(3 * H, H, (int)K, 2) || !sa_pk_enabled(3 * H, I0, (int)K, 2)) return false;
    y.g_each = sa_pk_operand_bytes(4 * H, (int)K);
    y.h_each = sa_pk_operand_bytes(H, (int)K);
    y.lo_bytes = sa_pk_operand_bytes(Imax, (int)K);
    y.cs_each = sa_align_up((size_t)((B + 15) / 16) * 4 * H * sizeof(float), 256);
    y.slot = 2 * (y.g_each + y.cs_each);   // per layer parity: G[d], then the row sums [d]
    y.hp_off = 2 * y.slot;
    y.lo_off = y.hp_off + 2 * y.h_each;
    y.sk_off = y.lo_off + y.lo_bytes;
    const size_t a = sa_gemm_pk_group_workspace_bytes(2, 3 * H, H, (int)K);
    const size_t b = sa_gemm_pk_group_workspace_bytes(2, 3 * H, Imax, (int)K);
    // the input-gradient products (the caller's stream): dai packed by the kernel (per layer parity and direction), W_ih
    // packed per layer, split-K scratch of their own
    if (!sa_pk_enabled((int)K, I0 < 2 * H ? I0 : 2 * H, 6 * H, 1)) return false;
    // (the two directions side by side along k: ONE operand of reduction length 6H, so an element of d in is written once)
    y.k_each = sa_pk_operand_bytes((int)K, 6 * H);
    y.w_each = sa_pk_operand_bytes(Imax, 6 * H);
    y.kslot_off = y.sk_off + sa_align_up(a > b ? a : b, 256);
    y.w_off = y.kslot_off + 2 * y.k_each;
    y.skx_off = y.w_off + y.w_each;
    y.skx_bytes = sa_align_up(sa_gemm_pk_group_workspace_bytes(2, (int)K, Imax, 6 * H), 256);
    y.total = y.skx_off + y.skx_bytes;
    (void)L;
    return true;
}
static size_t shared_pack_ws_bytes(int L, int D, int B, int T, int H, int I0) {
    SharedPackLayout y;
    SharedPackLayoutBi z;
    if (D == 2) return shared_pack_layout_bi(L, D, B, T, H, I0, z) ? z.total : 0;
    return shared_pack_layout(L, D, B, T, H, I0, y) ? y.total : 0;
}

// split-K workspace of the weight-gradient products (one region: the products of a call run on ONE stream, in order)
static size_t wgrad_ws_bytes(int L, int D, int B, int T, int H, int I0) {
    const int nmax = I0 > D * H ? I0 : D * H;
    size_t w = 0;
    for (int np = 1; np <= L * D && np <= kMaxJobs; ++np) {  // same-shaped products of all layers share a launch
        const size_t a = sa_gemm_group_workspace_bytes(np, 3 * H, nmax, T * B);
        const size_t b = sa_gemm_group_workspace_bytes(np, 3 * H, H, T * B);
        if (a > w) w = a;
        if (b > w) w = b;
    }
    const size_t sp = shared_pack_ws_bytes(L, D, B, T, H, I0);
    return sa_align_up(w > sp ? w : sp, 256);
}

extern "C" size_t sa_gru_stack_bwd_workspace_bytes(int L, int D, int B, int T, int H, int I0) {
    if (L <= 0 || D <= 0 || B <= 0 || T <= 0 || H <= 0 || I0 <= 0) return 0;
    const size_t per_dir = sa_align_up((size_t)2 * B * H * sizeof(float), 256) +      // dh ping-pong
                           sa_align_up((size_t)3 * H * H * sizeof(float), 256);       // W_hh^T
    const size_t mid = sa_align_up(((size_t)T * B + 128) * D * H * sizeof(float), 256);  // d h_out of a lower layer (+ one
                                                                                          // row block: chunked products end on whole blocks)
    size_t gw = stack_gemm_ws_max(L, B, T, H, false);
    {   // the whole-sequence input-gradient products (d x of layer 0; every layer of a bidirectional stack): packed copies
        const size_t a = sa_align_up(sa_gemm_group_workspace_bytes(1, T * B, I0, 3 * H), 256);
        const size_t b = D == 2 ? sa_align_up(sa_gemm_group_workspace_bytes(1, T * B, 2 * H, 3 * H), 256) : 0;
        if (a > gw) gw = a;
        if (b > gw) gw = b;
    }
    // the fused backward kernel: W_ih^T of the upper layers, and a tiled exchange copy of dai per layer
    const size_t wih_t = (D == 1 && L > 1 ? (size_t)(L - 1) * sa_align_up((size_t)3 * H * H * sizeof(float), 256) : 0) +
                         (size_t)L * D * sa_align_up((size_t)T * ((B + 15) / 16) * 16 * 3 * H * sizeof(float), 256) +
                         (size_t)256 * 256 * sizeof(float);
    return (size_t)L * D * per_dir + (size_t)(L > 1 ? L - 1 : 0) * mid + wih_t + gw +
           wgrad_ws_bytes(L, D, B, T, H, I0) + kSyncBytes;
}

// dh_top (T, B, D*H): gradient wrt the top layer's output.  Fills dai / dah [l*D+d] (T, B, 3H) for every layer and
// direction, and dx (T, B, I0) = gradient wrt the stack input (may be NULL).
namespace {
// Parameter gradients of the stack, produced INSIDE the backward call so that the library can schedule them:
//   dW_ih[l,d] = dai[l,d]^T in_l      (in_0 = x, in_l = h_out[l-1])        db_ih[l,d] = column sums of dai[l,d]
//   dW_hh[l,d] = dah[l,d]^T h_prev    (h_prev = stash[l,d][:, 4H:5H])      db_hh[l,d] = column sums of dah[l,d]
// Same-shaped products of all layers share grouped launches with the bias gradients fused in (row sums taken by the
// pack kernel of the split-bf16 path, gemm_f32.hip).  Bidirectional stacks: a finished layer's products go out on a SIDE
// stream as XCD-filtered launches that run on the XCDs the next layer's recurrence leaves idle (SA_GRU_OVERLAP=0: on the
// caller's stream after the recurrence, as unidirectional stacks always do -- their recurrence holds all 8 XCDs).
struct WGrad {
    const float* x;               // (T, B, I0)
    const float* const* h_out;    // [L] (T, B, D*H)
    float* const* dw_ih;          // [L*D] (3H, I_l)
    float* const* dw_hh;          // [L*D] (3H, H)
    float* const* db_ih;          // [L*D] (3H)
    float* const* db_hh;          // [L*D] (3H)
};

// Issues the weight-gradient products of layer-direction k = l*D+d over the time steps [t0, t1) on `stream`.
// `first[k]` tracks whether the slot has been written yet (beta = 0 the first time, 1 afterwards).
struct WGradIssuer {
    WGrad wg;
    const float* const* stash;
    float* const* dai;
    float* const* dah;
    int L, D, B, T, H, I0;
    const float* lower[kMaxJobs]; // layer l's input as the forward pass fed it (l >= 1): h_out[l-1], or its dropped copy
    unsigned xcc_mask = 0;        // != 0: the launches keep to these XCDs (the ones the recurrence leaves idle)
    unsigned* counters = nullptr; // zeroed device words, one per filtered launch
    unsigned* err_word = nullptr; // the sticky health word: a filtered launch that missed tiles reports there
    int next_counter = 0, max_counters = 0;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    bool first_ih[2 * kMaxJobs], first_hh[2 * kMaxJobs];
    bool gates_prepacked = false;  // gru_bwd_fused_kernel<PACKG> wrote the gate operands and their row sums (gsum_parts
    int gsum_parts = 0;            // partial sums per row) into the shared-pack workspace
    WGradIssuer(const WGrad& w, const float* const* st, float* const* da, float* const* dh, int L_, int D_, int B_,
                int T_, int H_, int I0_)
        : wg(w), stash(st), dai(da), dah(dh), L(L_), D(D_), B(B_), T(T_), H(H_), I0(I0_) {
        for (int i = 0; i < 2 * kMaxJobs; ++i) first_ih[i] = first_hh[i] = true;
        for (int l = 0; l < kMaxJobs; ++l) lower[l] = l >= 1 && l < L && wg.h_out ? wg.h_out[l - 1] : nullptr;
    }
    // Every product of a unidirectional stack over the whole sequence, on operands packed ONCE (shared_pack_layout):
    // 4 pack launches (gate gradients of all layers with their row sums; h_prev; the layer inputs; x), one fold launch per
    // bias family, one grouped product for dW_hh of all layers + dW_ih of the upper ones, one for dW_ih of layer 0.
    ctcStatus_t issue_shared(const SharedPackLayout& y, hipStream_t stream) {
        const int K = T * B;
        char* base = (char*)ws;
        const float* src[kMaxJobs]; const float* hi[kMaxJobs];
        for (int l = 0; l < L; ++l) { src[l] = dai[l]; hi[l] = dah[l] + 2 * H; }
        float* cs = (float*)(base + y.cs_off);
        ctcStatus_t st = CTC_STATUS_SUCCESS;
        if (!gates_prepacked) st = sa_pk_pack(L, src, hi, 3 * H, 3 * H, 4 * H, K, 0, base + y.g_off, y.g_each, cs, stream);
        if (st != CTC_STATUS_SUCCESS) return st;
        const int parts = gates_prepacked ? gsum_parts : y.parts;
        st = sa_pk_rowsum_fold(L, cs, parts, 4 * H, 3 * H, 3 * H, 0, wg.db_ih, 0.f, stream, 2 * H, H, wg.db_hh);  // both bias families
        if (st != CTC_STATUS_SUCCESS) return st;
        const bool lead = y.lead > 0 && wg.h_out != nullptr;
        const int kbs = lead ? (K + 15) / 16 + y.lead : 0;  // tiles per row block of the h operands
        bool dropped = false;  // the upper layers read something else than the layer below's output (its dropped copy)
        for (int l = 1; l < L; ++l) dropped = dropped || lower[l] != wg.h_out[l - 1];
        if (lead) {  // one copy of every layer's output, y.lead zero tiles in front (see SharedPackLayout)
            for (int l = 0; l < L; ++l) src[l] = wg.h_out[l];
            st = sa_pk_pack(L, src, nullptr, 0, H, H, K, 0, base + y.hp_off, y.h_each, nullptr, stream, kbs, y.lead, 1);
            if (st != CTC_STATUS_SUCCESS) return st;
            if (dropped) {
                for (int l = 1; l < L; ++l) src[l - 1] = lower[l];
                st = sa_pk_pack(L - 1, src, nullptr, 0, H, H, K, 0, base + y.lo_off, y.h_each, nullptr, stream, kbs, y.lead, 0);
                if (st != CTC_STATUS_SUCCESS) return st;
            }
        } else {
            for (int l = 0; l < L; ++l) src[l] = stash[l] + 4 * H;
            st = sa_pk_pack(L, src, nullptr, 0, 5 * H, H, K, 0, base + y.hp_off, y.h_each, nullptr, stream);
            if (st != CTC_STATUS_SUCCESS) return st;
            if (L > 1) {
                for (int l = 1; l < L; ++l) src[l - 1] = lower[l];
                st = sa_pk_pack(L - 1, src, nullptr, 0, H, H, K, 0, base + y.lo_off, y.h_each, nullptr, stream);
                if (st != CTC_STATUS_SUCCESS) return st;
            }
        }
        src[0] = wg.x;
        st = sa_pk_pack(1, src, nullptr, 0, I0, I0, K, 0, base + y.x_off, y.x_bytes, nullptr, stream);
        if (st != CTC_STATUS_SUCCESS) return st;
        const char* pa[kMaxJobs]; const char* pb[kMaxJobs]; float* pc[kMaxJobs];
        int np = 0;
        for (int l = 0; l < L; ++l, ++np) { pa[np] = base + y.g_off + l * y.g_each; pb[np] = base + y.hp_off + l * y.h_each; pc[np] = wg.dw_hh[l]; }
        const unsigned jump_probs = (1u << L) - 1u;
        const size_t in_tiles = lead ? (size_t)y.lead * 12288 : 0;  // the matrix itself starts y.lead tiles into a row block
        for (int l = 1; l < L; ++l, ++np) {
            pa[np] = base + y.g_off + l * y.g_each;
            pb[np] = (lead && !dropped ? base + y.hp_off : base + y.lo_off) + (l - 1) * y.h_each + in_tiles;
            pc[np] = wg.dw_ih[l];
        }
        SaGemmOpts go;
        go.no_split = 0; go.colsum = nullptr; go.xcc_mask = 0; go.tile_counter = nullptr; go.b_kb_stride = kbs;
        st = sa_gemm_pk_group(np, 3 * H, H, K, pa, 2 * H, H, jump_probs, pb, 0.f, pc, H, base + y.sk_off, ws_bytes - y.sk_off, stream,
                              &go);
        if (st != CTC_STATUS_SUCCESS) return st;
        pa[0] = base + y.g_off; pb[0] = base + y.x_off; pc[0] = wg.dw_ih[0];
        st = sa_gemm_pk_group(1, 3 * H, I0, K, pa, 0, 0, 0u, pb, 0.f, pc, I0, base + y.sk_off, ws_bytes - y.sk_off, stream);
        if (st != CTC_STATUS_SUCCESS) return st;
        for (int k = 0; k < L; ++k) first_ih[k] = first_hh[k] = false;
        return CTC_STATUS_SUCCESS;
    }
    // spans[k] = {t0, t1} (t1 <= t0: nothing).  Same-shaped problems share a grouped launch.
    // One layer of a bidirectional stack on shared packed operands (shared_pack_layout_bi); the gate operands and their
    // row sums are in the layer's parity slot already.  mask != 0: XCD-filtered launches (side stream).
    bool shared_ok_bi(SharedPackLayoutBi& y) const {
        if (D != 2 || !wg.x || !wg.h_out || !wg.dw_ih || !wg.dw_hh || !wg.db_ih || !wg.db_hh) return false;
        for (int k = 0; k < 2 * L; ++k)
            if (!wg.dw_ih[k] || !wg.dw_hh[k] || !wg.db_ih[k] || !wg.db_hh[k] || (k >= 2 && !lower[k / 2])) return false;
        return shared_pack_layout_bi(L, D, B, T, H, I0, y) && y.total <= ws_bytes;
    }
    char* bi_gpk(const SharedPackLayoutBi& y, int l, int d) const { return (char*)ws + (size_t)(l & 1) * y.slot + (size_t)d * y.g_each; }
    char* bi_kpk(const SharedPackLayoutBi& y, int l, int d) const {
        return (char*)ws + y.kslot_off + (size_t)(l & 1) * y.k_each + (size_t)d * (3 * H / 16) * 12288;
    }
    // d in (T B, I) = [dai[l, 0] | dai[l, 1]] [W_ih[l, 0]; W_ih[l, 1]] on the operand the recurrence kernel packed (both
    // directions side by side along k) -- first the weights, packed the same way ...
    ctcStatus_t input_grad_bi_weights(const SharedPackLayoutBi& y, int l, const float* const* w_ih, hipStream_t stream) {
        const int I = l == 0 ? I0 : 2 * H;
        const float* src[2] = {w_ih[l * 2], w_ih[l * 2 + 1]};
        return sa_pk_pack(2, src, nullptr, 0, I, I, 3 * H, 0, (char*)ws + y.w_off, (size_t)(3 * H / 16) * 12288, nullptr, stream,
                          6 * H / 16);
    }
    // ... then the product over nrb row blocks from rb0[p], p < np (np = 2: the two ends of a chunk share a launch; the
    // last block of the sequence may be partial -- its surplus rows land in the padding of the mid buffers); mask != 0:
    // XCD-filtered (side stream)
    ctcStatus_t input_grad_bi_rows(const SharedPackLayoutBi& y, int l, float* din, int np, const int* rb0, int nrb,
                                   hipStream_t stream, unsigned mask, const SaDrop* drop = nullptr, unsigned drop_stream = 0) {
        const int I = l == 0 ? I0 : 2 * H;
        char* base = (char*)ws;
        const char* pa[2]; const char* pb[2]; float* pc[2];
        for (int p = 0; p < np; ++p) {
            pa[p] = bi_kpk(y, l, 0) + (size_t)rb0[p] * (6 * H / 16) * 12288;
            pb[p] = base + y.w_off;
            pc[p] = din + (size_t)rb0[p] * 128 * I;
        }
        SaGemmOpts o;
        o.no_split = 0; o.colsum = nullptr; o.xcc_mask = 0; o.tile_counter = nullptr; o.err_word = err_word;
        if (mask && counters && next_counter < max_counters) { o.xcc_mask = mask; o.tile_counter = counters + next_counter++; }
        // the layer below fed this layer its DROPPED output (nn.GRU(dropout=p)): the same mask routes the gradient -- in the
        // product's epilogue (din is the masked tensor itself, so an element's mask index is its offset in din)
        o.drop = drop; o.drop_stream = drop_stream; o.drop_base = din;
        const int rows_total = T * B;
        // whole-sequence call: the exact row count (the caller's dx has no padding); chunks: whole blocks
        const int M = (np == 1 && rb0[0] == 0 && nrb * 128 >= rows_total) ? rows_total : nrb * 128;
        return sa_gemm_pk_group(np, M, I, 6 * H, pa, 0, 0, 0u, pb, 0.f, pc, I, base + y.skx_off, y.skx_bytes, stream, &o);
    }
    float* bi_gsum(const SharedPackLayoutBi& y, int l, int d) const {
        return (float*)((char*)ws + (size_t)(l & 1) * y.slot + 2 * y.g_each + (size_t)d * y.cs_each);
    }
    ctcStatus_t issue_shared_bi(const SharedPackLayoutBi& y, int l, hipStream_t stream, unsigned mask) {
        const int K = T * B, I = l == 0 ? I0 : 2 * H, nbt = (B + 15) / 16;
        char* base = (char*)ws;
        const float* src[2];
        for (int d = 0; d < 2; ++d) src[d] = stash[l * 2 + d] + 4 * H;
        ctcStatus_t st = sa_pk_pack(2, src, nullptr, 0, 5 * H, H, K, 0, base + y.hp_off, y.h_each, nullptr, stream);
        if (st != CTC_STATUS_SUCCESS) return st;
        src[0] = l == 0 ? wg.x : lower[l];
        st = sa_pk_pack(1, src, nullptr, 0, I, I, K, 0, base + y.lo_off, y.lo_bytes, nullptr, stream);
        if (st != CTC_STATUS_SUCCESS) return st;
        for (int d = 0; d < 2; ++d) {  // (the two directions' row sums are not one [prob][part][row] array: a fold each)
            float* o_ih[1] = {wg.db_ih[l * 2 + d]}; float* o_hh[1] = {wg.db_hh[l * 2 + d]};
            st = sa_pk_rowsum_fold(1, bi_gsum(y, l, d), nbt, 4 * H, 3 * H, 3 * H, 0, o_ih, 0.f, stream, 2 * H, H, o_hh);
            if (st != CTC_STATUS_SUCCESS) return st;
        }
        SaGemmOpts o;
        o.no_split = 0; o.colsum = nullptr; o.xcc_mask = 0; o.tile_counter = nullptr; o.err_word = err_word;
        const char* pa[2]; const char* pb[2]; float* pc[2];
        for (int pass = 0; pass < 2; ++pass) {  // 0: dW_hh (rows [0, 2H) + [3H, 4H) of G, times h_prev), 1: dW_ih
            for (int d = 0; d < 2; ++d) {
                pa[d] = bi_gpk(y, l, d);
                pb[d] = pass == 0 ? base + y.hp_off + (size_t)d * y.h_each : base + y.lo_off;
                pc[d] = pass == 0 ? wg.dw_hh[l * 2 + d] : wg.dw_ih[l * 2 + d];
            }
            if (mask && counters && next_counter < max_counters) { o.xcc_mask = mask; o.tile_counter = counters + next_counter++; }
            else { o.xcc_mask = 0; o.tile_counter = nullptr; }
            const int N = pass == 0 ? H : I;
            st = sa_gemm_pk_group(2, 3 * H, N, K, pa, 2 * H, H, pass == 0 ? 3u : 0u, pb, 0.f, pc, N, base + y.sk_off,
                                  ws_bytes - y.sk_off, stream, &o);
            if (st != CTC_STATUS_SUCCESS) return st;
        }
        first_ih[l * 2] = first_ih[l * 2 + 1] = first_hh[l * 2] = first_hh[l * 2 + 1] = false;
        return CTC_STATUS_SUCCESS;
    }
    // issue_shared applies to this stack when every product is still owed in full
    bool shared_ok(SharedPackLayout& y) const {
        if (D != 1 || xcc_mask || !wg.x || !wg.dw_ih || !wg.dw_hh || !wg.db_ih || !wg.db_hh) return false;
        for (int k = 0; k < L; ++k)
            if (!first_ih[k] || !first_hh[k] || !wg.dw_ih[k] || !wg.dw_hh[k] || !wg.db_ih[k] || !wg.db_hh[k] || (k > 0 && !lower[k]))
                return false;
        return shared_pack_layout(L, D, B, T, H, I0, y) && y.total <= ws_bytes;
    }
    ctcStatus_t issue(const int (*spans)[2], hipStream_t stream, bool allow_split) {
        const int n = L * D;
        if (D == 1 && allow_split) {
            bool whole = true;
            for (int k = 0; k < n; ++k) whole = whole && spans[k][0] == 0 && spans[k][1] == T;
            SharedPackLayout y;
            if (whole && shared_ok(y)) return issue_shared(y, stream);
        }
        if (gates_prepacked) return CTC_STATUS_EXECUTION_FAILED;  // the row-major gate gradients were not written
        bool done_ih[2 * kMaxJobs], done_hh[2 * kMaxJobs];
        for (int k = 0; k < n; ++k) done_ih[k] = done_hh[k] = spans[k][1] <= spans[k][0];
        SaGemmOpts o;
        o.no_split = allow_split ? 0 : 1;
        o.xcc_mask = 0; o.tile_counter = nullptr;
        for (int pass = 0; pass < 2; ++pass) {      // 0: dW_hh (N = H, B operand = stash h_prev), 1: dW_ih
            bool* done = pass == 0 ? done_hh : done_ih;
            bool* first = pass == 0 ? first_hh : first_ih;
            for (int k0 = 0; k0 < n; ++k0) {
                if (done[k0]) continue;
                const int rows = (spans[k0][1] - spans[k0][0]) * B;
                const int N0 = pass == 0 ? H : (k0 / D == 0 ? I0 : D * H);
                const float* gA[kMaxJobs]; const float* gB[kMaxJobs]; float* gC[kMaxJobs]; float* gS[kMaxJobs];
                int ng = 0;
                for (int k = k0; k < n && ng < kMaxJobs; ++k) {
                    if (done[k]) continue;
                    const int Nk = pass == 0 ? H : (k / D == 0 ? I0 : D * H);
                    if ((spans[k][1] - spans[k][0]) * B != rows || Nk != N0 || first[k] != first[k0]) continue;
                    const long r0 = (long)spans[k][0] * B;
                    const int l = k / D;
                    if (pass == 0) {
                        gA[ng] = dah[k] + r0 * 3 * H; gB[ng] = stash[k] + r0 * 5 * H + 4 * H;
                        gC[ng] = wg.dw_hh[k]; gS[ng] = wg.db_hh[k];
                    } else {
                        gA[ng] = dai[k] + r0 * 3 * H; gB[ng] = (l == 0 ? wg.x : lower[l]) + r0 * N0;
                        gC[ng] = wg.dw_ih[k]; gS[ng] = wg.db_ih[k];
                    }
                    done[k] = true;
                    ++ng;
                }
                o.colsum = gS;
                if (xcc_mask && counters && next_counter < max_counters) {
                    o.xcc_mask = xcc_mask; o.tile_counter = counters + next_counter++;
                    o.err_word = err_word;
                } else {
                    o.xcc_mask = 0; o.tile_counter = nullptr;
                }
                const float beta = first[k0] ? 0.f : 1.f;
                const ctcStatus_t st = sa_gemm_f32_group_impl(ng, 1, 0, 3 * H, N0, rows, 1.f, gA, 3 * H, gB,
                                                              pass == 0 ? 5 * H : N0, beta, gC, N0, nullptr, nullptr,
                                                              ws, ws_bytes, stream, &o);
                if (st != CTC_STATUS_SUCCESS) return st;
            }
            for (int k = 0; k < n; ++k)
                if (spans[k][1] > spans[k][0]) first[k] = false;
        }
        return CTC_STATUS_SUCCESS;
    }
};

ctcStatus_t stack_bwd_impl(const float* dh_top, const float* const* stash, const float* const* w_ih,
                           const float* const* w_hh, float* const* dai, float* const* dah, float* dx,
                           int I0, int L, int D, int B, int T, int H, int chunk, void* workspace,
                           size_t workspace_bytes, void* stream_, void* const* aux_streams, int n_aux,
                           const WGrad* wg, const DropCtx& dc) {
    SA_CLEAR_ERR();
    std::lock_guard<std::mutex> device_lock(g_stack_mutex[current_device()]);
    if (!dh_top || !stash || !w_ih || !w_hh || !dai || !dah || !workspace) return CTC_STATUS_INVALID_VALUE;
    const bool drop_on = dc.on() && L > 1;
    if (drop_on) {
        if (!dc.h_drop) return CTC_STATUS_INVALID_VALUE;
        for (int l = 0; l + 1 < L; ++l) if (!dc.h_drop[l]) return CTC_STATUS_INVALID_VALUE;
    }
    if (L <= 0 || L > kMaxJobs || (D != 1 && D != 2) || B <= 0 || T <= 0 || H <= 0 || (H & 3) || I0 <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if (workspace_bytes < sa_gru_stack_bwd_workspace_bytes(L, D, B, T, H, I0)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    (void)g_health.poll(false);  // a failure seen by now switches the persistent path off (persist_mode() == 0)
    if (chunk <= 0 && n_aux <= 0 && D == 1 && xcd_shape_ok(L, B, H)) chunk = persistent_chunk(L, B, T, H);
    chunk = clamp_chunk(chunk > 64 ? 64 : chunk, T);
    const size_t per_dir = sa_align_up((size_t)2 * B * H * sizeof(float), 256) +
                           sa_align_up((size_t)3 * H * H * sizeof(float), 256);
    const size_t mid_bytes = sa_align_up(((size_t)T * B + 128) * D * H * sizeof(float), 256);
    char* ws = (char*)workspace;
    const size_t wih_t_each = sa_align_up((size_t)3 * H * H * sizeof(float), 256);
    const size_t wih_t_off = (size_t)L * D * per_dir + (size_t)(L > 1 ? L - 1 : 0) * mid_bytes;
    const size_t xch_each = sa_align_up((size_t)T * ((B + 15) / 16) * 16 * 3 * H * sizeof(float), 256);
    const size_t xch_off = wih_t_off + (D == 1 && L > 1 ? (size_t)(L - 1) * wih_t_each : 0);
    const size_t dump_off = xch_off + (size_t)L * D * xch_each;
    const size_t fixed_bytes = dump_off + (size_t)256 * 256 * sizeof(float);
    char* gws = ws + fixed_bytes;
    const size_t wws_bytes = wgrad_ws_bytes(L, D, B, T, H, I0);
    const size_t gws_bytes = workspace_bytes - fixed_bytes - kSyncBytes - wws_bytes;
    char* wws = gws + gws_bytes;
    WGradIssuer issuer(wg ? *wg : WGrad{}, stash, dai, dah, L, D, B, T, H, I0);
    issuer.ws = wws; issuer.ws_bytes = wws_bytes;
    if (drop_on) for (int l = 1; l < L; ++l) issuer.lower[l] = dc.h_drop[l - 1];
    // everything that is still owed when the recurrence is done goes out on the caller's stream (the fallback path)
    int wg_hi[2 * kMaxJobs];
    for (int k = 0; k < L * D; ++k) wg_hi[k] = T;
    auto wgrad_rest = [&]() -> ctcStatus_t {
        if (!wg) return CTC_STATUS_SUCCESS;
        int spans[2 * kMaxJobs][2];
        for (int k = 0; k < L * D; ++k) { spans[k][0] = 0; spans[k][1] = wg_hi[k]; wg_hi[k] = 0; }
        return issuer.issue(spans, stream, true);
    };
    unsigned* sync = (unsigned*)(ws + workspace_bytes - kSyncBytes);
    auto dh_buf = [&](int l, int d, int which) {
        return (float*)(ws + (size_t)(l * D + d) * per_dir) + (size_t)which * B * H;
    };
    auto wt_of = [&](int l, int d) {
        return (float*)(ws + (size_t)(l * D + d) * per_dir + sa_align_up((size_t)2 * B * H * sizeof(float), 256));
    };
    auto mid_of = [&](int l) { return (float*)(ws + (size_t)L * D * per_dir + (size_t)l * mid_bytes); };  // l < L-1
    const long DH = (long)D * H;
    auto transpose_whh = [&]() {   // W_hh^T per (layer, direction): what every kernel but the one-launch fused one reads
        TransposeList tl;
        int n = 0;
        for (int l = 0; l < L; ++l)
            for (int d = 0; d < D; ++d) {
                tl.in[n] = w_hh[l * D + d]; tl.out[n] = wt_of(l, d);
                if (++n == 16 || (l == L - 1 && d == D - 1)) {
                    hipLaunchKernelGGL(transpose_many_kernel, dim3((H + 31) / 32, (3 * H + 31) / 32, n), dim3(256), 0, stream, tl, 3 * H, H);
                    n = 0;
                }
            }
    };
    dim3 grid((H + 15) / 16, (B + 15) / 16, 1);
    BwdJobs P;
    P.B = B; P.H = H; P.rb = 1; P.rt = B; P.bt0 = 0; P.tile_rows = 16; P.stamp = nullptr;
    ctcStatus_t st;
    // step counter per (layer, dir) selects the ping-pong buffer
    auto make_job = [&](int l, int d, int t, int t_next, int step) {
        BwdJob J;
        const float* dho = (l == L - 1) ? dh_top : mid_of(l);
        J.dh_out = dho + (long)d * H; J.ds_b = DH; J.ds_t = (long)B * DH;
        J.stash = stash[l * D + d]; J.w_hh_t = wt_of(l, d); J.dai = dai[l * D + d]; J.dah = dah[l * D + d];
        J.dh_ping = dh_buf(l, d, step & 1); J.dh_pong = dh_buf(l, d, (step & 1) ^ 1);
        J.t = t; J.t_next = t_next;
        return J;
    };

    if (D == 2) {
        transpose_whh();
        const int bi_nbt = (B + 15) / 16, bi_tpp = tiles_per_pass(2, H);
        unsigned bi_launches = 0;
        const size_t bi_lds = xcd_lds((size_t)2 * 4 * 256 * sizeof(float));
        const bool bi_xcd = n_aux <= 0 && xcd_shape_ok(2, B, H) && bi_lds <= 160 * 1024 && L * 2 * bi_nbt <= kSyncErr &&
                            (long)T * B * 3 * H * 4 < 0x7fffffffL;
        // a layer's groups (2 directions x the batch tiles of a pass) sit on XCDs 0 .. used-1: the weight gradients of
        // the layer above run on the OTHER XCDs meanwhile (XCD-filtered persistent-tile GEMM launches on the side stream)
        const int bi_used = (2 * min(bi_tpp, bi_nbt) + groups_per_xcd(H) - 1) / groups_per_xcd(H);
        const unsigned bi_mask = bi_used >= 8 ? 0u : (0xffu & ~((1u << bi_used) - 1u));
        const bool bi_side = wg && bi_xcd && bi_mask && overlap_enabled() && g_side.init();
        issuer.counters = sync + kSyncTiles; issuer.max_counters = kSyncTileWords; issuer.err_word = g_health.dev;
        // tiled exchange, operands a step ahead (gru_bwd_fused_kernel without the second product): H = 512 / 256
        const bool bi_tiled = bi_xcd && flagless_mode() && (long)T * bi_nbt * 16 * 3 * H * 4 < 0x7fffffffL;
        // ... which also writes the weight gradients' gate operands, packed (gru_bwd_fused_kernel<.., PACKG>); the products
        // of a layer then run on shared operands (issue_shared_bi), filtered beside the next layer or plain on `stream`
        SharedPackLayoutBi spb;
        const bool bi_packg = bi_tiled && wg && sa_opt(SA_OPT_GRU_PACK_IN_KERNEL) != 0 && bi_nbt <= bi_tpp && issuer.shared_ok_bi(spb) &&
                              bwd_fused_fn(H, false, false, true) != nullptr;
        const BwdPersistFn bi_tiled_fn = bi_tiled ? bwd_fused_fn(H, false, false, bi_packg) : nullptr;
        const size_t bi_lds_run = bi_packg ? xcd_lds((size_t)(4 * 4 * 256 + 2 * 4 * 16 * 20) * sizeof(float)) : bi_lds;
        if (bi_xcd) {
            if (hipMemsetAsync(sync, 0, 1024, stream) != hipSuccess) return CTC_STATUS_MEMOPS_FAILED;
            if (hipFuncSetAttribute(bi_tiled_fn ? (const void*)bi_tiled_fn : (const void*)bwd_persist_fn(),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bi_lds_run) != hipSuccess)
                return CTC_STATUS_EXECUTION_FAILED;
        }
        hipEvent_t slot_read[2] = {nullptr, nullptr};
        bool slot_busy[2] = {false, false};
        for (int l = L - 1; l >= 0; --l) {
            if (bi_packg && slot_busy[l & 1]) {
                if (hipStreamWaitEvent(stream, slot_read[l & 1], 0) != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;
                slot_busy[l & 1] = false;
            }
            if (bi_xcd) {  // ONE persistent launch unwinds both directions of the layer over all T steps
                PBwdJobs Q;
                Q.B = B; Q.H = H; Q.nbt_all = bi_nbt; Q.ntile_u = H / 16; Q.rb = 1; Q.rt = B;
                Q.flagless = flagless_mode() ? 1 : 0;
                if (Q.flagless) {  // both directions' exchange buffers in one launch
                    FillBatch fb(stream, fill_batch_enabled());
                    for (int d = 0; d < 2; ++d)
                        fb.add(bi_tiled_fn ? (float*)(ws + xch_off + (size_t)(l * 2 + d) * xch_each) : dah[l * 2 + d],
                               bi_tiled_fn ? (size_t)(xring_enabled() ? min(T, kXRing) : T) * bi_nbt * 16 * 3 * H
                                           : (size_t)T * B * 3 * H, kSentinel);
                    fb.flush();
                    if (!fb.ok) return CTC_STATUS_MEMOPS_FAILED;
                }
                Q.err = g_health.dev; Q.spin_limit = spin_limit(); Q.fault = fault_injection(); Q.prio = persist_prio(); Q.packed = (bi_tiled_fn && xring_enabled()) ? 2 : 1; Q.reg = sync + kSyncReg; Q.w_rowmajor = 0;
                Q.stamp = nullptr; Q.n = 2; Q.timing = nullptr; Q.drop = sa_drop_make(0.f, 0ull);
                Q.pk_kb = bi_packg ? (long)T * B / 16 : 0;
                Q.kpk_kb = 6 * H / 16;
                for (int d = 0; d < 2; ++d) {
                    PBwdJob& J = Q.j[d];
                    J.dx_drop_stream = 0u;
                    J.gpk = bi_packg ? issuer.bi_gpk(spb, l, d) : nullptr;
                    J.gsum = bi_packg ? issuer.bi_gsum(spb, l, d) : nullptr;
                    J.kpk = bi_packg ? issuer.bi_kpk(spb, l, d) : nullptr;
                    const float* dho = (l == L - 1) ? dh_top : mid_of(l);
                    J.dh_out = dho + (long)d * H; J.ds_b = DH; J.ds_t = (long)B * DH;
                    J.stash = stash[l * 2 + d]; J.w_hh_t = wt_of(l, d); J.dai = dai[l * 2 + d]; J.dah = dah[l * 2 + d];
                    J.dh_state = dh_buf(l, d, 0); J.counters = sync + (l * 2 + d) * bi_nbt;
                    J.nsteps = T; J.base = 0;
                    J.w_ih_t = nullptr; J.dx_out = nullptr; J.xs_b = J.xs_t = 0;
                    J.xch = bi_tiled_fn ? (float*)(ws + xch_off + (size_t)(l * 2 + d) * xch_each) : nullptr;
                    J.dump = (float*)(ws + dump_off);
                    J.dt = d ? 1 : -1; J.t0 = J.t_first = d ? 0 : T - 1;   // the reverse chain unwinds forward in time
                }
                for (int bt0 = 0; bt0 < bi_nbt; bt0 += bi_tpp) {  // passes over the batch tiles
                    Q.bt0 = bt0; Q.nbt = min(bi_tpp, bi_nbt - bt0);
                    Q.reg_base = bi_launches++ * 32u;
                    hipLaunchKernelGGL(bi_tiled_fn ? bi_tiled_fn : bwd_persist_fn(), dim3(256), dim3(256), bi_lds_run, stream, Q);
                }
            } else {
            P.n = 2; grid.z = 2;
            for (int s = 0; s < T; ++s) {
                P.j[0] = make_job(l, 0, T - 1 - s, s == 0 ? -1 : T - s, s);      // forward-in-time direction unwinds
                P.j[1] = make_job(l, 1, s, s == 0 ? -1 : s - 1, s);              // reverse direction unwinds forward
                hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(256), 0, stream, P);
            }
            }
            // gradient wrt this layer's input = sum over directions of dai W_ih
            float* din = l > 0 ? mid_of(l - 1) : dx;
            const int I = l > 0 ? 2 * H : I0;
            bool dx_streamed = false;
            if (din && bi_packg) {
                st = issuer.input_grad_bi_weights(spb, l, w_ih, stream);
                if (st != CTC_STATUS_SUCCESS) return st;
                const int RB = (T * B + 127) / 128;
                const SaDrop* dmask = (drop_on && l > 0) ? &dc.drop : nullptr;   // the mask of h_out[l-1], in the epilogue
                const unsigned dstream = dc.stream0 + (unsigned)(l > 0 ? l - 1 : 0);
                // Streamed (layers above the bottom one, side stream available): the next layer's two chains start at the two
                // ENDS of the sequence, so only the end rows have to exist before its recurrence is launched; the rest arrives
                // chunk by chunk from the side stream (XCD-filtered, beside that recurrence, ahead of this layer's weight
                // gradients) while the chains work inwards -- the consumer polls the sentinel this buffer is filled with
                // (gru_bwd_fused_kernel<.., PACKK>).  Chunk 0 = a quarter of the rows on the whole chip.
                // (more, smaller chunks lose: 6 +0.6 ms, 8 +1.4; first chunks of 8 - 28 row blocks per end are within noise: round 3)
                const int nchunk = 4, per_end = (RB / 2 + nchunk - 1) / nchunk;
                if (l > 0 && bi_side && RB >= 16 && issuer.next_counter + nchunk + 4 <= issuer.max_counters) {
                    if (!sentinel_fill(din, ((size_t)T * B + 128) * I, stream)) return CTC_STATUS_MEMOPS_FAILED;
                    int lo = 0, hi = RB;  // row blocks [lo, hi) still owed
                    const int n0 = min(per_end, (hi - lo) / 2);
                    int ends[2] = {lo, hi - n0};
                    st = issuer.input_grad_bi_rows(spb, l, din, 2, ends, n0, stream, 0u, dmask, dstream);
                    if (st != CTC_STATUS_SUCCESS) return st;
                    lo += n0; hi -= n0;
                    if (!g_side.order(stream, g_side.s)) return CTC_STATUS_EXECUTION_FAILED;
                    hipLaunchKernelGGL(side_wait_dispatched_kernel, dim3(1), dim3(1), 0, g_side.s, (const unsigned*)(sync + kSyncReg),
                                       256u * (bi_launches + 1u));  // the NEXT layer's recurrence launch is in
                    while (hi - lo >= 2) {
                        const int n = min(per_end, (hi - lo) / 2);
                        int e2[2] = {lo, hi - n};
                        st = issuer.input_grad_bi_rows(spb, l, din, 2, e2, n, g_side.s, bi_mask, dmask, dstream);
                        if (st != CTC_STATUS_SUCCESS) return st;
                        lo += n; hi -= n;
                    }
                    if (hi > lo) {
                        st = issuer.input_grad_bi_rows(spb, l, din, 1, &lo, hi - lo, g_side.s, bi_mask, dmask, dstream);
                        if (st != CTC_STATUS_SUCCESS) return st;
                    }
                    dx_streamed = true;
                } else {
                    const int zero = 0;
                    st = issuer.input_grad_bi_rows(spb, l, din, 1, &zero, RB, stream, 0u, dmask, dstream);
                    if (st != CTC_STATUS_SUCCESS) return st;
                }
            } else if (din)
                for (int d = 0; d < 2; ++d) {
                    st = gemm_whole(0, 0, T * B, I, 3 * H, dai[l * 2 + d], 3 * H, w_ih[l * 2 + d], I, d ? 1.f : 0.f, din, I,
                                    nullptr, gws, gws_bytes, stream);
                    if (st != CTC_STATUS_SUCCESS) return st;
                }
            if (drop_on && l > 0 && !bi_packg) {  // the layer below fed this one its dropped output: the same mask routes the gradient
                st = sa_dropout_apply_impl(din, din, (size_t)T * B * DH, 0, dc.drop, dc.stream0 + l - 1, stream);
                if (st != CTC_STATUS_SUCCESS) return st;
            }
            if (bi_packg) {  // the layer's products on the operands its recurrence kernel packed: beside the next layer's
                             // recurrence (side stream, filtered) when there is one, else plain on the caller's stream
                const bool side = bi_side && l > 0;
                for (int d = 0; d < 2; ++d) wg_hi[l * 2 + d] = 0;
                if (side && !dx_streamed) {
                    if (!g_side.order(stream, g_side.s)) return CTC_STATUS_EXECUTION_FAILED;
                    hipLaunchKernelGGL(side_wait_dispatched_kernel, dim3(1), dim3(1), 0, g_side.s, (const unsigned*)(sync + kSyncReg),
                                       256u * (bi_launches + 1u));  // the NEXT layer's recurrence launch is in
                } else if (!side && bi_side && !g_side.order(g_side.s, stream)) {  // the scratch operands are the side stream's
                    return CTC_STATUS_EXECUTION_FAILED;
                }
                st = issuer.issue_shared_bi(spb, l, side ? g_side.s : stream, side ? bi_mask : 0u);
                if (st != CTC_STATUS_SUCCESS) return st;
                if (side) {  // layer l - 2 packs into this layer's slot: not before these products have read it
                    slot_read[l & 1] = g_side.ev[g_side.next];
                    g_side.next = (g_side.next + 1) % SideStream::kEvents;
                    if (hipEventRecord(slot_read[l & 1], g_side.s) != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;
                    slot_busy[l & 1] = true;
                }
            } else
            if (bi_xcd && wg && bi_side && l > 0) {  // (layer 0's products have no recurrence left to hide behind: they
                                                      // run unfiltered on the caller's stream, wgrad_rest below)
                // this layer's weight gradients go to the side stream NOW, behind the input-gradient products above
                // (those are on the critical path and need the whole chip: a persistent side GEMM started earlier
                // would hold the idle XCDs' CUs and stall their share of the tiles): they run beside the NEXT layer's
                // persistent launch, on the XCDs it leaves idle
                int spans[2 * kMaxJobs][2];
                for (int k = 0; k < L * 2; ++k) { spans[k][0] = spans[k][1] = 0; }
                for (int d = 0; d < 2; ++d) { spans[l * 2 + d][1] = T; wg_hi[l * 2 + d] = 0; }
                if (!g_side.order(stream, g_side.s)) return CTC_STATUS_EXECUTION_FAILED;
                hipLaunchKernelGGL(side_wait_dispatched_kernel, dim3(1), dim3(1), 0, g_side.s, (const unsigned*)(sync + kSyncReg),
                                       256u * (bi_launches + 1u));  // the NEXT layer's recurrence launch is in
                issuer.xcc_mask = bi_mask;
                st = issuer.issue(spans, g_side.s, true);
                issuer.xcc_mask = 0;
                if (st != CTC_STATUS_SUCCESS) return st;
            }
        }
        SA_CHECK_LAUNCH();
        if (bi_xcd) g_health.submit(stream);
        if (bi_side && !g_side.order(g_side.s, stream)) return CTC_STATUS_EXECUTION_FAILED;  // join
        return wgrad_rest();
    }

    // unidirectional: the wavefront runs top layer first, time chunks from the end
    Chains ch(stream, aux_streams, n_aux, B);
    if (!ch.ok) return CTC_STATUS_EXECUTION_FAILED;
    P.tile_rows = ch.tile_rows;
    const int nch = (T + chunk - 1) / chunk;
    // persistent XCD-local chunk kernel (SA_GRU_PERSIST=2; see gru_bwd_persist_kernel)
    const int nbt = (B + 15) / 16, ntile_u = H / 16;
    const size_t plds = xcd_lds((size_t)2 * 4 * 256 * sizeof(float));
    const bool xcd = ch.n == 1 && xcd_shape_ok(L, B, H) && L * nbt <= kSyncErr && plds <= 160 * 1024 &&
                     (long)T * B * 3 * H * 4 < 0x7fffffffL;
    unsigned persist_launches = 0;
    const bool flagless = xcd && flagless_mode();
    // the lower layers' d h_out inside the recurrence kernel (gru_bwd_fused_kernel): no GEMM between the launches
    const bool tiled = flagless && bwd_fused_fn(H, false) != nullptr && (long)T * nbt * 16 * 3 * H * 4 < 0x7fffffffL;
    const bool fused = tiled && L > 1 && fuse_dx_enabled();
    // ONE launch for the whole backward recurrence of the stack (fused kernel only): every layer runs all T steps, a
    // lower layer picks each row of its d h_out up as the layer above stores it (sentinel pre-fill, see the kernel) --
    // T + a few steps per layer of lag instead of (T / chunk + L - 1) chunks, and no launch ramps in between.
    const bool one_launch = fused && sa_opt(SA_OPT_GRU_BWD_ONE) != 0;  // (r6: the phase clocks run in the one-launch kernel too)
    // ... and that launch writes the weight-gradient products' gate operand itself, packed (gru_bwd_fused_kernel<PACKG>)
    SharedPackLayout spl;
    const bool packg = one_launch && wg && (B % 16) == 0 && sa_opt(SA_OPT_GRU_PACK_IN_KERNEL) != 0 && packg_available(H, true) &&
                       issuer.shared_ok(spl);
    const int xring = tiled && xring_enabled() ? 2 : 1;  // PBwdJobs::packed
    const BwdPersistFn tiled_fn = tiled ? bwd_fused_fn(H, fused, fused && drop_on, packg, xring == 2 && T >= 8) : nullptr;
    const int xgates = tiled_fn != nullptr && bwd_fused_is16(tiled_fn) ? 4 : 3;  // values per element in the exchange ring
    const size_t flds = xcd_lds((size_t)(4 * 4 * 256 + (packg ? 2 * 4 * 16 * 20 : 0)) * sizeof(float));
    auto wih_t_of = [&](int l) { return (float*)(ws + wih_t_off + (size_t)(l - 1) * wih_t_each); };  // l >= 1
    auto xch_of = [&](int l) { return (float*)(ws + xch_off + (size_t)l * xch_each); };
    FillBatch fills(stream, one_launch && fill_batch_enabled());
    if (flagless)
        for (int l = 0; l < L; ++l)  // the exchanged values are their own flags (ring: kXRing time slots, re-armed in the kernel)
            fills.add(tiled ? xch_of(l) : dah[l],
                      tiled ? (size_t)(xring == 2 ? min(T, kXRing) : T) * nbt * 16 * xgates * H : (size_t)T * B * 3 * H, kSentinel);
    if (!one_launch) fills.flush();
    if (!fills.ok) return CTC_STATUS_MEMOPS_FAILED;
    // the one-launch kernel reads its weight fragments from the matrices as stored (PBwdJobs::w_rowmajor): 2 L - 1 transpose
    // launches less per step
    const bool rowmajor = one_launch;
    if (!rowmajor) transpose_whh();
    if (fused && !rowmajor) {
        for (int l = 1; l < L; ++l)
            hipLaunchKernelGGL(transpose_kernel, dim3((H + 31) / 32, (3 * H + 31) / 32), dim3(256), 0, stream, w_ih[l],
                               wih_t_of(l), 3 * H, H);
    }
    if (tiled && hipFuncSetAttribute((const void*)tiled_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds) != hipSuccess)
        return CTC_STATUS_EXECUTION_FAILED;
    // (unidirectional stacks hold all 8 XCDs: weight-gradient GEMMs beside the recurrence slowed both -- 12.1 -> 15.1 ms
    // per step in round 2, profiles/r02_overlap_experiments.txt -- and that side-stream path is retired; the products
    // run behind the recurrence, wgrad_rest below)
    if (xcd) {
        fills.add(sync, (sa_opt(SA_OPT_GRU_TIMING) ? kSyncBytes : 1024) / 4, 0u);
        if (!one_launch) fills.flush();
        if (!fills.ok) return CTC_STATUS_MEMOPS_FAILED;
        if (hipFuncSetAttribute((const void*)bwd_persist_fn(), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)plds) != hipSuccess)
            return CTC_STATUS_EXECUTION_FAILED;
    }
    if (one_launch) {
        for (int l = 0; l + 1 < L; ++l) fills.add(mid_of(l), (size_t)T * B * H, kSentinel);
        fills.flush();
        if (!fills.ok) return CTC_STATUS_MEMOPS_FAILED;
        PBwdJobs Q;
        Q.B = B; Q.H = H; Q.nbt_all = nbt; Q.ntile_u = ntile_u; Q.rb = 1; Q.rt = B; Q.flagless = 1;
        Q.timing = sa_opt(SA_OPT_GRU_TIMING) ? (unsigned long long*)(sync + 512) : nullptr;  // 10 KB of the sync page
        Q.drop = dc.drop;
        Q.pk_kb = packg ? (long)T * B / 16 : 0; Q.kpk_kb = 0;
        Q.err = g_health.dev; Q.spin_limit = spin_limit(); Q.fault = fault_injection(); Q.prio = persist_prio(); Q.packed = xring; Q.reg = sync + kSyncReg; Q.w_rowmajor = rowmajor ? 1 : 0;
        if (packg) { issuer.gates_prepacked = true; issuer.gsum_parts = nbt; }
        for (int l = L - 1, n = 0; l >= 0; --l, ++n) {
            PBwdJob& J = Q.j[n];
            J.dh_out = (l == L - 1) ? dh_top : mid_of(l); J.ds_b = DH; J.ds_t = (long)B * DH;
            J.stash = stash[l]; J.w_hh_t = rowmajor ? w_hh[l] : wt_of(l, 0); J.dai = dai[l]; J.dah = dah[l];
            J.dh_state = dh_buf(l, 0, 0); J.counters = sync + l * nbt;
            J.w_ih_t = l > 0 ? (rowmajor ? w_ih[l] : wih_t_of(l)) : nullptr; J.dx_out = l > 0 ? mid_of(l - 1) : nullptr;
            J.xs_b = DH; J.xs_t = (long)B * DH; J.xch = xch_of(l); J.dump = (float*)(ws + dump_off);
            J.dx_drop_stream = dc.stream0 + (unsigned)(l > 0 ? l - 1 : 0);
            J.gpk = packg ? wws + spl.g_off + (size_t)l * spl.g_each : nullptr;
            J.gsum = packg ? (float*)(wws + spl.cs_off) + (size_t)l * nbt * 4 * H : nullptr;
            J.kpk = nullptr;
            J.t0 = T - 1; J.nsteps = T; J.dt = -1; J.t_first = T - 1; J.base = 0;
        }
        Q.n = L;
        Q.stamp = g_prof.slot(1, true, false, T);
        const int tpp = tiles_per_pass(L, H);
        for (int bt0 = 0; bt0 < nbt; bt0 += tpp) {  // passes over the batch tiles
            Q.bt0 = bt0; Q.nbt = min(tpp, nbt - bt0);
            Q.reg_base = persist_launches++ * 32u;
            if (bt0 > 0) Q.stamp = nullptr;
            hipLaunchKernelGGL(tiled_fn, dim3(256), dim3(256), flds, stream, Q);
        }
    }
    for (int w = 0; !one_launch && w < nch + L - 1; ++w) {
        // d h_out of each lower layer's next chunk = dai of the layer above (finished last wave) times its W_ih
        if (!fused) {
            const float* gA[kMaxJobs]; const float* gB[kMaxJobs]; float* gC[kMaxJobs];
            int gL[kMaxJobs], gT[kMaxJobs];
            int ng = 0;
            for (int l = L - 2; l >= 0; --l) {
                const int cc = w - (L - 1 - l);
                if (cc < 0 || cc >= nch) continue;
                const int c = nch - 1 - cc;
                const int t0 = c * chunk, t1 = min(T, t0 + chunk);
                const float* Ap = dai[l + 1] + (long)t0 * B * 3 * H;
                float* Cp = mid_of(l) + (long)t0 * B * H;
                if (t1 - t0 == chunk) {
                    gA[ng] = Ap; gB[ng] = w_ih[l + 1]; gC[ng] = Cp; gL[ng] = l; gT[ng] = t0; ++ng;
                } else {
                    st = sa_gemm_f32_impl(0, 0, (t1 - t0) * B, H, 3 * H, 1.f, Ap, 3 * H, w_ih[l + 1], H, 0.f, Cp, H,
                                          nullptr, nullptr, gws, gws_bytes, stream);
                    if (st != CTC_STATUS_SUCCESS) return st;
                    if (drop_on) {  // layer l+1 read layer l's DROPPED output: the same mask routes the gradient
                        st = sa_dropout_apply_impl(Cp, Cp, (size_t)(t1 - t0) * B * H, (size_t)t0 * B * H, dc.drop, dc.stream0 + l, stream);
                        if (st != CTC_STATUS_SUCCESS) return st;
                    }
                }
            }
            if (ng > 0) {
                st = sa_gemm_f32_group_impl(ng, 0, 0, chunk * B, H, 3 * H, 1.f, gA, 3 * H, gB, H, 0.f, gC, H, nullptr,
                                            nullptr, gws, gws_bytes, stream);
                if (st != CTC_STATUS_SUCCESS) return st;
                for (int k = 0; drop_on && k < ng; ++k) {
                    st = sa_dropout_apply_impl(gC[k], gC[k], (size_t)chunk * B * H, (size_t)gT[k] * B * H, dc.drop, dc.stream0 + gL[k], stream);
                    if (st != CTC_STATUS_SUCCESS) return st;
                }
            }
        }
        if (xcd) {  // ONE launch unwinds the whole chunk of every active layer
            PBwdJobs Q;
            Q.B = B; Q.H = H; Q.nbt_all = nbt; Q.ntile_u = ntile_u; Q.rb = 1; Q.rt = B; Q.flagless = flagless ? 1 : 0;
            Q.timing = sa_opt(SA_OPT_GRU_TIMING) ? (unsigned long long*)(sync + 512) : nullptr;  // 10 KB of the sync page
            Q.pk_kb = 0; Q.kpk_kb = 0;
            Q.drop = dc.drop;
            Q.err = g_health.dev; Q.spin_limit = spin_limit(); Q.fault = fault_injection(); Q.prio = persist_prio(); Q.packed = xring; Q.reg = sync + kSyncReg; Q.w_rowmajor = 0;
            int n = 0;
            for (int l = L - 1; l >= 0; --l) {
                const int cc = w - (L - 1 - l);
                if (cc < 0 || cc >= nch) continue;
                const int c = nch - 1 - cc;
                PBwdJob& J = Q.j[n++];
                J.dh_out = (l == L - 1) ? dh_top : mid_of(l); J.ds_b = DH; J.ds_t = (long)B * DH;
                J.stash = stash[l]; J.w_hh_t = wt_of(l, 0); J.dai = dai[l]; J.dah = dah[l];
                J.dh_state = dh_buf(l, 0, 0); J.counters = sync + l * nbt;
                J.w_ih_t = fused && l > 0 ? wih_t_of(l) : nullptr; J.dx_out = fused && l > 0 ? mid_of(l - 1) : nullptr;
                J.xs_b = DH; J.xs_t = (long)B * DH; J.xch = tiled ? xch_of(l) : nullptr; J.dump = (float*)(ws + dump_off);
                J.gpk = nullptr; J.gsum = nullptr; J.kpk = nullptr;
                J.dx_drop_stream = dc.stream0 + (unsigned)(l > 0 ? l - 1 : 0);
                J.t0 = min(T, (c + 1) * chunk) - 1; J.nsteps = J.t0 - c * chunk + 1; J.dt = -1; J.t_first = T - 1;
                J.base = (unsigned)ntile_u * (unsigned)(T - 1 - J.t0);
            }
            Q.n = n;
            Q.stamp = g_prof.slot(1, n == L, false, chunk);
            {
                const int tpp = tiles_per_pass(L, H);
                for (int bt0 = 0; bt0 < nbt; bt0 += tpp) {  // passes over the batch tiles
                    Q.bt0 = bt0; Q.nbt = min(tpp, nbt - bt0);
                    Q.reg_base = persist_launches++ * 32u;
                    if (bt0 > 0) Q.stamp = nullptr;
                    if (tiled) hipLaunchKernelGGL(tiled_fn, dim3(256), dim3(256), flds, stream, Q);
                    else hipLaunchKernelGGL(bwd_persist_fn(), dim3(256), dim3(256), plds, stream, Q);
                }
            }
            continue;
        }
        ch.fork();
        for (int s = 0; s < chunk; ++s) {
            int n = 0;
            for (int l = L - 1; l >= 0; --l) {
                const int cc = w - (L - 1 - l);
                if (cc < 0 || cc >= nch) continue;
                const int c = nch - 1 - cc;
                const int t = c * chunk + (chunk - 1 - s);
                if (t >= T) continue;
                P.j[n++] = make_job(l, 0, t, t == T - 1 ? -1 : t + 1, T - 1 - t);
            }
            if (n == 0) continue;
            P.n = n; grid.z = n;
            for (int k = 0; k < ch.n; ++k) {
                P.bt0 = ch.bt0[k]; grid.y = ch.nbt[k];
                P.stamp = k == 0 ? g_prof.slot(1, n == L, s > 0) : nullptr;
                hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(256), 0, ch.s[k], P);
            }
        }
        ch.join();
    }
    SA_CHECK_LAUNCH();
    if (xcd) g_health.submit(stream);
    if (dx) {
        st = gemm_whole(0, 0, T * B, I0, 3 * H, dai[0], 3 * H, w_ih[0], I0, 0.f, dx, I0, nullptr, gws, gws_bytes, stream);
        if (st != CTC_STATUS_SUCCESS) return st;
    }
    return wgrad_rest();
}
}  // namespace

extern "C" ctcStatus_t sa_gru_stack_bwd(const float* dh_top, const float* const* stash, const float* const* w_ih,
                                        const float* const* w_hh, float* const* dai, float* const* dah, float* dx,
                                        int I0, int L, int D, int B, int T, int H, int chunk, void* workspace,
                                        size_t workspace_bytes, void* stream_, void* const* aux_streams, int n_aux) {
    const DropCtx dc{sa_drop_make(0.f, 0ull), 0u, nullptr};
    return stack_bwd_impl(dh_top, stash, w_ih, w_hh, dai, dah, dx, I0, L, D, B, T, H, chunk, workspace,
                          workspace_bytes, stream_, aux_streams, n_aux, nullptr, dc);
}

extern "C" ctcStatus_t sa_gru_stack_bwd_wgrad(const float* dh_top, const float* const* stash,
                                              const float* const* w_ih, const float* const* w_hh, float* const* dai,
                                              float* const* dah, float* dx, int I0, int L, int D, int B, int T, int H,
                                              int chunk, const float* x, const float* const* h_out,
                                              float* const* dw_ih, float* const* dw_hh, float* const* db_ih,
                                              float* const* db_hh, void* workspace, size_t workspace_bytes,
                                              void* stream_) {
    if (!x || !h_out || !dw_ih || !dw_hh || !db_ih || !db_hh) return CTC_STATUS_INVALID_VALUE;
    for (int k = 0; k < L * D && k < 2 * kMaxJobs; ++k)
        if (!dw_ih[k] || !dw_hh[k] || !db_ih[k] || !db_hh[k]) return CTC_STATUS_INVALID_VALUE;
    WGrad wg{x, h_out, dw_ih, dw_hh, db_ih, db_hh};
    const DropCtx dc{sa_drop_make(0.f, 0ull), 0u, nullptr};
    return stack_bwd_impl(dh_top, stash, w_ih, w_hh, dai, dah, dx, I0, L, D, B, T, H, chunk, workspace,
                          workspace_bytes, stream_, nullptr, 0, &wg, dc);
}

// The backward pass of sa_gru_stack_fwd_dropout (same p, seed, mask_stream0, and the h_drop buffers it filled).
extern "C" ctcStatus_t sa_gru_stack_bwd_wgrad_dropout(const float* dh_top, const float* const* stash,
                                                      const float* const* w_ih, const float* const* w_hh,
                                                      float* const* dai, float* const* dah, float* dx, int I0, int L,
                                                      int D, int B, int T, int H, int chunk, const float* x,
                                                      const float* const* h_out, float* const* h_drop,
                                                      float* const* dw_ih, float* const* dw_hh, float* const* db_ih,
                                                      float* const* db_hh, void* workspace, size_t workspace_bytes,
                                                      float p, unsigned long long seed, unsigned int mask_stream0,
                                                      void* stream_) {
    if (!x || !h_out || !dw_ih || !dw_hh || !db_ih || !db_hh || !sa_drop_valid(p)) return CTC_STATUS_INVALID_VALUE;
    for (int k = 0; k < L * D && k < 2 * kMaxJobs; ++k)
        if (!dw_ih[k] || !dw_hh[k] || !db_ih[k] || !db_hh[k]) return CTC_STATUS_INVALID_VALUE;
    WGrad wg{x, h_out, dw_ih, dw_hh, db_ih, db_hh};
    const DropCtx dc{sa_drop_make(p, seed), mask_stream0, h_drop};
    return stack_bwd_impl(dh_top, stash, w_ih, w_hh, dai, dah, dx, I0, L, D, B, T, H, chunk, workspace,
                          workspace_bytes, stream_, nullptr, 0, &wg, dc);
}

extern "C" size_t sa_colsum_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)colsum_parts(M) * N * sizeof(float);
}

extern "C" ctcStatus_t sa_colsum_f32(const float* a, long lda, int M, int N, float* out, int accumulate,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    if (!a || !out || M < 0 || N <= 0) return CTC_STATUS_INVALID_VALUE;
    const int R = colsum_parts(M > 0 ? M : 1);
    if (!workspace || workspace_bytes < (size_t)R * N * sizeof(float)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    const int rows_per = (M + R - 1) / R;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, R), dim3(256), 0, stream, a, lda, M, N,
                       rows_per > 0 ? rows_per : 1, (float*)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, (const float*)workspace, R, N,
                       out, accumulate);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_add_rows_f32(const float* a, long lda, const float* b, long ldb, float* y, long ldy,
                                       int rows, int cols, void* stream_) {
    SA_CLEAR_ERR();
    if (!a || !b || !y || rows < 0 || cols < 0) return CTC_STATUS_INVALID_VALUE;
    const long total = (long)rows * cols;
    if (total == 0) return CTC_STATUS_SUCCESS;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a,
                       lda, b, ldb, y, ldy, rows, cols);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
