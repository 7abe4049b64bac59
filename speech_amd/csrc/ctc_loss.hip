// ctc_loss.hip -- CTC loss forward (alpha) / backward (beta) / gradient for gfx950.
//
// Replaces the warp-ctc extension the reference calls at /root/reference/speech/models/ctc_model.py:38-39
// (functions.ctc.CTCLoss; out-of-tree, see include/speech_amd.h section 1).  Algorithm: SURVEY.md Appendix A.
//
// Design (MI355X-first, not a translation of warp-ctc's per-utterance thread-block kernels):
//   K_A  ctc_logsoftmax2   all rows in parallel: ly2[b,t,k] = log2 softmax(acts[b,t,:])[k]  (log2 domain, so
//                          the recurrences run on bare v_exp_f32 / v_log_f32).
//   K_B  ctc_alphabeta     one workgroup per utterance.  The 2L+1 lattice states are paired per lane
//                          (blank_j, label_j): a time step needs ONE cross-lane value (the neighbour pair's label
//                          state, one DPP wave_shr/shl).  64 pairs per wave ("chunk").  alpha dependencies only
//                          flow from lower to higher states, beta the other way, so the chunks of one direction form
//                          a one-way software pipeline: the producing chunk runs kU steps ahead and hands its edge
//                          lane's value to the next chunk through an LDS array -- no workgroup barrier inside the
//                          T-step loop.  alpha waves and beta waves run concurrently in the same workgroup (the
//                          serial chain is T steps, not 2T) and stash alpha/beta planes to HBM with fire-and-forget
//                          stores.  The utterance's emission rows are staged in LDS once, so the T-step loop issues
//                          no VGPR-destination VMEM load: on gfx9 loads and stores retire through one in-order
//                          vmcnt queue, and a waited load behind a store would put HBM store latency on every step.
//                          Every kU steps a wave subtracts the integer part of its running maximum from its states
//                          (an exact operation) and carries the integer offset separately, so alpha/beta stay O(100)
//                          where fp32 resolves ~1e-5 instead of O(|log2 p|) ~ 4000 where it resolves 5e-4.
//   K_C  ctc_grad          all rows in parallel again: occupancy gamma = exp2(alpha + beta - ly2 - log2 p),
//                          blank states by a DPP wave reduction, label states by LDS float atomics,
//                          grad = softmax - occupancy written with the caller's strides.
// log(0) is the finite sentinel SA_NEG, so no inf/NaN guards sit on the dependent chain.
// HIPCC_FLAGS: -fno-honor-nans
// (no NaN is ever an operand here -- log(0) is the finite SA_NEG -- so fmaxf/fminf need no canonicalising v_max.)
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "internal.h"

namespace {

constexpr int kU = 8;          // steps per hand-off batch (producer lead, emission prefetch depth, renorm period)
constexpr int kRenorm = 4;     // renormalise every kRenorm batches (32 steps): states drift <~ 200 log2 units between
constexpr int kMaxChunks = 8;  // chunks per direction: labels up to 64 * 8 - 1 = 511 per utterance
constexpr int kWideMinBatch = 512;  // from this many utterances per call the one-wave-per-utterance kernel (K_W) runs
constexpr int kDirectMinBatch = 1024;  // ... and from this many its probability-domain pass reads the activations itself

// ---------------------------------------------------------------------------------------------------------- K_A
// grid (row blocks, B).  G lanes cooperate on one row (G = 16, 32 or 64); a wave handles 64 / G rows at a time.
template <int G>
__global__ __launch_bounds__(256) void ctc_logsoftmax2_kernel(const float* __restrict__ acts, long st, long sb,
                                                              const int* __restrict__ in_lens, int K, long ly_sb,
                                                              float* __restrict__ ly2, const int* __restrict__ only) {
    constexpr int RPW = 64 / G;
    const int b = blockIdx.y;
    if (only && !only[b]) return;  // (the pass behind ctc_wave_p_kernel: flagged utterances only)
    const int T = in_lens[b];
    const int lane = threadIdx.x & 63;
    const int sub = lane / G, gl = lane % G;
    const int t = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
    const bool act = t < T;
    const float* a = acts + (long)b * sb + (long)t * st;
    float m = -3.0e38f;
    if (act)
        for (int k = gl; k < K; k += G) m = fmaxf(m, a[k]);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float z = 0.f;
    if (act)
        for (int k = gl; k < K; k += G) z += sa_exp2((a[k] - m) * SA_LOG2E);
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
    const float lz = sa_log2(z);
    if (act) {
        float* o = ly2 + (long)b * ly_sb + (long)t * K;
        for (int k = gl; k < K; k += G) o[k] = fmaxf((a[k] - m) * SA_LOG2E - lz, SA_NEG);
    }
}

// K_A for small alphabets (K <= 64, i.e. every character-level model of the reference): a workgroup stages kRowTile rows
// in LDS with coalesced loads, then ONE LANE normalises ONE ROW out of LDS (row stride odd -> conflict-free) and the
// tile is written back coalesced.  ~8 wave instructions per row instead of ~50 for the lane-group kernel above
// (whose two butterfly reductions per row dominate at K = 29), which makes K_A HBM-bound at large B.
constexpr int kRowTile = 64;  // rows (= lanes) per workgroup: one wave, so its phases need no cross-wave barrier and
                              // a CU interleaves ~20 independent tiles
__global__ __launch_bounds__(kRowTile) void ctc_logsoftmax2_rows_kernel(const float* __restrict__ acts, long st, long sb,
                                                                   const int* __restrict__ in_lens, int K, long ly_sb,
                                                                   float* __restrict__ ly2, const int* __restrict__ only) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);  // [kRowTile][KS]
    const int KS = K | 1;
    const int b = blockIdx.y;
    if (only && !only[b]) return;
    const int T = in_lens[b];
    const int t0 = blockIdx.x * kRowTile;
    const int nrows = min(kRowTile, T - t0);
    if (nrows <= 0) return;
    const int n = nrows * K;
    const int dr = kRowTile / K, dk = kRowTile - dr * K;  // idx += kRowTile -> (row, k) += (dr, dk), one carry
    const float* src = acts + (long)b * sb + (long)t0 * st;
    {
        int r = threadIdx.x / K, k = threadIdx.x - r * K;
        for (int i = threadIdx.x; i < n; i += kRowTile) {
            tile[r * KS + k] = src[(long)r * st + k];
            r += dr; k += dk;
            if (k >= K) { k -= K; ++r; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nrows) {
        float* row = tile + threadIdx.x * KS;
        float m = -3.0e38f;
        for (int k = 0; k < K; ++k) m = fmaxf(m, row[k]);
        float z = 0.f;
        for (int k = 0; k < K; ++k) z += sa_exp2((row[k] - m) * SA_LOG2E);
        const float lz = sa_log2(z);
        for (int k = 0; k < K; ++k) row[k] = fmaxf((row[k] - m) * SA_LOG2E - lz, SA_NEG);
    }
    __syncthreads();
    {
        float* dst = ly2 + (long)b * ly_sb + (long)t0 * K;
        int r = threadIdx.x / K, k = threadIdx.x - r * K;
        for (int i = threadIdx.x; i < n; i += kRowTile) {
            dst[i] = tile[r * KS + k];
            r += dr; k += dk;
            if (k >= K) { k -= K; ++r; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- K_B
struct AbShared {
    int prog[2][kMaxChunks];  // steps completed by (direction, chunk)
    float fin[4];             // alpha-hat[T-1, 2L], its offset, alpha-hat[T-1, 2L-1], its offset
    int label_off;
    int timeout;
    int suspect;              // probability-domain chain: a wave saw its states lose more range than the pass tolerates
    int nrep;                 // labels equal to their predecessor: each costs the lattice's front one more step
};
constexpr int kAbSharedBytes = 256;

__device__ __forceinline__ float lse2_1p(float a, float b) {  // log2(2^a + 2^b)
    const float m = fmaxf(a, b);
    return m + sa_log2(1.0f + sa_exp2(fminf(a, b) - m));
}

// Inclusive prefix sum over the 64 lanes on the DPP network (the same six steps as sa_wave_sum_dpp, whose lane 63
// holds the total): row_shr 1/2/4/8 scan each 16-lane row, row_bcast15 / row_bcast31 carry the row totals forward.
__device__ __forceinline__ float wave_scan_dpp(float v) {
    v += SA_DPP_F(0.f, v, 0x111, 0xf);
    v += SA_DPP_F(0.f, v, 0x112, 0xf);
    v += SA_DPP_F(0.f, v, 0x114, 0xf);
    v += SA_DPP_F(0.f, v, 0x118, 0xf);
    v += SA_DPP_F(0.f, v, 0x142, 0xa);
    v += SA_DPP_F(0.f, v, 0x143, 0xc);
    return v;
}

// Exclusive prefix (suffix) maximum over the 64 lanes of an int: lane i receives max(v[0 .. i-1]) (max(v[i+1 .. 63])), the
// lane without a source `none`.  In-row scans on the DPP network (row_shr / row_shl 1, 2, 4, 8), the three row totals through
// SGPRs (v_readlane), one wave shift: ~20 instructions, once per frozen batch (ctc_chain_p / wavep_refresh: exponent lifting).
#define sa_dpp_i(old_, src_, ctrl_) __builtin_amdgcn_update_dpp((old_), (src_), (ctrl_), 0xf, 0xf, false)
__device__ __forceinline__ int wave_prefix_max_excl(int v, int none, int lane) {
    int x = v;
    x = max(x, sa_dpp_i(none, x, 0x111));
    x = max(x, sa_dpp_i(none, x, 0x112));
    x = max(x, sa_dpp_i(none, x, 0x114));
    x = max(x, sa_dpp_i(none, x, 0x118));
    const int t0 = __builtin_amdgcn_readlane(x, 15);
    const int t1 = max(t0, __builtin_amdgcn_readlane(x, 31));
    const int t2 = max(t1, __builtin_amdgcn_readlane(x, 47));
    const int row = lane >> 4;
    const int before = row == 0 ? none : (row == 1 ? t0 : (row == 2 ? t1 : t2));
    x = max(x, before);  // inclusive
    return __builtin_amdgcn_update_dpp(none, x, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ int wave_suffix_max_excl(int v, int none, int lane) {
    int x = v;
    x = max(x, sa_dpp_i(none, x, 0x101));
    x = max(x, sa_dpp_i(none, x, 0x102));
    x = max(x, sa_dpp_i(none, x, 0x104));
    x = max(x, sa_dpp_i(none, x, 0x108));
    const int t3 = __builtin_amdgcn_readlane(x, 48);
    const int t2 = max(t3, __builtin_amdgcn_readlane(x, 32));
    const int t1 = max(t2, __builtin_amdgcn_readlane(x, 16));
    const int row = lane >> 4;
    const int after = row == 3 ? none : (row == 2 ? t3 : (row == 1 ? t2 : t1));
    x = max(x, after);  // inclusive
    return __builtin_amdgcn_update_dpp(none, x, 0x130, 0xf, 0xf, false);  // wave_shl:1
}

// Hand-off arrays exist only for the chunks that HAVE a consumer: alpha chunks 0 .. n-2, beta chunks 1 .. n-1 -- n - 1 per
// direction (none for a one-chunk lattice): with all 2n of them the emission copy stopped fitting LDS from three chunks.
__device__ __forceinline__ int ctc_hand_slot(int dir, int chunk, int nchunks) {
    return dir * (nchunks - 1) + (dir == 0 ? chunk : chunk - 1);
}

struct AbArgs {
    const float* ly2;     // [B][T_max][K] global
    const int* labels;
    const int* label_lens;
    const int* in_lens;
    int K, T_max, blank, nchunks, Ppad, hand_stride, nbatch;
    long ly_sb;           // per-utterance stride of ly2 in floats (multiple of 4: 16-byte aligned rows block)
    float* stash;         // [B][T_max][6][Ppad]
    float* goffs;         // [B][2][nchunks][nbatch]
    float* logp2_out;     // [B][2]: log2 p = hat + offset
    float* costs;         // [B]
    int* lsort;           // [B][1 + Ppad + K]: {flat-label offset | sorted slot of label j | end of class k's run}: the
                          // utterance's label states counting-sorted by class, once, by the alpha / beta kernel, so
                          // that ctc_grad_kernel folds the occupancies of a class with a prefix sum (no atomics)
    int* flags;           // [B]: nonzero = the probability-domain pass may have lost mass for this utterance (see
                          // ctc_chain_p): the log-domain kernels, launched behind it with gate = 1, redo exactly those
    int no_fast;          // debug / tests ((retired switch)): keep ctc_chain_p in its per-step alignment (phase 1) throughout
    int gate;             // 1: process only the utterances whose flag is set -- and finish them here, gradient rows
                          // included (ONE launch behind the probability-domain pass, returning at once when nothing is flagged)
    float* grads;         // gate only: where ctc_grad_row writes
    long g_st, g_sb;
    float gscale;         // every gradient element is multiplied by this (the caller's 1 / batch size)
    unsigned long long* dbg;  // debug (option ctc.dbg): wave 0 of block 0 stores {shader cycles, 100 MHz ticks} of its T loop
    unsigned long long* prof; // sa_ctc_profile_*: {earliest workgroup entry, latest workgroup exit} of this launch (100 MHz ticks) or null
    // (r6) the batch reduction of the reduced form folded into the gated launch: the LAST workgroup of that launch to finish
    // (a device-scope arrival counter, zeroed by the probability-domain launch in front of it and again by its last arrival)
    // sums the costs in ctc_cost_sum_kernel's order and writes *loss_out = loss_scale * sum -- one launch fewer per loss
    float* loss_out;          // or null
    float loss_scale;
    unsigned* done_ctr;       // one word of the workspace (behind the flags)
};

// One (direction, chunk) wave over all T steps.  DIR 0 = alpha (time forward), 1 = beta (time backward).
template <int DIR, bool WITH_BETA, bool LDS_EM>
__device__ __forceinline__ void ctc_chain(const AbArgs& A, AbShared* sh, float* hand_all, float* hoff_all,
                                          const float* em_lds, int b, int chunk, int lane, int L, int T) {
    const int nchunks = A.nchunks, K = A.K, Ppad = A.Ppad;
    const int* lab = A.labels + sh->label_off;
    const int j = chunk * 64 + lane;  // pair index: (blank_j, label_j) for alpha, (label_{j-1}, blank_j) for beta
    const int own = DIR == 0 ? j : j - 1;
    const bool own_ok = own >= 0 && own < L;
    const int own_lab = own_ok ? lab[own] : A.blank;
    const bool skip = (j >= 1 && j <= L - 1) ? (lab[j] != lab[j - 1]) : false;

    const int prod_chunk = DIR == 0 ? chunk - 1 : chunk + 1;
    const bool has_prod = prod_chunk >= 0 && prod_chunk < nchunks;
    const bool has_cons = DIR == 0 ? (chunk + 1 < nchunks) : (chunk > 0);
    // (a chunk without a consumer never writes, one without a producer never reads: their slot index is clamped, unused)
    const int slot_w = has_cons ? ctc_hand_slot(DIR, chunk, nchunks) : 0;
    const int slot_r = has_prod ? ctc_hand_slot(DIR, prod_chunk, nchunks) : 0;
    float* hand_w = hand_all + (long)slot_w * A.hand_stride;
    float* hoff_w = hoff_all + (long)slot_w * A.nbatch;
    const float* hand_r = hand_all + (long)slot_r * A.hand_stride;
    const float* hoff_r = hoff_all + (long)slot_r * A.nbatch;
    constexpr int edge_lane = DIR == 0 ? 63 : 0;

    float Bst = (DIR == 0 ? (j == 0) : (j == L)) ? 0.0f : SA_NEG;
    float Lst = SA_NEG;
    float off = 0.f;  // integer-valued: true state = hat state + off

    // time moves forward for alpha, backward for beta
    const int t0 = DIR == 0 ? 0 : (T > 0 ? T - 1 : 0);
    const int dK = DIR == 0 ? K : -K;
    const long dS = DIR == 0 ? 6L * Ppad : -6L * Ppad;  // 6 planes per (b, t): the last two are the probability-domain chain's
    const float* em = LDS_EM ? em_lds : A.ly2 + (long)b * A.ly_sb;
    int e_l = t0 * K + own_lab;   // element index of the next step's label emission
    int e_b = t0 * K + A.blank;
    float* pB = nullptr;
    float* pL = nullptr;
    float* goff = nullptr;
    if (WITH_BETA) {
        // stash planes per (b, t): [alpha blank | alpha label | beta blank | beta label], Ppad floats each.
        // beta's label state belongs to pair j-1; lanes without a label state dump into a never-read slot
        // (label index Ppad-1 cannot exist because Ppad >= L+1).
        float* base = A.stash + ((long)b * A.T_max + t0) * 6 * Ppad;
        pB = base + (DIR == 0 ? 0 : 2) * Ppad + j;
        pL = base + (DIR == 0 ? 1 : 3) * Ppad + (own >= 0 ? own : Ppad - 1);
        goff = A.goffs + ((long)(b * 2 + DIR) * nchunks + chunk) * A.nbatch;
    }

    float el[kU], eb[kU];
    auto load_emissions = [&](int r0, float* pel, float* peb) {
        if (r0 + kU <= T) {
#pragma unroll
            for (int k = 0; k < kU; ++k) {
                peb[k] = em[e_b];
                pel[k] = em[e_l];
                e_b += dK;
                e_l += dK;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kU; ++k) {
                const bool ok = r0 + k < T;
                peb[k] = ok ? em[e_b] : 0.f;
                pel[k] = ok ? em[e_l] : 0.f;
                if (ok) { e_b += dK; e_l += dK; }
            }
        }
    };
    load_emissions(0, el, eb);

    // One time step of this wave's 64 state pairs.  Both updates share ONE pivot m = max3(L, B, n): the three exp2 and
    // the two log2 are then mutually independent (dependent chain: dpp -> max3 -> sub -> exp2 -> add -> log2 -> add),
    // instead of two back-to-back log-sum-exp chains.  A sum can flush to 0 only when its own operands sit more than
    // 126 log2-units below the pivot; it is then replaced by the larger operand (exact for the unreachable sentinel,
    // so infeasible alignments stay infeasible; otherwise an error below 2^-126 of the neighbouring state's mass).
    auto do_step = [&](float e_l_raw, float e_b_, float h, int r) {
        const float n = DIR == 0 ? sa_wave_shr1(Lst, h) : sa_wave_shl1(Lst, h);
        const float e_lv = own_ok ? e_l_raw : SA_NEG;
        const float mLB = fmaxf(Lst, Bst);
        const float mBn = fmaxf(Bst, n);
        const float m = fmaxf(mLB, n);
        const float xL = sa_exp2(Lst - m);
        const float xB = sa_exp2(Bst - m);
        const float xn = sa_exp2(n - m);
        const float sB = xB + xn;
        const float sL = xL + xB + (skip ? xn : 0.f);
        const float nB = e_b_ + (sB > 0.f ? m + sa_log2(sB) : mBn);
        const float nL = e_lv + (sL > 0.f ? m + sa_log2(sL) : mLB);
        Bst = nB;
        Lst = nL;
        if (WITH_BETA) {
            *pB = Bst;
            *pL = Lst;
            pB += dS;
            pL += dS;
        }
        if (has_cons && lane == edge_lane) hand_w[r + 1] = Lst;
    };

    int avail = 0;  // producer progress last seen
    int bi = 0;     // batch index
    const bool dbg_me = A.dbg != nullptr && b == 0 && DIR == 0 && chunk == 0 && lane == 0;
    unsigned long long c0 = 0, w0 = 0;
    if (dbg_me) { c0 = clock64(); w0 = wall_clock64(); }
    for (int r0 = 0; r0 < T; r0 += kU, ++bi) {
        // (1) every kRenorm batches: subtract the integer part of the wave maximum (exact in fp32), carry it in `off`
        if ((bi & (kRenorm - 1)) == 0) {
            const float m = sa_wave_max_dpp(fmaxf(Bst, Lst));
            const float d = m > SA_NEG_TEST ? __builtin_rintf(m) : 0.f;
            Bst -= d;
            Lst -= d;
            off += d;
            if (has_cons && lane == edge_lane) hand_w[r0] = Lst;  // re-express the carried edge value
        }
        if (has_cons && lane == 0) hoff_w[bi] = off;
        if (WITH_BETA && lane == 0) goff[bi] = off;
        float nel[kU], neb[kU];
        load_emissions(r0 + kU, nel, neb);  // prefetch the next batch's emissions (LDS)

        // (2) the neighbour chunk's edge values for this batch, re-based to this wave's offset
        float hv[kU];
        if (has_prod) {
            const int need = min(r0 + kU, T);
            if (avail < need) {
                int spins = 0;
                while ((avail = __hip_atomic_load(&sh->prog[DIR][prod_chunk], __ATOMIC_ACQUIRE,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) {  // bounded: never hang the GPU on a protocol bug
                        sh->timeout = 1;
                        break;
                    }
                }
            }
            const float4 h0 = *reinterpret_cast<const float4*>(hand_r + r0);
            const float4 h1 = *reinterpret_cast<const float4*>(hand_r + r0 + 4);
            const float delta = hoff_r[bi] - off;  // integers: exact
            hv[0] = h0.x + delta; hv[1] = h0.y + delta; hv[2] = h0.z + delta; hv[3] = h0.w + delta;
            hv[4] = h1.x + delta; hv[5] = h1.y + delta; hv[6] = h1.z + delta; hv[7] = h1.w + delta;
        } else {
#pragma unroll
            for (int k = 0; k < kU; ++k) hv[k] = SA_NEG;
        }

        // (3) kU dependent steps
        if (r0 + kU <= T) {
#pragma unroll
            for (int k = 0; k < kU; ++k) do_step(el[k], eb[k], hv[k], r0 + k);
        } else {
#pragma unroll
            for (int k = 0; k < kU; ++k)
                if (r0 + k < T) do_step(el[k], eb[k], hv[k], r0 + k);
        }
        if (has_cons && lane == 0)
            __hip_atomic_store(&sh->prog[DIR][chunk], min(r0 + kU, T), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 0; k < kU; ++k) { el[k] = nel[k]; eb[k] = neb[k]; }
    }

    if (dbg_me) { A.dbg[0] = clock64() - c0; A.dbg[1] = wall_clock64() - w0; A.dbg[2] = (unsigned long long)T; }
    if (DIR == 0) {
        if (j == L) { sh->fin[0] = Bst; sh->fin[1] = off; }
        if (j == L - 1) { sh->fin[2] = Lst; sh->fin[3] = off; }
    }
}

// ---- the same chain in the PROBABILITY domain (the default; option ctc.prob = 0 keeps the log-domain chain above) ----
// alpha_t(s) = (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [skip] alpha_{t-1}(s-2)) y_t(s): adds and multiplies, no exp2 / log2
// on the 1000-step dependent chain (the log-domain step above is ~27 mostly dependent instructions, ~240 cycles).  The
// emissions are converted once (exp2 while they are staged in LDS).
// Range.  A lattice with T >> L spreads its states over hundreds of bits (alpha_t of "k labels emitted" goes like C(t, k):
// C(500, 100) = 2^355), and the states that carry the posterior are NOT the largest ones, so one exponent per wave is not
// enough (measured: with one, M-CTC loses mass from T ~ 750 on).  Every LANE therefore carries an integer exponent e for
// its state pair (true value = hat * 2^e):
//   * a step aligns the own pair and the value arriving from the neighbour lane to max(e, e_neighbour) -- three ldexp, the
//     exponent travels beside the value on a second DPP; what is shifted out is below 2^-126 of the larger of two ADJACENT
//     states, which differ by a few bits.  A pair no mass has reached yet carries the exponent "nothing" and adopts the
//     neighbour's when the front arrives;
//   * every kPRenorm steps a lane brings its pair's larger state back to 2^kPTarget (an exact power-of-two shift), so
//     between two of them a pair may lose 2^-(126 + kPTarget) -- 56 bits per step -- before anything is flushed, and
//     grows by at most 3^kPRenorm.
// What can still go wrong is a pair decaying faster than that (every label in reach below 2^-56 for steps on end), or two
// adjacent states more than 2^126 apart of which the smaller matters later -- emissions far beyond any model's.  The pass
// is therefore CERTIFIED rather than trusted: ctc_grad_kernel checks flow conservation, sum_s alpha_t(s) beta_t(s) / y_t(s)
// = p for every row t (a lost path shows up as rows that no longer sum to one; so does a NaN), and flags the utterance
// otherwise; so does an utterance the pass finds infeasible.  Flagged utterances are recomputed by the log-domain kernels
// launched right behind (they return at once when no flag is set).  test_gpu_ctc.py drives both paths and the hand-over.
// Against an fp64 evaluation of the same recurrences the pass is the MORE accurate of the two: M-CTC max |gradient error| 7e-6 (log domain: 3e-4,
// whose states sit at |log2 p| ~ 5000 where fp32 resolves 5e-4).
constexpr int kPTarget = 100;
constexpr int kPRenorm = 4;
constexpr int kFTarget = 10;    // phase 2: a pair is re-normalised to 2^kFTarget at every batch ...
constexpr int kFMaxShift = 90;  // ... and may sit up to 2^kFMaxShift below its neighbour (2^(10 + 13 + 90) < 2^127)
#ifndef SA_KFLIFT
#define SA_KFLIFT 32
#endif
constexpr int kFLift = SA_KFLIFT;  // ... but is LIFTED to 2^kFLift below it when the batch starts (round 5): on the logits a trained
                                // model emits consecutive pairs ahead of the alignment differ by 2^29 per frame, mass that moves two
                                // pairs inside one frozen batch then multiplies two such shifts -- 2^(10 + 90 + 90) overflowed,
                                // UNNOTICED (-inf / garbage costs on 4 of 6 margin-20 utterances, tests/test_gpu_ctc.py) -- and a pair
                                // whose own mass is 2^-32 of what is about to arrive can drop it (fp32 resolves 2^-24)

template <int DIR, bool WITH_BETA, bool LDS_EM, bool HAS_PROD, bool HAS_CONS>  // a chunk before / after this one in the pipeline
__device__ __forceinline__ void ctc_chain_p(const AbArgs& A, AbShared* sh, float* hand_all, float* hdummy,
                                            const float* em_lds, int b, int chunk, int lane, int L, int T) {
    const int nchunks = A.nchunks, K = A.K, Ppad = A.Ppad;
    const int* lab = A.labels + sh->label_off;
    const int j = chunk * 64 + lane;  // pair index: (blank_j, label_j) for alpha, (label_{j-1}, blank_j) for beta
    const int own = DIR == 0 ? j : j - 1;
    const bool own_ok = own >= 0 && own < L;
    const int own_lab = own_ok ? lab[own] : A.blank;
    const float skipf = (j >= 1 && j <= L - 1 && lab[j] != lab[j - 1]) ? 1.f : 0.f;

    const int prod_chunk = DIR == 0 ? chunk - 1 : chunk + 1;
    constexpr bool has_prod = HAS_PROD;  // = prod_chunk >= 0 && prod_chunk < nchunks
    constexpr bool has_cons = HAS_CONS;  // = DIR == 0 ? (chunk + 1 < nchunks) : (chunk > 0), decided by the caller: a
                                         // wave-uniform branch per step is still a taken branch per step
    // hand-off arrays of {label hat, exponent} pairs: entry r holds the edge lane's state after step r - 1
    float* hand_w = hand_all + (long)(has_cons ? ctc_hand_slot(DIR, chunk, nchunks) : 0) * 2 * A.hand_stride;
    const float* hand_r = hand_all + (long)(has_prod ? ctc_hand_slot(DIR, prod_chunk, nchunks) : 0) * 2 * A.hand_stride;
    constexpr int edge_lane = DIR == 0 ? 63 : 0;
    constexpr int kNoExp = -(1 << 28);  // exponent of "nothing": loses every max()

    float Bst = (DIR == 0 ? (j == 0) : (j == L)) ? 1.0f : 0.f;
    float Lst = 0.f;
    int e = Bst > 0.f ? 0 : kNoExp;  // true states = (Bst, Lst) * 2^e; a pair no mass has reached yet yields to any neighbour

    const int t0 = DIR == 0 ? 0 : (T > 0 ? T - 1 : 0);
    const int dK = DIR == 0 ? K : -K;
    const float* em = LDS_EM ? em_lds : A.ly2 + (long)b * A.ly_sb;
    int e_l = t0 * K + own_lab;
    int e_b = t0 * K + A.blank;
    // The stash, per (b, t): alpha triples {blank_j, label_j, e_j}[Ppad], then beta triples {blank_j, label_{j-1}, e_j}[Ppad]
    // -- ONE 12-byte buffer store per step and wave: per-lane byte offset in a VGPR that never changes, the step's offset
    // in an SGPR (no 64-bit vector address arithmetic on the chain).
    __amdgpu_buffer_rsrc_t sres = __builtin_amdgcn_make_buffer_rsrc((void*)(A.stash + (long)b * A.T_max * 6 * Ppad), 0,
                                                                     0x7fffffff, 0x00020000);
    const int s_lane = (DIR * Ppad + j) * 12;
    int s_step = t0 * 6 * Ppad * 4;
    const int s_dstep = (DIR == 0 ? 6 : -6) * Ppad * 4;
    // The hand-off to the next chunk: the edge lane's (label hat, exponent) as ONE 8-byte LDS write per step; the other
    // lanes write the same bytes into a scratch slot of their own (no divergent branch on the chain).
    float* hand_lane = (lane == edge_lane) ? hand_w : hdummy + 2 * lane;
    const int hand_dlane = (lane == edge_lane) ? 2 : 0;

    // Emissions of a batch of kU steps (probabilities; a lane without a label state gets 0).  No branch: the rows past the
    // utterance's end that the one-batch-ahead prefetch touches are slack rows of the LDS copy (never used), and the
    // global path clamps its row.
    float el[kU], eb[kU];
    const int e_lo = 0, e_hi = (T > 0 ? T - 1 : 0) * K;
    auto load_emissions = [&](float* pel, float* peb) {
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            if (LDS_EM) {
                peb[k] = em[e_b];
                pel[k] = own_ok ? em[e_l] : 0.f;
            } else {
                const int rb = min(max(e_b - A.blank, e_lo), e_hi);  // row offset, clamped into the utterance
                peb[k] = sa_exp2(em[rb + A.blank]);
                pel[k] = own_ok ? sa_exp2(em[rb + own_lab]) : 0.f;
            }
            e_b += dK;
            e_l += dK;
        }
    };
    load_emissions(el, eb);

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef int i32x3 __attribute__((ext_vector_type(3)));
    const f32x2 skip2 = {1.0f, skipf};
    auto do_step = [&](float yl, float yb, float h, int he, bool renorm) {
        if (renorm) {  // every kPRenorm steps: the pair's larger state back to 2^kPTarget (exact: a power of two)
            const float mx = fmaxf(Bst, Lst);
            const int f = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) - kPTarget : 0;
            Bst = __builtin_amdgcn_ldexpf(Bst, -f);
            Lst = __builtin_amdgcn_ldexpf(Lst, -f);
            e += f;
        }
        const float n = DIR == 0 ? sa_wave_shr1(Lst, h) : sa_wave_shl1(Lst, h);
        const int en = __builtin_bit_cast(int, DIR == 0 ? sa_wave_shr1(__builtin_bit_cast(float, e), __builtin_bit_cast(float, he))
                                                       : sa_wave_shl1(__builtin_bit_cast(float, e), __builtin_bit_cast(float, he)));
        const int ec = max(e, en);
        const float Bs = __builtin_amdgcn_ldexpf(Bst, e - ec), Ls = __builtin_amdgcn_ldexpf(Lst, e - ec);
        const float ns = __builtin_amdgcn_ldexpf(n, en - ec);
        const f32x2 nn = {ns, ns}, base = {Bs, Ls + Bs}, yy = {yb, yl};
        const f32x2 sum = __builtin_elementwise_fma(nn, skip2, base);  // (blank_j + n, label_j + blank_j + [skip] n)
        const f32x2 nxt = sum * yy;
        Bst = nxt.x;
        Lst = nxt.y;
        // a pair that is still empty keeps "nothing": an exponent adopted ahead of the mass would run in front of it, never
        // re-normalised (only a pair that holds something re-normalises), while the front behind it decays away from it
        e = fmaxf(sum.x, sum.y) > 0.f ? ec : kNoExp;
        if (WITH_BETA) {
            const i32x3 tr = {__builtin_bit_cast(int, Bst), __builtin_bit_cast(int, Lst), e};
            __builtin_amdgcn_raw_buffer_store_b96(tr, sres, s_lane, s_step, 0);
            // A 12-byte store reads its data registers over several cycles AFTER it has issued, and hipcc's hazard recogniser
            // only guards stores without an SGPR offset: the next step's first VALU result (the alignment shift, allocated
            // to the register that held e) landed in the third dword of the LAST lanes' triples -- found by the row check
            // of ctc_grad_kernel on runs of repeated labels.  Two idle issue cycles close the window.
            asm volatile("s_nop 1" ::: "memory");
            s_step += s_dstep;
        }
        if (has_cons) {  // wave-uniform
            hand_lane += hand_dlane;
            *reinterpret_cast<float2*>(hand_lane) = make_float2(Lst, __builtin_bit_cast(float, e));
        }
    };

    int avail = 0;
    const bool dbg_me = A.dbg != nullptr && b == 0 && DIR == 0 && chunk == 0 && lane == 0;
    unsigned long long c0 = 0, w0 = 0;
    if (dbg_me) { c0 = clock64(); w0 = wall_clock64(); }
    // the neighbour chunk's edge values (hat, exponent) for the batch starting at r0
    float hv[kU];
    int hx[kU];
    auto take_edges = [&](int r0) {
        if (has_prod) {
            const int need = min(r0 + kU, T);
            if (avail < need) {
                int spins = 0;
                while ((avail = __hip_atomic_load(&sh->prog[DIR][prod_chunk], __ATOMIC_ACQUIRE,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 24)) {  // bounded: never hang the GPU on a protocol bug
                        sh->timeout = 1;
                        break;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < kU / 2; ++q) {  // {hat, exponent} x 2 per 16-byte read
                const float4 hq = *reinterpret_cast<const float4*>(hand_r + 2 * r0 + 4 * q);
                hv[2 * q] = hq.x; hx[2 * q] = __builtin_bit_cast(int, hq.y);
                hv[2 * q + 1] = hq.z; hx[2 * q + 1] = __builtin_bit_cast(int, hq.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kU; ++k) { hv[k] = 0.f; hx[k] = kNoExp; }
        }
    };
    // Phase 2, once the front has passed every pair of this chunk (all its pairs carry an exponent): the exponents are
    // FROZEN for a batch of kU steps.  At the start of the batch a lane re-normalises its pair to 2^kFTarget, reads the
    // neighbour's exponent once and keeps pd = 2^(e_neighbour - e) as a float; a step is then
    //     (blank, label) <- ((blank, label + blank) + n * (pd, [skip] pd)) * (y_blank, y_label)
    // -- one DPP, one add, one packed fma, one packed multiply.  The values arriving from the neighbour chunk each carry
    // their own exponent and are brought to the edge lane's scale when the batch is loaded (uniform work, off the chain).
    // A neighbour more than 2^kFMaxShift above the own pair would overflow the frozen product: it flags the utterance.
    float pdx = 0.f, pdy = 0.f;
    float hs[kU];
    auto refresh = [&]() {
        const float mx = fmaxf(Bst, Lst);
        const bool blown = !(mx < 3.0e38f);  // a state overflowed inside the last batch (inf, or NaN behind it): flagged below
        const int f = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) - kFTarget : 0;
        Bst = __builtin_amdgcn_ldexpf(Bst, -f);
        Lst = __builtin_amdgcn_ldexpf(Lst, -f);
        e += f;
        // lift (see wavep_refresh): the max-plus scan along this chunk's 64 pairs, seeded with the neighbour chunk's scale --
        // its hand-off values carry their own (post-lift) exponents -- as the pair in front of the edge lane.  Only when
        // some pair IS more than 2^kFLift below its predecessor (flat distributions: never; one DPP and a ballot then).
        {
            constexpr int kNone = -(1 << 29);
            // (the test looks at the gaps INSIDE the chunk only: the hand-off values of the neighbour chunk are still on their
            // way out of LDS here, and waiting for them would put an LDS round trip on every batch of the chain -- measured:
            // 89 -> 105 ns per step; an edge pair that alone sits too deep overflows its shift and is flagged a batch later)
            const int en0 = __builtin_bit_cast(int, DIR == 0 ? sa_wave_shr1(__builtin_bit_cast(float, e), __builtin_bit_cast(float, kNoExp))
                                                            : sa_wave_shl1(__builtin_bit_cast(float, e), __builtin_bit_cast(float, kNoExp)));
            if (__builtin_amdgcn_ballot_w64(e != kNoExp && en0 != kNoExp && en0 - e > kFLift) != 0) {
                int hmax = kNone;
#pragma unroll
                for (int k = 0; k < kU; ++k) hmax = max(hmax, hv[k] > 0.f ? hx[k] + __builtin_amdgcn_frexp_expf(hv[k]) - kFTarget : kNone);
                if (!has_prod) hmax = kNone;
                int m;
                if (DIR == 0) {
                    const int v = e + lane * kFLift;
                    m = max(max(v, wave_prefix_max_excl(v, kNone, lane)), hmax - kFLift) - lane * kFLift;   // seed = pair -1
                } else {
                    const int v = e - lane * kFLift;
                    m = max(max(v, wave_suffix_max_excl(v, kNone, lane)), hmax - 64 * kFLift) + lane * kFLift;  // seed = pair 64
                }
                const int up = e != kNoExp ? min(m - e, 300) : 0;
                Bst = __builtin_amdgcn_ldexpf(Bst, -up);
                Lst = __builtin_amdgcn_ldexpf(Lst, -up);
                e += up;
            }
        }
        const int e_edge = __builtin_amdgcn_readlane(e, DIR == 0 ? 0 : 63);
#pragma unroll
        for (int k = 0; k < kU; ++k)
            hs[k] = (has_prod && e_edge != kNoExp) ? __builtin_amdgcn_ldexpf(hv[k], max(hx[k], kNoExp) - e_edge) : 0.f;
        const int en = __builtin_bit_cast(int, DIR == 0 ? sa_wave_shr1(__builtin_bit_cast(float, e), __builtin_bit_cast(float, e))
                                                       : sa_wave_shl1(__builtin_bit_cast(float, e), __builtin_bit_cast(float, e)));
        const bool valid = e != kNoExp && en != kNoExp;
        const int d = valid ? en - e : 0;  // (the neighbour's own lift can leave more than kFLift between the two)
        if (d > kFMaxShift || blown) atomicOr(&sh->suspect, 1);
        const float pd = valid ? __builtin_amdgcn_ldexpf(1.0f, min(d, kFMaxShift)) : 0.f;
        pdx = pd;
        pdy = pd * skipf;
    };
    auto fast_step = [&](float yl, float yb, float h) {
        const float n = DIR == 0 ? sa_wave_shr1(Lst, h) : sa_wave_shl1(Lst, h);
        const f32x2 nn = {n, n}, pp = {pdx, pdy}, base = {Bst, Lst + Bst}, yy = {yb, yl};
        const f32x2 nxt = __builtin_elementwise_fma(nn, pp, base) * yy;
        Bst = nxt.x;
        Lst = nxt.y;
        if (WITH_BETA) {
            const i32x3 tr = {__builtin_bit_cast(int, Bst), __builtin_bit_cast(int, Lst), e};
            __builtin_amdgcn_raw_buffer_store_b96(tr, sres, s_lane, s_step, 0);
            // A 12-byte store reads its data registers over several cycles AFTER it has issued, and hipcc's hazard recogniser
            // only guards stores without an SGPR offset: the next step's first VALU result (the alignment shift, allocated
            // to the register that held e) landed in the third dword of the LAST lanes' triples -- found by the row check
            // of ctc_grad_kernel on runs of repeated labels.  Two idle issue cycles close the window.
            asm volatile("s_nop 1" ::: "memory");
            s_step += s_dstep;
        }
        if (has_cons) {
            hand_lane += hand_dlane;
            *reinterpret_cast<float2*>(hand_lane) = make_float2(Lst, __builtin_bit_cast(float, e));
        }
    };
    const int nfull = T / kU;
    // batches of phase 1: until the front -- one pair per step from the lattice's start, one more step per repeated label
    // (label_j reaches label_{j+1} of the same class only through the blank between them) -- has crossed this chunk
    const int front = (DIR == 0 ? 64 * (chunk + 1) : max(L - 64 * chunk, 0)) + sh->nrep;
    const int nslow = A.no_fast ? nfull : min(nfull, (front + 2 * kU) / kU);
    for (int bi = 0; bi < nslow; ++bi) {  // whole batches: no condition inside (a taken branch costs ~40 cycles of the chain)
        float nel[kU], neb[kU];
        load_emissions(nel, neb);  // prefetch the next batch's emissions
        take_edges(bi * kU);
#pragma unroll
        for (int k = 0; k < kU; ++k) do_step(el[k], eb[k], hv[k], hx[k], k % kPRenorm == 0);
        if (has_cons && lane == 0)
            __hip_atomic_store(&sh->prog[DIR][chunk], (bi + 1) * kU, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 0; k < kU; ++k) { el[k] = nel[k]; eb[k] = neb[k]; }
    }
    {
        float nel[kU], neb[kU];
        int bi = nslow;
        for (; bi + 1 < nfull; bi += 2) {  // two batches per trip: the emission registers alternate instead of being copied
            load_emissions(nel, neb);
            take_edges(bi * kU);
            refresh();
#pragma unroll
            for (int k = 0; k < kU; ++k) fast_step(el[k], eb[k], hs[k]);
            if (has_cons && lane == 0)
                __hip_atomic_store(&sh->prog[DIR][chunk], (bi + 1) * kU, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            load_emissions(el, eb);
            take_edges((bi + 1) * kU);
            refresh();
#pragma unroll
            for (int k = 0; k < kU; ++k) fast_step(nel[k], neb[k], hs[k]);
            if (has_cons && lane == 0)
                __hip_atomic_store(&sh->prog[DIR][chunk], (bi + 2) * kU, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (bi < nfull) {
            load_emissions(nel, neb);
            take_edges(bi * kU);
            refresh();
#pragma unroll
            for (int k = 0; k < kU; ++k) fast_step(el[k], eb[k], hs[k]);
            if (has_cons && lane == 0)
                __hip_atomic_store(&sh->prog[DIR][chunk], (bi + 1) * kU, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int k = 0; k < kU; ++k) { el[k] = nel[k]; eb[k] = neb[k]; }
        }
    }
    if (nfull * kU < T) {  // the last, partial batch
        take_edges(nfull * kU);
#pragma unroll
        for (int k = 0; k < kU; ++k)
            if (nfull * kU + k < T) do_step(el[k], eb[k], hv[k], hx[k], k % kPRenorm == 0);
        if (has_cons && lane == 0)
            __hip_atomic_store(&sh->prog[DIR][chunk], T, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    if (dbg_me) { A.dbg[0] = clock64() - c0; A.dbg[1] = wall_clock64() - w0; A.dbg[2] = (unsigned long long)T; }
    if (DIR == 0) {  // as (log2 hat, offset), the form the log-domain chain leaves
        if (j == L) { sh->fin[0] = Bst > 0.f ? sa_log2(Bst) : SA_NEG; sh->fin[1] = (float)e; }
        if (j == L - 1) { sh->fin[2] = Lst > 0.f ? sa_log2(Lst) : SA_NEG; sh->fin[3] = (float)e; }
    }
}

template <bool PROB>
__device__ __forceinline__ void ctc_grad_row(const AbArgs& A, float* __restrict__ grads, long st, long sb, int b, int t,
                                             int lane, float* srt);

// wave 0 of a workgroup of the GATED launch, on its way out (its utterance's cost is in memory): see AbArgs::loss_out
__device__ __forceinline__ void ctc_fold_cost(const AbArgs& A, int lane) {
    int last = 0;
    if (lane == 0) {
        __threadfence();  // this workgroup's cost (if it wrote one) before its arrival
        last = atomicAdd(A.done_ctr, 1u) == gridDim.x - 1u;
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    __threadfence();
    float v = 0.f;  // ctc_cost_sum_kernel's order: lane-strided partial sums, then the wave's DPP tree
    for (int b = lane; b < (int)gridDim.x; b += 64) v += __hip_atomic_load(A.costs + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = sa_wave_sum_dpp(v);
    if (lane == 0) {
        *A.loss_out = v * A.loss_scale;
        *A.done_ctr = 0u;
    }
}

template <bool WITH_BETA, bool LDS_EM, bool PROB>
__global__ __launch_bounds__(1024) void ctc_alphabeta_kernel(AbArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AbShared* sh = reinterpret_cast<AbShared*>(smem_raw);
    // hand-off arrays [2][nchunks][hand_stride]: floats in the log domain, {hat, exponent} pairs in the probability domain
    // (the region holds 2 * hand_stride floats per (direction, chunk) either way)
    // The layout differs by domain (the host sizes it the same way, ab_fixed_lds): what does not fit costs the emissions'
    // place in LDS.
    //   PROB: hand {hat, exponent}[2][nchunks][hand_stride] | scratch slots [waves][64 x 2] | class counts | emissions
    //   log : hand [2][nchunks][hand_stride] | offsets [2][nchunks][nbatch] | class counts | gradient-row scratch
    //         [waves][align4(Ppad + 1)] (the hand-over pass, ctc_grad_row) | emissions
    float* hand_all = reinterpret_cast<float*>(smem_raw + kAbSharedBytes);
    const int nslots = 2 * (A.nchunks - 1);                                              // see ctc_hand_slot
    float* hoff_all = hand_all + (long)nslots * (PROB ? 2 : 1) * A.hand_stride;         // log only
    float* hdummy_all = hoff_all;                                                          // PROB only
    int* cls_cnt = reinterpret_cast<int*>(hoff_all + (PROB ? (long)2 * A.nchunks * 128
                                                            : (((long)nslots * A.nbatch + 3) & ~3L)));  // 16-byte steps
    float* srt_all = reinterpret_cast<float*>(cls_cnt + ((A.K + 1 + 3) & ~3));           // log only
    float* em_lds = srt_all + (PROB ? 0L : (long)2 * A.nchunks * ((A.Ppad + 1 + 3) & ~3)) +
                    kU * A.K;  // [kU slack rows][T][K][kU slack rows] (LDS_EM only; see ctc_chain_p)

    const int b = blockIdx.x;
    if (A.gate && A.flags[b] == 0) {  // the log-domain pass behind a probability-domain one: flagged utterances only
        if (A.loss_out && threadIdx.x < 64) ctc_fold_cost(A, threadIdx.x);
        return;
    }
    if (PROB && A.done_ctr && b == 0 && threadIdx.x == 0) *A.done_ctr = 0u;  // (the gated launch behind this one counts arrivals)
    if (A.prof && threadIdx.x == 0) atomicMin(A.prof, (unsigned long long)wall_clock64());
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = wave / A.nchunks;
    const int chunk = wave - dir * A.nchunks;
    const int L = A.label_lens[b];
    const int T = A.in_lens[b];

    if (wave == 0) {  // flat-label offset of this utterance
        int acc = 0;
        for (int i = lane; i < b; i += 64) acc += A.label_lens[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        {
            const int* labp = A.labels + acc;
            float nr = 0.f;
            for (int j = 1 + lane; j < L; j += 64) nr += labp[j] == labp[j - 1] ? 1.f : 0.f;
            nr = sa_wave_sum_dpp(nr);
            if (lane == 0) sh->nrep = (int)nr;
        }
        if (lane == 0) {
            sh->label_off = acc;
            sh->timeout = 0;
            sh->suspect = 0;
            sh->fin[0] = SA_NEG; sh->fin[1] = 0.f; sh->fin[2] = SA_NEG; sh->fin[3] = 0.f;
        }
        if (WITH_BETA) {
            // Counting sort of the label states by class (this wave alone, before the chains start): slot[j] = position of
            // label j among the labels ordered by class (ties in label order: the LDS atomic serves its lanes in order),
            // cend[k] = one past the last slot of class k.
            int* ls = A.lsort + (long)b * (1 + A.Ppad + A.K);
            const int* lab = A.labels + acc;
            for (int k = lane; k <= A.K; k += 64) cls_cnt[k] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            int rank[kMaxChunks];
#pragma unroll
            for (int c = 0; c < kMaxChunks; ++c) {
                const int j = c * 64 + lane;
                rank[c] = (c < A.nchunks && j < L) ? atomicAdd(&cls_cnt[lab[j]], 1) : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            int carry = 0;  // exclusive scan of the class counts, 64 classes at a time
            for (int k0 = 0; k0 < A.K; k0 += 64) {
                const int k = k0 + lane;
                const int cnt = k < A.K ? cls_cnt[k] : 0;
                const int incl = (int)wave_scan_dpp((float)cnt) + carry;  // counts <= 511: exact in fp32
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (k < A.K) { cls_cnt[k] = incl - cnt; ls[1 + A.Ppad + k] = incl; }
                carry = __builtin_amdgcn_readlane(incl, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int c = 0; c < kMaxChunks; ++c) {
                const int j = c * 64 + lane;
                if (c < A.nchunks && j < L) ls[1 + j] = cls_cnt[lab[j]] + rank[c];
            }
            if (lane == 0) ls[0] = acc;
        }
    }
    if (lane == 0) {
        sh->prog[dir][chunk] = 0;
        const bool cons = dir == 0 ? (chunk + 1 < A.nchunks) : (chunk > 0);
        if (cons) {  // entry 0 of this chunk's hand-off array: "nothing yet"
            const long slot = ctc_hand_slot(dir, chunk, A.nchunks);
            if (PROB) {
                hand_all[slot * 2 * A.hand_stride] = 0.f;
                hand_all[slot * 2 * A.hand_stride + 1] = __builtin_bit_cast(float, -(1 << 28));
            } else {
                hand_all[slot * A.hand_stride] = SA_NEG;
            }
        }
    }
    if (LDS_EM) {
        // Stage this utterance's emission rows once: contiguous T*K floats, 16-byte aligned at both ends, as
        // float4 with 8 independent loads in flight per lane (a dependent load->store loop would cost ~0.5 us
        // per trip and dominate the kernel).
        const float4* src = reinterpret_cast<const float4*>(A.ly2 + (long)b * A.ly_sb);
        float4* dst = reinterpret_cast<float4*>(em_lds);
        const int n4 = (T * A.K + 3) >> 2;
        const int nt = blockDim.x;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * nt) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * nt;
                v[u] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * nt;
                if (PROB) {  // log2 softmax -> probabilities, once (exp2 of the SA_NEG floor is 0)
                    v[u].x = sa_exp2(v[u].x); v[u].y = sa_exp2(v[u].y); v[u].z = sa_exp2(v[u].z); v[u].w = sa_exp2(v[u].w);
                }
                if (i < n4) dst[i] = v[u];
            }
        }
    }
    __syncthreads();

    if (PROB) {
        float* hd = hdummy_all + wave * 128;
        const bool first = dir == 0 ? chunk == 0 : chunk + 1 == A.nchunks;  // of the pipeline: takes no edge values
        const bool last = dir == 0 ? chunk + 1 == A.nchunks : chunk == 0;   // hands none on
#define SA_CHAIN_P(D_, P_, C_) ctc_chain_p<D_, WITH_BETA, LDS_EM, P_, C_>(A, sh, hand_all, hd, em_lds, b, chunk, lane, L, T)
        if (dir == 0) {
            if (first) { if (last) SA_CHAIN_P(0, false, false); else SA_CHAIN_P(0, false, true); }
            else { if (last) SA_CHAIN_P(0, true, false); else SA_CHAIN_P(0, true, true); }
        } else {
            if (first) { if (last) SA_CHAIN_P(1, false, false); else SA_CHAIN_P(1, false, true); }
            else { if (last) SA_CHAIN_P(1, true, false); else SA_CHAIN_P(1, true, true); }
        }
#undef SA_CHAIN_P
    } else {
        if (dir == 0)
            ctc_chain<0, WITH_BETA, LDS_EM>(A, sh, hand_all, hoff_all, em_lds, b, chunk, lane, L, T);
        else
            ctc_chain<1, WITH_BETA, LDS_EM>(A, sh, hand_all, hoff_all, em_lds, b, chunk, lane, L, T);
    }

    __syncthreads();
    if (threadIdx.x == 0) {
        // log2 p = lse2(fin0 + off0, fin2 + off2), kept as (hat, integer offset)
        const float o0 = sh->fin[1];
        const float f0 = sh->fin[0];
        const float f1 = sh->fin[2] + (sh->fin[3] - o0);
        const float lp = lse2_1p(f0, f1);
        const bool dead = lp < SA_NEG_TEST;
        const float lpi = PROB && !dead ? __builtin_rintf(lp) : 0.f;  // PROB: the integer part joins the offset (exact)
        A.logp2_out[2 * b] = dead ? SA_NEG : lp - lpi;
        A.logp2_out[2 * b + 1] = dead ? 0.f : o0 + lpi;
        float c = dead ? __builtin_inff() : (float)(-((double)lp + (double)o0) * 0.6931471805599453);
        if (sh->timeout) c = __builtin_bit_cast(float, 0x7fc00000);  // NaN marks a hand-off timeout (never expected)
        A.costs[b] = c;
        if (PROB) A.flags[b] = (dead || sh->suspect || !(lp < 3.0e38f)) ? 1 : 0;  // (inf / NaN: an overflow inside the last batch)  // every utterance writes its flag (no memset between calls); "infeasible" is
                                              // the log-domain kernels' call
    }
    if (A.prof && threadIdx.x == 0) atomicMax(A.prof + 1, (unsigned long long)wall_clock64());  // (the chains are done: the
                                                                                                // hand-over rows below are not chain time)
    if (!PROB && WITH_BETA && A.gate) {  // the hand-over pass: this block also writes the utterance's gradient rows
        __threadfence();                 // the stash, costs and log2 p written above are read back through memory
        __syncthreads();
        float* srt = srt_all + (long)wave * ((A.Ppad + 1 + 3) & ~3);
        const int nw = blockDim.x >> 6;
        for (int t = wave; t < A.T_max; t += nw) ctc_grad_row<false>(A, A.grads, A.g_st, A.g_sb, b, t, lane, srt);
    }
    if (!PROB && A.gate && A.loss_out && wave == 0) ctc_fold_cost(A, lane);  // (thread 0 wrote A.costs[b] above: same wave)
}

// ---------------------------------------------------------------------------------------------------------- K_C
// One wave, one lattice row t of utterance b; srt: Ppad + 1 floats of LDS owned by the wave.
template <bool PROB>  // PROB: the stash holds hats of the probability-domain chain (log2 is taken here, off the chain)
__device__ __forceinline__ void ctc_grad_row(const AbArgs& A, float* __restrict__ grads, long st, long sb, int b, int t,
                                             int lane, float* srt) {
    const int K = A.K, Ppad = A.Ppad, nchunks = A.nchunks;
    const int L = A.label_lens[b];
    const int T = A.in_lens[b];
    float* g = grads + (long)b * sb + (long)t * st;
    const float lp = A.logp2_out[2 * b];
    const float lpo = A.logp2_out[2 * b + 1];
    if (t >= T || lp < SA_NEG_TEST) {  // padding row, or infeasible alignment: zero gradient
        for (int k = lane; k < K; k += 64) g[k] = 0.f;
        return;
    }
    const int* ls = A.lsort + (long)b * (1 + Ppad + K);  // see AbArgs
    const int* lab = A.labels + ls[0];

    const float* row = A.ly2 + (long)b * A.ly_sb + (long)t * K;
    const float* sp = A.stash + ((long)b * A.T_max + t) * 6 * Ppad;
    const float* goA = A.goffs + (long)(b * 2 + 0) * nchunks * A.nbatch + t / kU;
    const float* goB = A.goffs + (long)(b * 2 + 1) * nchunks * A.nbatch + (T - 1 - t) / kU;
    const float lyblank = row[A.blank];
    float accB = 0.f;
    for (int c = 0; c < nchunks; ++c) {
        const int j = c * 64 + lane;
        const float oA = goA[c * A.nbatch];
        const float oB = goB[c * A.nbatch];
        // beta's label state i lives in the wave of pair i+1: lane 63 crosses into the next chunk
        const float oBn = (c + 1 < nchunks) ? goB[(c + 1) * A.nbatch] : oB;
        const float io_b = oA + oB - lpo;                       // integers: exact
        const float io_l = oA + (lane == 63 ? oBn : oB) - lpo;
        if (PROB) {
            // log2 of a state = its integer exponents (the hat's own and the pair's: exact) + log2 of the hat's mantissa in
            // [-1, 0): the float sum below only holds small numbers wherever the occupancy is not itself negligible
            // stash of the probability-domain chain: alpha triples {blank_j, label_j, e_j}, then beta triples
            // {blank_j, label_{j-1}, e_j} (ctc_chain_p)
            const float* ta = sp + 3 * j;
            const float* tb = sp + 3 * (Ppad + j);
            if (j <= L) {
                const float a = ta[0], bt = tb[0];
                const float fr = sa_log2(__builtin_amdgcn_frexp_mantf(a)) + sa_log2(__builtin_amdgcn_frexp_mantf(bt));
                const int in = __builtin_amdgcn_frexp_expf(a) + __builtin_amdgcn_frexp_expf(bt) +
                               __builtin_bit_cast(int, ta[2]) + __builtin_bit_cast(int, tb[2]);
                accB += sa_exp2((fr - lyblank - lp) + ((float)in - lpo));  // log2 0 = -inf: an empty state adds exp2(-inf) = 0
            }
            if (j < L) {
                const int k = lab[j];
                const float a = ta[1], bt = tb[4];  // beta's label state j belongs to beta pair j + 1
                const float fr = sa_log2(__builtin_amdgcn_frexp_mantf(a)) + sa_log2(__builtin_amdgcn_frexp_mantf(bt));
                const int in = __builtin_amdgcn_frexp_expf(a) + __builtin_amdgcn_frexp_expf(bt) +
                               __builtin_bit_cast(int, ta[2]) + __builtin_bit_cast(int, tb[5]);
                srt[ls[1 + j]] = sa_exp2((fr - row[k] - lp) + ((float)in - lpo));
            }
        } else {
            if (j <= L) accB += sa_exp2((sp[j] + sp[2 * Ppad + j] - lyblank - lp) + io_b);
            if (j < L) {
                const int k = lab[j];
                const float gm = sa_exp2((sp[Ppad + j] + sp[3 * Ppad + j] - row[k] - lp) + io_l);
                srt[ls[1 + j]] = gm;
            }
        }
    }
    accB = sa_wave_sum_dpp(accB);
    // Occupancy of class k = the sum over its run of sorted slots = a difference of two prefix sums.  (LDS float atomics, the
    // obvious way, cost ~4 cycles per lane on the CU's one LDS pipe: 100 label states x 32000 rows bounded this kernel.)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float carry = 0.f;
    for (int c = 0; c < nchunks; ++c) {
        const int i = c * 64 + lane;
        const float v = i < L ? srt[i] : 0.f;
        const float incl = wave_scan_dpp(v) + carry;
        srt[i] = incl;   // in place: the sum of slots 0 .. i
        carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float total = accB + carry;
    for (int k = lane; k < K; k += 64) {
        const float y = sa_exp2(row[k]);
        const int hi = ls[1 + Ppad + k], lo = k > 0 ? ls[Ppad + k] : 0;
        const float run = hi > lo ? srt[hi - 1] - (lo > 0 ? srt[lo - 1] : 0.f) : 0.f;
        const float o = (k == A.blank) ? accB : run;
        g[k] = (y - o) * A.gscale;
    }
    if (PROB) {
        // certification (2) of the probability-domain pass: the occupancies of a row sum to one -- the paths through the
        // lattice at time t are all the paths -- unless mass was lost on the way (NaN compares false: flagged too)
        if (!(fabsf(total - 1.0f) < 1e-4f) && lane == 0) atomicOr(&A.flags[b], 2);
    }
}

// grid (ceil(T_max / 4), B), 256 threads: one wave per lattice row.
template <bool PROB>
__global__ __launch_bounds__(256) void ctc_grad_kernel(AbArgs A, float* __restrict__ grads, long st, long sb) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* srt_all = reinterpret_cast<float*>(smem_raw);  // [4 waves][Ppad + 1]
    const int b = blockIdx.y;
    if (A.gate && A.flags[b] == 0) return;  // see ctc_alphabeta_kernel
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wave;
    if (t >= A.T_max) return;
    ctc_grad_row<PROB>(A, grads, st, sb, b, t, lane, srt_all + wave * (A.Ppad + 1));
}

// ---------------------------------------------------------------------------------------------------------- K_W
// Throughput regime (several utterances per CU, e.g. SURVEY.md 8d "M-CTC at saturating B"): ONE WAVE per utterance.
// K_B above spends a whole workgroup, ~120 KB of LDS and 4 HBM stash planes per utterance to cut the latency of one
// lattice; with thousands of lattices the chip is instead bound by stash traffic and by how many lattices a CU can
// hold.  Here lane i owns R CONTIGUOUS state pairs (R = 1, 2, 4, 8 for up to 63 / 127 / 255 / 511 labels): the R
// pair updates of a step are independent (ILP inside the wave, one DPP for the lane boundary), there is no
// cross-wave hand-off at all, and alpha then beta run back to back in the same wave.
//   alpha pass  keeps only a CHECKPOINT of its states every KU steps (1/KU of a full alpha plane stash);
//   beta pass   walks the batches of KU steps backwards: reloads the batch's checkpoint (prefetched one batch
//               ahead), recomputes the batch's alpha states into registers -- bit-identical, the arithmetic and the
//               renormalisation points are the same -- then runs its own KU steps, forming the occupancies on the
//               fly (blank: DPP wave sum, labels: LDS float atomics into a K-entry row owned by the wave) and
//               writing the gradient row itself.  No beta stash, no K_C.
// A score-only alpha pass of 4096 utterances takes 0.48 ms, so recomputing alpha is cheaper than the 8.4 GB of HBM
// traffic a full alpha stash costs.  Emission rows are staged KU rows at a time into a per-wave LDS ring with
// coalesced loads issued a batch ahead.  ~4.5 KB of LDS per wave at K = 29: occupancy is VGPR-bound (4 waves / SIMD).
struct WaveArgs {
    const float* acts;  // ctc_wave_p_kernel: the activations themselves (strides st / sb below), normalised while staged
    const float* ly2;
    const int* labels;
    const int* label_lens;
    const int* in_lens;
    int K, T_max, blank, B, nren, nq, wave_lds_floats;
    long ly_sb;
    float* stash;  // [B][nq][2][64 * R]: alpha (blank, label) states at the start of every batch of KU steps
    float* costs;
    float* grads;
    long st, sb;
    float gscale;  // every gradient element is multiplied by this
    const int* only;  // non-null: redo only the utterances whose flag is set (the pass behind ctc_wave_p_kernel)
};

template <int R>
struct WaveCfg {
    static constexpr int KU = R <= 4 ? 8 : 4;  // steps per batch (bounds the recomputed-alpha registers: 2 * KU * R)
    static constexpr int NU = 8;                // staging registers: KU * K <= 64 * NU takes the SMALLK path
};

// Emission rows [tlo, tlo + nrows) of one utterance -> registers (issue) -> the wave's LDS ring (commit).
// Every VMEM access in the T loops of this kernel is unconditional (clamped index, then a select): a branch around
// a load or store makes the outstanding-operation count unknown to the compiler, which then waits for vmcnt(0) --
// the HBM round trip of the previous step's stores -- before every later use of any loaded value.
template <int R, bool SMALLK>
struct RowStager {
    static constexpr int KU = WaveCfg<R>::KU, NU = WaveCfg<R>::NU;
    const float* ly;
    int K, lane;
    float pv[NU];
    __device__ __forceinline__ void issue(int tlo, int nrows) {
        if (!SMALLK) return;
        const int n = nrows * K;
        const float* src = ly + (long)tlo * K;
#pragma unroll
        for (int u = 0; u < NU; ++u) pv[u] = src[min(lane + 64 * u, n - 1)];
    }
    __device__ __forceinline__ void commit(int tlo, int nrows, float* dst) {
        if (SMALLK) {  // the ring buffers hold 64 * NU floats on this path: no bounds test
#pragma unroll
            for (int u = 0; u < NU; ++u) dst[lane + 64 * u] = pv[u];
        } else {
            const int n = nrows * K;
            const float* src = ly + (long)tlo * K;
            for (int i = lane; i < n; i += 64) dst[i] = src[i];
        }
    }
    __device__ __forceinline__ int buf_floats() const { return SMALLK ? 64 * NU : KU * K; }
};

// One time step of a wave's 64 * R state pairs.  DIR 0: alpha pairs (blank_j, label_j), neighbour = label_{j-1};
// DIR 1: beta pairs (label_{j-1}, blank_j), neighbour = label_j.  Same shared-pivot arithmetic as K_B's do_step.
template <int R, int DIR>
__device__ __forceinline__ void wave_step(float (&Bst)[R], float (&Lst)[R], const float (&el)[R], float eb,
                                          const bool (&skip)[R]) {
    const float edge = DIR == 0 ? sa_wave_shr1(Lst[R - 1], SA_NEG) : sa_wave_shl1(Lst[0], SA_NEG);
    float nB[R], nL[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float n = DIR == 0 ? (r == 0 ? edge : Lst[r - 1]) : (r == R - 1 ? edge : Lst[r + 1]);
        const float mLB = fmaxf(Lst[r], Bst[r]);
        const float mBn = fmaxf(Bst[r], n);
        const float m = fmaxf(mLB, n);
        const float xL = sa_exp2(Lst[r] - m);
        const float xB = sa_exp2(Bst[r] - m);
        const float xn = sa_exp2(n - m);
        const float sB = xB + xn;
        const float sL = xL + xB + (skip[r] ? xn : 0.f);
        // a sum that flushed to 0 gives log2 = -inf and the max picks the larger operand (K_B's fallback, one
        // v_max instead of compare + select); otherwise lse >= max holds up to one rounding
        nB[r] = eb + fmaxf(m + sa_log2(sB), mBn);
        nL[r] = el[r] + fmaxf(m + sa_log2(sL), mLB);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { Bst[r] = nB[r]; Lst[r] = nL[r]; }
}

template <int R>
__device__ __forceinline__ float wave_renorm(float (&Bst)[R], float (&Lst)[R]) {  // exact: integer shift
    float m = SA_NEG;
#pragma unroll
    for (int r = 0; r < R; ++r) m = fmaxf(m, fmaxf(Bst[r], Lst[r]));
    m = sa_wave_max_dpp(m);
    const float d = m > SA_NEG_TEST ? __builtin_rintf(m) : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) { Bst[r] -= d; Lst[r] -= d; }
    return d;
}

template <int R, bool WITH_GRAD, bool SMALLK>
__device__ __forceinline__ void ctc_wave_alpha(const WaveArgs& A, int b, int lane, int L, int T, const int* lab,
                                               float* ring, float* aoff, float* out_lp, float* out_off) {
    constexpr int KU = WaveCfg<R>::KU, P = 64 * R;
    const int K = A.K;
    int own_lab[R];
    bool own_ok[R], skip[R];
    float Bst[R], Lst[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = lane * R + r;
        own_ok[r] = j < L;
        own_lab[r] = own_ok[r] ? lab[j] : A.blank;
        skip[r] = (j >= 1 && j <= L - 1) ? (lab[j] != lab[j - 1]) : false;
        Bst[r] = j == 0 ? 0.0f : SA_NEG;
        Lst[r] = SA_NEG;
    }
    float off = 0.f;
    RowStager<R, SMALLK> stage;
    stage.ly = A.ly2 + (long)b * A.ly_sb; stage.K = K; stage.lane = lane;
    if (T > 0) {
        stage.issue(0, min(KU, T));
        stage.commit(0, min(KU, T), ring);
    }
    __builtin_amdgcn_s_waitcnt(0);  // everything loaded so far (labels, first rows) is waited for once, here

    int q = 0;
    for (int r0 = 0; r0 < T; r0 += KU, ++q) {
        const float* cur = ring + (q & 1) * stage.buf_floats();
        float* nxt = ring + ((q + 1) & 1) * stage.buf_floats();
        if ((r0 & 31) == 0) {
            off += wave_renorm<R>(Bst, Lst);
            if (WITH_GRAD && lane == 0) aoff[r0 >> 5] = off;
        }
        if (WITH_GRAD) {
            float* ck = A.stash + ((long)b * A.nq + q) * 2 * P + lane * R;
#pragma unroll
            for (int r = 0; r < R; ++r) { ck[r] = Bst[r]; ck[P + r] = Lst[r]; }
        }
        const bool more = r0 + KU < T;
        if (more) stage.issue(r0 + KU, min(KU, T - r0 - KU));
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            if (r0 + k < T) {
                const float* rowp = cur + k * K;
                const float eb = rowp[A.blank];
                float el[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float v = rowp[own_lab[r]];  // own_lab is the blank where there is no label state
                    el[r] = own_ok[r] ? v : SA_NEG;
                }
                wave_step<R, 0>(Bst, Lst, el, eb, skip);
            }
        }
        if (more) stage.commit(r0 + KU, min(KU, T - r0 - KU), nxt);
    }

    float f0 = SA_NEG, f1 = SA_NEG;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = lane * R + r;
        if (j == L) f0 = Bst[r];
        if (j == L - 1) f1 = Lst[r];
    }
    f0 = sa_wave_max_dpp(f0);
    f1 = sa_wave_max_dpp(f1);
    *out_lp = lse2_1p(f0, f1);
    *out_off = off;
}

template <int R, bool SMALLK>
__device__ __forceinline__ void ctc_wave_beta(const WaveArgs& A, int b, int lane, int L, int T, const int* lab,
                                              float* ring, float* occ, const float* aoff, float lp, float lpo) {
    constexpr int KU = WaveCfg<R>::KU, P = 64 * R;
    const int K = A.K;
    int a_lab[R], b_lab[R];
    bool a_ok[R], b_ok[R], skip[R];
    float Bst[R], Lst[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = lane * R + r;
        a_ok[r] = j < L;
        a_lab[r] = a_ok[r] ? lab[j] : A.blank;
        b_ok[r] = j >= 1 && j - 1 < L;
        b_lab[r] = b_ok[r] ? lab[j - 1] : A.blank;
        skip[r] = (j >= 1 && j <= L - 1) ? (lab[j] != lab[j - 1]) : false;
        Bst[r] = j == L ? 0.0f : SA_NEG;
        Lst[r] = SA_NEG;
    }
    // Label occupancies of a row, occ[c] = sum over the label states i with lab[i] = c.  LDS float atomics cost
    // ~4 cycles per lane on the CU's one LDS pipe and were THE bottleneck of this pass (adding 27 % more atomic
    // lanes cost 28 % more kernel time).  For K <= 64 the scatter-add is a fixed sparse matrix per utterance, so it
    // is done as a sorted segment sum instead: once, label states are counting-sorted by class (pos[] is a
    // permutation of the P slots, deterministic: rank = number of earlier equal labels); per step each lane drops
    // its R occupancies at pos[] (plain conflict-free LDS writes), the wave takes an inclusive prefix sum over the
    // sorted array (DPP scan), and class lane c reads two prefix sums: S[last of c] - S[before first of c].
    int pos[R], seg_hi = P, seg_lo = P;  // occ[] doubles as the sorted array [P + 1]; slot P stays 0
    if (SMALLK) {
        int cnt = 0, rank[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rank[r] = 0;
        for (int i = 0; i < L; ++i) {
            const int li = lab[i];
            cnt += li == lane ? 1 : 0;
#pragma unroll
            for (int r = 0; r < R; ++r) rank[r] += (i < lane * R + r - 1 && li == b_lab[r]) ? 1 : 0;
        }
        const int incl = (int)wave_scan_dpp((float)cnt);  // counts <= 511: exact in fp32
        const int start = incl - cnt;
        occ[lane] = (float)start;
        if (lane == 0) occ[P] = 0.f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = lane * R + r;
            // label state j-1 -> its sorted slot; the P - L pairs without one (j = 0, j > L) fill slots [L, P)
            pos[r] = b_ok[r] ? (int)occ[b_lab[r]] + rank[r] : (j == 0 ? P - 1 : j - 1);
        }
        seg_lo = start > 0 ? start - 1 : P;
        seg_hi = cnt > 0 ? start + cnt - 1 : seg_lo;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    float off = 0.f;
    RowStager<R, SMALLK> stage;
    stage.ly = A.ly2 + (long)b * A.ly_sb; stage.K = K; stage.lane = lane;
    const float* ck_base = A.stash + (long)b * A.nq * 2 * P + lane * R;
    const int q_last = (T - 1) / KU;
    float cB[R], cL[R];  // checkpoint of the batch about to be processed
    {
        const int tlo = q_last * KU;
        stage.issue(tlo, T - tlo);
        stage.commit(tlo, T - tlo, ring);
        const float* ck = ck_base + (long)q_last * 2 * P;
#pragma unroll
        for (int r = 0; r < R; ++r) { cB[r] = ck[r]; cL[r] = ck[P + r]; }
    }
    __builtin_amdgcn_s_waitcnt(0);

    int bi = 0;
    for (int q = q_last; q >= 0; --q, ++bi) {
        const int tlo = q * KU;
        const int nrows = min(KU, T - tlo);
        const float* cur = ring + (bi & 1) * stage.buf_floats();
        float* nxt = ring + ((bi + 1) & 1) * stage.buf_floats();
        if ((bi & 3) == 0) off += wave_renorm<R>(Bst, Lst);
        float aBs[R], aLs[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { aBs[r] = cB[r]; aLs[r] = cL[r]; }
        if (q > 0) {  // the previous batch in time: rows and checkpoint, a batch ahead of their use
            stage.issue(tlo - KU, KU);
            const float* ck = ck_base + (long)(q - 1) * 2 * P;
#pragma unroll
            for (int r = 0; r < R; ++r) { cB[r] = ck[r]; cL[r] = ck[P + r]; }
        }
        const float io = aoff[tlo >> 5] + off - lpo;  // alpha offset + beta offset - log2 p offset: integers, exact

        // (1) recompute this batch's alpha states from its checkpoint
        float aB[KU][R], aL[KU][R];
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            if (k < nrows) {
                const float* rowp = cur + k * K;
                const float eb = rowp[A.blank];
                float el[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float v = rowp[a_lab[r]];
                    el[r] = a_ok[r] ? v : SA_NEG;
                }
                wave_step<R, 0>(aBs, aLs, el, eb, skip);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) { aB[k][r] = aBs[r]; aL[k][r] = aLs[r]; }
        }
        // (2) beta steps, backwards in time, with the occupancies and the gradient row of each step
#pragma unroll
        for (int k = KU - 1; k >= 0; --k) {
            if (k < nrows) {
                const int t = tlo + k;
                const float* rowp = cur + k * K;
                const float eb = rowp[A.blank];
                float el[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float v = rowp[b_lab[r]];
                    el[r] = b_ok[r] ? v : SA_NEG;
                }
                wave_step<R, 1>(Bst, Lst, el, eb, skip);
                // occupancy = exp2(alpha + beta - one emission - log2 p); alpha's label state of pair j-1
                const float aedge = sa_wave_shr1(aL[k][R - 1], SA_NEG);
                float gb = 0.f;
                float* g = A.grads + (long)b * A.sb + (long)t * A.st;
                if (SMALLK) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        // states that do not exist hold SA_NEG in alpha and in beta: their exp2 is exactly 0
                        gb += sa_exp2((aB[k][r] + Bst[r] - eb - lp) + io);
                        const float al = r == 0 ? aedge : aL[k][r - 1];
                        occ[pos[r]] = sa_exp2((al + Lst[r] - (b_ok[r] ? el[r] : 0.f) - lp) + io);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    float sv[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) sv[r] = occ[lane * R + r];
#pragma unroll
                    for (int r = 1; r < R; ++r) sv[r] += sv[r - 1];
                    const float excl = wave_scan_dpp(sv[R - 1]) - sv[R - 1];
                    gb = sa_wave_sum_dpp(gb);
#pragma unroll
                    for (int r = 0; r < R; ++r) occ[lane * R + r] = sv[r] + excl;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    // K <= 64: lanes past K repeat lane K-1's store (same address, same value)
                    const int c = min(lane, K - 1);
                    const float o = c == A.blank ? gb : occ[seg_hi] - occ[seg_lo];
                    g[c] = (sa_exp2(rowp[c]) - o) * A.gscale;
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        gb += sa_exp2((aB[k][r] + Bst[r] - eb - lp) + io);
                        const float al = r == 0 ? aedge : aL[k][r - 1];
                        if (b_ok[r]) atomicAdd(&occ[b_lab[r]], sa_exp2((al + Lst[r] - el[r] - lp) + io));
                    }
                    gb = sa_wave_sum_dpp(gb);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    for (int c = lane; c < K; c += 64) {
                        const float o = c == A.blank ? gb : occ[c];
                        g[c] = (sa_exp2(rowp[c]) - o) * A.gscale;
                        occ[c] = 0.f;
                    }
                }
            }
        }
        if (q > 0) stage.commit(tlo - KU, KU, nxt);
    }
}

// ---- K_W in the PROBABILITY domain (gradient calls, K <= 64, up to 255 labels; certified like ctc_chain_p) ----
// The same per-pair exponents and two phases as ctc_chain_p, for R pairs per lane: until the front has crossed the lattice
// (L + 1 steps) a step aligns every pair with its predecessor's exponent; afterwards the exponents are frozen for a batch
// of KU steps and a step is one add, one packed fma and one packed multiply per pair.  Hats are re-normalised to
// 2^kFTarget in BOTH phases here (they stay below 2^23, so that a product of an alpha and a beta hat cannot overflow).
// alpha keeps a checkpoint {blank, label, exponent} of every pair at the START of each batch; the beta pass replays the
// batch from it -- the same instructions in the same order, so the recomputed states are bit-identical -- and forms the
// occupancies from alpha_t(s) and beta's PRE-emission sums (gamma = alpha_t(s) * sum_beta_t(s) / p: no division by y).
// The emission rows are converted to probabilities once, when they are committed to the LDS ring.
// Certification: the beta pass checks that every row's occupancies sum to one and flags the utterance otherwise (as does
// an "infeasible" verdict or a frozen shift beyond 2^kFMaxShift); flagged utterances are redone by the log-domain kernel
// (ctc_wave_kernel with `only` = the flags), launched right behind.
template <int R, int DIR>
__device__ __forceinline__ void wavep_slow_step(float (&Bst)[R], float (&Lst)[R], int (&e)[R], const float (&yl)[R], float yb,
                                                const float (&skipf)[R], float (&sB)[R], float (&sL)[R], bool renorm) {
    constexpr int kNoExp = -(1 << 28);
    if (renorm) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float mx = fmaxf(Bst[r], Lst[r]);
            const int f = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) - kFTarget : 0;
            Bst[r] = __builtin_amdgcn_ldexpf(Bst[r], -f);
            Lst[r] = __builtin_amdgcn_ldexpf(Lst[r], -f);
            e[r] += f;
        }
    }
    const float nedge = DIR == 0 ? sa_wave_shr1(Lst[R - 1], 0.f) : sa_wave_shl1(Lst[0], 0.f);
    const int eedge = __builtin_bit_cast(int, DIR == 0 ? sa_wave_shr1(__builtin_bit_cast(float, e[R - 1]), __builtin_bit_cast(float, kNoExp))
                                                       : sa_wave_shl1(__builtin_bit_cast(float, e[0]), __builtin_bit_cast(float, kNoExp)));
    float nB[R], nL[R];
    int ne[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float n = DIR == 0 ? (r == 0 ? nedge : Lst[r - 1]) : (r == R - 1 ? nedge : Lst[r + 1]);
        const int en = DIR == 0 ? (r == 0 ? eedge : e[r - 1]) : (r == R - 1 ? eedge : e[r + 1]);
        const int ec = max(e[r], en);
        const float Bs = __builtin_amdgcn_ldexpf(Bst[r], e[r] - ec), Ls = __builtin_amdgcn_ldexpf(Lst[r], e[r] - ec);
        const float ns = __builtin_amdgcn_ldexpf(n, en - ec);
        sB[r] = Bs + ns;
        sL[r] = __builtin_fmaf(ns, skipf[r], Ls + Bs);
        nB[r] = sB[r] * yb;
        nL[r] = sL[r] * yl[r];
        ne[r] = fmaxf(sB[r], sL[r]) > 0.f ? ec : kNoExp;  // a pair that is still empty keeps "nothing" (see ctc_chain_p)
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { Bst[r] = nB[r]; Lst[r] = nL[r]; e[r] = ne[r]; }
}

// start of a frozen batch: re-normalise, read the predecessors' exponents once, keep pd = 2^(e_pred - e); returns true when
// a shift beyond 2^kFMaxShift had to be clamped (the caller flags the utterance)
template <int R, int DIR>
__device__ __forceinline__ bool wavep_refresh(float (&Bst)[R], float (&Lst)[R], int (&e)[R], const float (&skipf)[R],
                                              float (&pdx)[R], float (&pdy)[R]) {
    constexpr int kNoExp = -(1 << 28);
    bool clamped = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float mx = fmaxf(Bst[r], Lst[r]);
        clamped = clamped || !(mx < 3.0e38f);  // a state overflowed inside the last batch (inf, or NaN behind it)
        const int f = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) - kFTarget : 0;
        Bst[r] = __builtin_amdgcn_ldexpf(Bst[r], -f);
        Lst[r] = __builtin_amdgcn_ldexpf(Lst[r], -f);
        e[r] += f;
    }
    // Exponent lifting (kFLift), only when some pair IS further below its predecessor than 2^kFLift: on flat distributions
    // none is, and the batch starts with one DPP and a ballot.
    int eedge = __builtin_bit_cast(int, DIR == 0 ? sa_wave_shr1(__builtin_bit_cast(float, e[R - 1]), __builtin_bit_cast(float, kNoExp))
                                                 : sa_wave_shl1(__builtin_bit_cast(float, e[0]), __builtin_bit_cast(float, kNoExp)));
    bool deep = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int en = DIR == 0 ? (r == 0 ? eedge : e[r - 1]) : (r == R - 1 ? eedge : e[r + 1]);
        deep = deep || (e[r] != kNoExp && en != kNoExp && en - e[r] > kFLift);
    }
    if (__builtin_amdgcn_ballot_w64(deep) != 0) {
        // lift: e'[j] = max over the pairs k that mass reaches j from (k <= j for alpha, k >= j for beta) of
        // e[k] - |j - k| kFLift -- a max-plus prefix (suffix) scan along the lattice: afterwards NO pair sits more than
        // 2^kFLift below its predecessor, and nothing was lifted that a predecessor's own lift would have pushed further (a
        // one-hop lift leaves a cliff at the end of the lifted run: measured, it flagged MORE utterances than none)
        const int lane = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
        constexpr int kNone = -(1 << 29);
        int m[R];
        if (DIR == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int v = e[r] + (lane * R + r) * kFLift;
                m[r] = r == 0 ? v : max(m[r - 1], v);
            }
            const int before = wave_prefix_max_excl(m[R - 1], kNone, lane);
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] = max(m[r], before) - (lane * R + r) * kFLift;
        } else {
#pragma unroll
            for (int r = R - 1; r >= 0; --r) {
                const int v = e[r] - (lane * R + r) * kFLift;
                m[r] = r == R - 1 ? v : max(m[r + 1], v);
            }
            const int after = wave_suffix_max_excl(m[0], kNone, lane);
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] = max(m[r], after) + (lane * R + r) * kFLift;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int up = e[r] != kNoExp ? min(m[r] - e[r], 300) : 0;  // (>= 0; 2^-300 of an fp32 hat is 0 anyway)
            Bst[r] = __builtin_amdgcn_ldexpf(Bst[r], -up);
            Lst[r] = __builtin_amdgcn_ldexpf(Lst[r], -up);
            e[r] += up;
        }
        eedge = __builtin_bit_cast(int, DIR == 0 ? sa_wave_shr1(__builtin_bit_cast(float, e[R - 1]), __builtin_bit_cast(float, kNoExp))
                                                 : sa_wave_shl1(__builtin_bit_cast(float, e[0]), __builtin_bit_cast(float, kNoExp)));
    }
    // ... then the shifts, on the exponents both sides will really use during the batch
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int en = DIR == 0 ? (r == 0 ? eedge : e[r - 1]) : (r == R - 1 ? eedge : e[r + 1]);
        const bool valid = e[r] != kNoExp && en != kNoExp;
        const int d = valid ? en - e[r] : 0;
        clamped = clamped || d > kFMaxShift;
        const float pd = valid ? __builtin_amdgcn_ldexpf(1.0f, min(d, kFMaxShift)) : 0.f;
        pdx[r] = pd;
        pdy[r] = pd * skipf[r];
    }
    return clamped;
}

// Lane i receives lane i-1's (i+1's) value, the lane without a source 0 (DPP bound_ctrl: no register to preset)
__device__ __forceinline__ float sa_wave_shr1_z(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float sa_wave_shl1_z(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// One step on frozen exponents.  R even: the packed operations run ACROSS pairs -- {B_2h, B_2h+1}, {L_2h, L_2h+1} -- so a
// state lives in the register pair the previous step's packed multiply left it in (pairing a state's blank with its
// label, as rounds 2-3 did, costs ~7 register moves per step to re-form the operands): one packed add, two packed
// fmas, two packed multiplies, the lane-edge DPP move and one move per pair.  Element for element the same operations
// as before (bit-identical).
template <int R, int DIR>
__device__ __forceinline__ void wavep_fast_step(float (&Bst)[R], float (&Lst)[R], const float (&yl)[R], float yb,
                                                const float (&pdx)[R], const float (&pdy)[R], float (&sB)[R],
                                                float (&sL)[R]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float nedge = DIR == 0 ? sa_wave_shr1_z(Lst[R - 1]) : sa_wave_shl1_z(Lst[0]);
    float nB[R], nL[R];
    if constexpr ((R & 1) == 0) {
#pragma unroll
        for (int h = 0; h < R / 2; ++h) {
            const int r0 = 2 * h, r1 = 2 * h + 1;
            const float n0 = DIR == 0 ? (h == 0 ? nedge : Lst[r0 - 1]) : Lst[r1];
            const float n1 = DIR == 0 ? Lst[r0] : (r1 == R - 1 ? nedge : Lst[r1 + 1]);
            const f32x2 nn = {n0, n1}, B2 = {Bst[r0], Bst[r1]}, L2 = {Lst[r0], Lst[r1]};
            const f32x2 px = {pdx[r0], pdx[r1]}, py = {pdy[r0], pdy[r1]}, y2 = {yl[r0], yl[r1]}, yb2 = {yb, yb};
            const f32x2 sb = __builtin_elementwise_fma(nn, px, B2);
            const f32x2 sl = __builtin_elementwise_fma(nn, py, L2 + B2);
            const f32x2 nb = sb * yb2, nl = sl * y2;
            sB[r0] = sb.x; sB[r1] = sb.y; sL[r0] = sl.x; sL[r1] = sl.y;
            nB[r0] = nb.x; nB[r1] = nb.y; nL[r0] = nl.x; nL[r1] = nl.y;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float n = DIR == 0 ? (r == 0 ? nedge : Lst[r - 1]) : (r == R - 1 ? nedge : Lst[r + 1]);
            const f32x2 nn = {n, n}, pp = {pdx[r], pdy[r]}, base = {Bst[r], Lst[r] + Bst[r]}, yy = {yb, yl[r]};
            const f32x2 sum = __builtin_elementwise_fma(nn, pp, base);
            const f32x2 nxt = sum * yy;
            sB[r] = sum.x; sL[r] = sum.y;
            nB[r] = nxt.x; nL[r] = nxt.y;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { Bst[r] = nB[r]; Lst[r] = nL[r]; }
}

// Emission rows -> registers -> the LDS ring AS PROBABILITIES, straight from the ACTIVATIONS (round 4: no log-softmax
// pass in front of this kernel, no 2 x |acts| round trip through the workspace).  A batch is KU = 8 rows; eight lanes
// share a row (lane = 8 row + j holds classes j, j + 8, ...: NU = 8 registers cover K <= 64), the row's max and
// normaliser are two 3-step DPP butterflies inside the 8-lane group, p = 2^((x - m) log2 e) / z.  The alpha pass and the
// beta pass stage a row with the same instructions on the same inputs, so the replayed alpha states are bit-identical.
__device__ __forceinline__ float sa_group8_max(float v) {
    v = fmaxf(v, SA_DPP_F(v, v, 0xB1, 0xf));   // quad_perm [1, 0, 3, 2]
    v = fmaxf(v, SA_DPP_F(v, v, 0x4E, 0xf));   // quad_perm [2, 3, 0, 1]
    v = fmaxf(v, SA_DPP_F(v, v, 0x141, 0xf));  // row_half_mirror: the other quad of the group
    return v;
}
__device__ __forceinline__ float sa_group8_sum(float v) {
    v += SA_DPP_F(v, v, 0xB1, 0xf);
    v += SA_DPP_F(v, v, 0x4E, 0xf);
    v += SA_DPP_F(v, v, 0x141, 0xf);
    return v;
}
constexpr int kRingPitch = 9;                 // floats per class in the probability ring: [class][row of the batch], odd pitch
constexpr int kRingP = 65 * kRingPitch;       // one batch: K <= 64 classes x 8 rows, and column 64 = zeros: the emission of a
                                              // pair WITHOUT a label state (no select per step)
constexpr int kRingZero = 64 * kRingPitch;
template <int R, bool NORM, int NUE>
struct RowStagerP {
    static constexpr int KU = WaveCfg<R>::KU;
    static_assert(KU == 8 && (NUE == 4 || NUE == 8), "eight lanes per row, NUE classes per lane (K <= 8 NUE)");
    const float* x;  // the utterance's row 0 -- of the activations (NORM), or of the log-softmax K_A left in the workspace
    long st;         // floats between its consecutive rows
    int K, lane;
    float pv[NUE];
    __device__ __forceinline__ void issue(int tlo, int nrows) {
        const int r = lane >> 3, j = lane & 7;
        const float* src = x + (long)(tlo + min(r, nrows - 1)) * st;  // (rows past the batch: a copy of its last row, unused)
#pragma unroll
        for (int u = 0; u < NUE; ++u) pv[u] = src[min(j + 8 * u, K - 1)];
    }
    // The ring holds the batch TRANSPOSED: class c of row k at c * kRingPitch + k -- a state's emission for the 8 rows of a
    // batch sits at one register address plus an immediate (no address arithmetic per step), and distinct classes fall
    // on distinct banks (pitch 9, K <= 32; two-way from there).
    __device__ __forceinline__ void commit(float* dst) {
        const int r = lane >> 3, j = lane & 7;
        float* out = dst + j * kRingPitch + r;
        if constexpr (!NORM) {  // log2-probabilities
#pragma unroll
            for (int u = 0; u < NUE; ++u)
                if (j + 8 * u < K) out[8 * u * kRingPitch] = sa_exp2(pv[u]);
        } else {  // p = 2^((x - m) log2 e) / z: one exp2 per class, the normaliser by two 8-lane butterflies
            float m = -3.0e38f;
#pragma unroll
            for (int u = 0; u < NUE; ++u) m = j + 8 * u < K ? fmaxf(m, pv[u]) : m;
            m = sa_group8_max(m);
            float z = 0.f;
#pragma unroll
            for (int u = 0; u < NUE; ++u) {
                pv[u] = sa_exp2((pv[u] - m) * SA_LOG2E);
                z += j + 8 * u < K ? pv[u] : 0.f;
            }
            const float rz = 1.0f / sa_group8_sum(z);
#pragma unroll
            for (int u = 0; u < NUE; ++u)
                if (j + 8 * u < K) out[8 * u * kRingPitch] = pv[u] * rz;
        }
    }
};

template <int R, bool NORM, int NUE>
__device__ __forceinline__ void ctc_wave_p_body(const WaveArgs& A, int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int KU = WaveCfg<R>::KU, P = 64 * R;
    constexpr int kNoExp = -(1 << 28);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= A.B) return;  // waves are independent: no workgroup barrier anywhere in this kernel
    float* base = reinterpret_cast<float*>(smem_raw) + (long)wave * A.wave_lds_floats;
    float* ring = base;                 // [2][kRingP]: two batches of emission probabilities, [class][row] (RowStagerP)
    float* occ = ring + 2 * kRingP;     // sorted occupancies [64 R + 1]
    if (lane < 2 * kRingPitch) ring[(lane >= kRingPitch ? kRingP - kRingPitch : 0) + kRingZero + lane] = 0.f;
    const int K = A.K, L = A.label_lens[b], T = A.in_lens[b];
    int loff = 0;
    for (int i = lane; i < b; i += 64) loff += A.label_lens[i];
    loff = (int)sa_wave_sum_dpp((float)loff);  // label counts: exact in fp32 below 2^24
    const int* lab = A.labels + loff;
    float* ckb = A.stash + (long)b * A.nq * 3 * P + lane * R;  // checkpoints [nq][3][P]

    // ------------------------------------------------------------------------------------------------ alpha
    int a_lab[R], a_off[R];  // a_off: the label's column of the probability ring
    const int bl_off = A.blank * kRingPitch;
    bool a_ok[R];
    float skipf[R];
    float aB[R], aL[R];
    int ae[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = lane * R + r;
        a_ok[r] = j < L;
        a_lab[r] = a_ok[r] ? lab[j] : A.blank;
        a_off[r] = a_ok[r] ? a_lab[r] * kRingPitch : kRingZero;
        skipf[r] = (j >= 1 && j <= L - 1 && lab[j] != lab[j - 1]) ? 1.f : 0.f;
        aB[r] = j == 0 ? 1.0f : 0.f;
        aL[r] = 0.f;
        ae[r] = j == 0 ? 0 : kNoExp;
    }
    // batches of phase 1: the front moves one pair per step, and a repeated label costs it one more (label_j can reach
    // label_{j+1} = the same class only through the blank between them) -- L + 1 + repeats steps to cross the lattice
    float nrep_f = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = lane * R + r;
        nrep_f += (j >= 1 && j <= L - 1 && skipf[r] == 0.f) ? 1.f : 0.f;
    }
    const int nslow = (L + 1 + (int)sa_wave_sum_dpp(nrep_f) + 2 * KU) / KU;
    RowStagerP<R, NORM, NUE> stage;
    stage.x = NORM ? A.acts + (long)b * A.sb : A.ly2 + (long)b * A.ly_sb;
    stage.st = NORM ? A.st : (long)K;
    stage.K = K; stage.lane = lane;
    if (T > 0) { stage.issue(0, min(KU, T)); stage.commit(ring); }
    __builtin_amdgcn_s_waitcnt(0);
    bool suspect = false;
    // one batch of alpha steps from the state in (aB, aL, ae); optionally records the states after every step
    // (FAST: frozen exponents; FULL: all KU rows exist; REC: record the state after every step -- compile-time, so that a
    // whole batch is one branch-free block: a taken branch costs ~40 cycles of a dependent chain)
    float rB[KU][R], rL[KU][R];  // the replayed alpha states of a batch (beta pass; indexed at compile time only: a pointer
    int rE[KU][R];               // to these arrays would send them to scratch memory)
    auto alpha_batch_t = [&](auto fast_c, auto full_c, auto rec_c, int nrows, const float* cur) {
        constexpr bool FAST = decltype(fast_c)::value, FULL = decltype(full_c)::value, REC = decltype(rec_c)::value;
        float pdx[R], pdy[R], sB[R], sL[R];
        if (FAST) suspect = wavep_refresh<R, 0>(aB, aL, ae, skipf, pdx, pdy) || suspect;
#pragma unroll
        for (int k = 0; k < KU; ++k) {
            if (FULL || k < nrows) {
                const float yb = cur[bl_off + k];
                float yl[R];
#pragma unroll
                for (int r = 0; r < R; ++r) yl[r] = cur[a_off[r] + k];  // (no label state: the ring's zero column)
                if (FAST) wavep_fast_step<R, 0>(aB, aL, yl, yb, pdx, pdy, sB, sL);
                // (re-normalised EVERY step while the front crosses: on the logits a trained model emits an off-alignment
                // state loses 2^-29 per step, four steps between re-normalisations took a 2^10 hat below the smallest fp32, the
                // pair fell back to "empty" -- and an empty pair cuts the lattice for good once the exponents freeze:
                // the alignment's mass then never reaches the end, p = 0: the "alpha-dead" quarter of margin-20 batches)
                else wavep_slow_step<R, 0>(aB, aL, ae, yl, yb, skipf, sB, sL, true);
            }
            if (REC) {
#pragma unroll
                for (int r = 0; r < R; ++r) { rB[k][r] = aB[r]; rL[k][r] = aL[r]; rE[k][r] = ae[r]; }
            }
        }
    };
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;
    auto alpha_batch = [&](auto rec_c, int q, int nrows, const float* cur) {
        const bool fast = q >= nslow, full = nrows == KU;
        if (fast && full) alpha_batch_t(Yes(), Yes(), rec_c, nrows, cur);
        else if (fast) alpha_batch_t(Yes(), No(), rec_c, nrows, cur);
        else alpha_batch_t(No(), No(), rec_c, nrows, cur);
    };
    {
        int q = 0;
        for (int r0 = 0; r0 < T; r0 += KU, ++q) {
            const float* cur = ring + (q & 1) * kRingP;
            float* nxt = ring + ((q + 1) & 1) * kRingP;
            float* ck = ckb + (long)q * 3 * P;
#pragma unroll
            for (int r = 0; r < R; ++r) { ck[r] = aB[r]; ck[P + r] = aL[r]; ck[2 * P + r] = __builtin_bit_cast(float, ae[r]); }
            const bool more = r0 + KU < T;
            if (more) stage.issue(r0 + KU, min(KU, T - r0 - KU));
            alpha_batch(No(), q, min(KU, T - r0), cur);
            if (more) stage.commit(nxt);
        }
    }
    // p = alpha_T(blank_L) + alpha_T(label_{L-1}) as p_hat * 2^pe with p_hat in [0.5, 1)
    float f0 = 0.f, f1 = 0.f;
    int e0 = kNoExp, e1 = kNoExp;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = lane * R + r;
        if (j == L) { f0 = aB[r]; e0 = ae[r]; }
        if (j == L - 1) { f1 = aL[r]; e1 = ae[r]; }
    }
    f0 = sa_wave_max_dpp(f0); f1 = sa_wave_max_dpp(f1);   // one lane holds each; hats are >= 0
    e0 = (int)sa_wave_max_dpp((float)e0); e1 = (int)sa_wave_max_dpp((float)e1);  // |exponents| < 2^24 ... kNoExp is exact too
    const int em = max(f0 > 0.f ? e0 : kNoExp, f1 > 0.f ? e1 : kNoExp);
    const float psum = (f0 > 0.f ? __builtin_amdgcn_ldexpf(f0, e0 - em) : 0.f) + (f1 > 0.f ? __builtin_amdgcn_ldexpf(f1, e1 - em) : 0.f);
    const bool dead = !(psum > 0.f) || !(psum < 3.0e38f);  // no mass, NaN -- or an overflow in the last batch: all redone in the log domain
    const int pe = dead ? 0 : em + __builtin_amdgcn_frexp_expf(psum);
    const float ph = dead ? 1.f : __builtin_amdgcn_frexp_mantf(psum);
    if (lane == 0) {
        A.costs[b] = dead ? __builtin_inff() : (float)(-((double)sa_log2(ph) + (double)pe) * 0.6931471805599453);
        flags[b] = dead ? 1 : 0;  // every utterance writes its flag; "infeasible" is the log-domain kernel's call
    }
    const int T_live = (dead || T <= 0) ? 0 : T;
    if (T_live > 0) {
        // ------------------------------------------------------------------------------------------------- beta
        const float rcp = 1.0f / ph;
        int b_lab[R], b_off[R];
        bool b_ok[R];
        float bB[R], bL[R];
        int be[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = lane * R + r;
            b_ok[r] = j >= 1 && j - 1 < L;
            b_lab[r] = b_ok[r] ? lab[j - 1] : A.blank;
            b_off[r] = b_ok[r] ? b_lab[r] * kRingPitch : kRingZero;
            bB[r] = j == L ? 1.0f : 0.f;
            bL[r] = 0.f;
            be[r] = j == L ? 0 : kNoExp;
        }
        // label states counting-sorted by class (see ctc_wave_beta)
        int pos[R], seg_hi = P, seg_lo = P;
        {
            int cnt = 0, rank[R];
#pragma unroll
            for (int r = 0; r < R; ++r) rank[r] = 0;
            for (int i = 0; i < L; ++i) {
                const int li = lab[i];
                cnt += li == lane ? 1 : 0;
#pragma unroll
                for (int r = 0; r < R; ++r) rank[r] += (i < lane * R + r - 1 && li == b_lab[r]) ? 1 : 0;
            }
            const int incl = (int)wave_scan_dpp((float)cnt);
            const int start = incl - cnt;
            occ[lane] = (float)start;
            if (lane == 0) occ[P] = 0.f;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int j = lane * R + r;
                pos[r] = b_ok[r] ? (int)occ[b_lab[r]] + rank[r] : (j == 0 ? P - 1 : j - 1);
            }
            seg_lo = start > 0 ? start - 1 : P;
            seg_hi = cnt > 0 ? start + cnt - 1 : seg_lo;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const int q_last = (T - 1) / KU;
        float cB[R], cL[R];
        int cE[R];
        {
            const int tlo = q_last * KU;
            stage.issue(tlo, T - tlo);
            stage.commit(ring);
            const float* ck = ckb + (long)q_last * 3 * P;
#pragma unroll
            for (int r = 0; r < R; ++r) { cB[r] = ck[r]; cL[r] = ck[P + r]; cE[r] = __builtin_bit_cast(int, ck[2 * P + r]); }
        }
        __builtin_amdgcn_s_waitcnt(0);
        bool bad_row = false;
        int bi = 0;
        for (int q = q_last; q >= 0; --q, ++bi) {
            const int tlo = q * KU;
            const int nrows = min(KU, T - tlo);
            const float* cur = ring + (bi & 1) * kRingP;
            float* nxt = ring + ((bi + 1) & 1) * kRingP;
#pragma unroll
            for (int r = 0; r < R; ++r) { aB[r] = cB[r]; aL[r] = cL[r]; ae[r] = cE[r]; }
            if (q > 0) {  // the previous batch in time: rows and checkpoint, a batch ahead of their use
                stage.issue(tlo - KU, KU);
                const float* ck = ckb + (long)(q - 1) * 3 * P;
#pragma unroll
                for (int r = 0; r < R; ++r) { cB[r] = ck[r]; cL[r] = ck[P + r]; cE[r] = __builtin_bit_cast(int, ck[2 * P + r]); }
            }
            // (1) replay this batch's alpha states from its checkpoint
            alpha_batch(Yes(), q, nrows, cur);
            // (2) beta steps, backwards in time, with the occupancies and the gradient row of each step
            auto beta_batch_t = [&](auto fast_c, auto full_c, auto both_c) {
                // BOTH: the replayed alpha batch ran on frozen exponents too -- one scale per pair serves the batch's rows
                constexpr bool FAST = decltype(fast_c)::value, FULL = decltype(full_c)::value, BOTH = decltype(both_c)::value;
                float pdx[R], pdy[R];
                if (FAST) suspect = wavep_refresh<R, 1>(bB, bL, be, skipf, pdx, pdy) || suspect;
                float scb[R], scl[R];
                if constexpr (BOTH) {
                    const int aee = __builtin_bit_cast(int, sa_wave_shr1(__builtin_bit_cast(float, rE[0][R - 1]), __builtin_bit_cast(float, kNoExp)));
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int eb_ = be[r] == kNoExp ? 0 : be[r];
                        const int ea_b = rE[0][r] == kNoExp ? 0 : rE[0][r];
                        const int eal = r == 0 ? aee : rE[0][r - 1];
                        // (clamped so that the scale stays finite: 0 x scale is then 0 for a pair whose hats are zero and whose
                        // exponents mean nothing; hats that DECAYED inside the batch -- peaked emissions: 2^-35 per step --
                        // legitimately meet exponents far above 0, so the clamp sits at the top of the fp32 range)
                        scb[r] = __builtin_amdgcn_ldexpf(rcp, min(ea_b + eb_ - pe, 125));
                        scl[r] = b_ok[r] ? __builtin_amdgcn_ldexpf(rcp, min((eal == kNoExp ? 0 : eal) + eb_ - pe, 125)) : 0.f;
                    }
                }
#pragma unroll
                for (int k = KU - 1; k >= 0; --k) {
                    if (FULL || k < nrows) {
                        const int t = tlo + k;
                        const float yb = cur[bl_off + k];
                        float yl[R], sB[R], sL[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) yl[r] = cur[b_off[r] + k];
                        int es[R];  // the exponent the pre-emission sums sB, sL are expressed in
                        if (FAST) {
#pragma unroll
                            for (int r = 0; r < R; ++r) es[r] = be[r];
                            wavep_fast_step<R, 1>(bB, bL, yl, yb, pdx, pdy, sB, sL);
                        } else {
                            wavep_slow_step<R, 1>(bB, bL, be, yl, yb, skipf, sB, sL, true);
#pragma unroll
                            for (int r = 0; r < R; ++r) es[r] = be[r];  // the aligned exponent of the step (kNoExp: sums are 0)
                        }
                        // occupancy = alpha_t(s) * sum_beta_t(s) / p; alpha's label state of pair j-1 comes across the lane edge
                        const float aedge = sa_wave_shr1_z(rL[k][R - 1]);
                        float gb = 0.f;
                        if constexpr (BOTH) {
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                gb += rB[k][r] * sB[r] * scb[r];
                                const float al = r == 0 ? aedge : rL[k][r - 1];
                                occ[pos[r]] = al * sL[r] * scl[r];
                            }
                        } else {
                            const int aeedge = __builtin_bit_cast(int, sa_wave_shr1(__builtin_bit_cast(float, rE[k][R - 1]), __builtin_bit_cast(float, kNoExp)));
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const int eb_ = es[r] == kNoExp ? 0 : es[r];
                                const int ea_b = rE[k][r] == kNoExp ? 0 : rE[k][r];
                                gb += __builtin_amdgcn_ldexpf(rB[k][r] * sB[r] * rcp, ea_b + eb_ - pe);
                                const float al = r == 0 ? aedge : rL[k][r - 1];
                                const int eal = r == 0 ? aeedge : rE[k][r - 1];
                                // (a pair without a label state drops a zero into its slot past the sorted labels)
                                occ[pos[r]] = b_ok[r] ? __builtin_amdgcn_ldexpf(al * sL[r] * rcp, (eal == kNoExp ? 0 : eal) + eb_ - pe) : 0.f;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        float sv[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) sv[r] = occ[lane * R + r];
#pragma unroll
                        for (int r = 1; r < R; ++r) sv[r] += sv[r - 1];
                        const float incl = wave_scan_dpp(sv[R - 1]);
                        const float excl = incl - sv[R - 1];
                        // The blank's occupancy is what the label states leave of the row's unit flow.  That the flow IS one
                        // (the certificate of this pass) is checked with the blank states' own sum on the first and the last
                        // LIVE row of every batch -- where the frozen exponents of beta resp. of the replayed alpha have drifted
                        // furthest -- instead of on every row (a 64-lane reduction per row: 10 of the step's ~100 instructions).
                        const float lab_tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
                        if (k == 0 || k == nrows - 1)  // (a partial last batch ends at row nrows - 1, not KU - 1)
                            bad_row = bad_row || !(fabsf(sa_wave_sum_dpp(gb) + lab_tot - 1.0f) < 1e-4f);  // (NaN compares false)
                        bad_row = bad_row || !(lab_tot > -1e-4f && lab_tot < 1.0001f);  // every row: NaN / inf / a sum outside [0, 1]
                        const float ob = 1.0f - lab_tot;
#pragma unroll
                        for (int r = 0; r < R; ++r) occ[lane * R + r] = sv[r] + excl;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        const int c = min(lane, K - 1);  // K <= 64: lanes past K repeat lane K-1's store
                        const float od = occ[seg_hi] - occ[seg_lo];
                        const float o = c == A.blank ? ob : od;
                        float* g = A.grads + (long)b * A.sb + (long)t * A.st;
                        g[c] = (cur[c * kRingPitch + k] - o) * A.gscale;
                    }
                }
            };
            {
                const bool fast = bi >= nslow, full = nrows == KU;
                if (fast && full && q >= nslow) beta_batch_t(Yes(), Yes(), Yes());
                else if (fast && full) beta_batch_t(Yes(), Yes(), No());
                else if (fast) beta_batch_t(Yes(), No(), No());
                else beta_batch_t(No(), No(), No());
            }
            if (q > 0) stage.commit(nxt);
        }
        if ((bad_row || suspect) && lane == 0) flags[b] = 2;
    }
    for (int t = T_live; t < A.T_max; ++t) {  // padding rows / infeasible alignment: zero gradient
        float* g = A.grads + (long)b * A.sb + (long)t * A.st;
        for (int c = lane; c < K; c += 64) g[c] = 0.f;
    }
}

// R <= 2: four waves per SIMD (<= 128 registers) -- the kernel lives off the latency hiding of its neighbours
// NUE: classes per staging lane (4: K <= 32, every character-level model; 8: K <= 64)
template <int R, bool NORM, int NUE>
__global__ __launch_bounds__(256, 4) void ctc_wave_p_kernel(WaveArgs A, int* __restrict__ flags) {
    ctc_wave_p_body<R, NORM, NUE>(A, flags);
}
template <int R, bool NORM, int NUE>
__global__ __launch_bounds__(256) void ctc_wave_p_wide_kernel(WaveArgs A, int* __restrict__ flags) {
    ctc_wave_p_body<R, NORM, NUE>(A, flags);
}

template <int R, bool WITH_GRAD, bool SMALLK>
__global__ __launch_bounds__(256) void ctc_wave_kernel(WaveArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= A.B) return;  // waves are independent: no workgroup barrier anywhere in this kernel
    if (A.only && A.only[b] == 0) return;
    constexpr int KU = WaveCfg<R>::KU, NU = WaveCfg<R>::NU;
    float* base = reinterpret_cast<float*>(smem_raw) + (long)wave * A.wave_lds_floats;
    float* ring = base;                                       // [2][KU][K]  (SMALLK: [2][64 * NU])
    float* occ = ring + 2 * (SMALLK ? 64 * NU : KU * A.K);    // SMALLK: sorted occupancies [64 R + 1]; else row [K]
    float* aoff = occ + (SMALLK ? 64 * R + 1 : A.K);          // [nren]
    const int L = A.label_lens[b];
    const int T = A.in_lens[b];
    int loff = 0;
    for (int i = lane; i < b; i += 64) loff += A.label_lens[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loff += __shfl_xor(loff, o, 64);
    const int* lab = A.labels + loff;
    if (WITH_GRAD && !SMALLK)
        for (int c = lane; c < A.K; c += 64) occ[c] = 0.f;

    float lp = SA_NEG, lpo = 0.f;
    ctc_wave_alpha<R, WITH_GRAD, SMALLK>(A, b, lane, L, T, lab, ring, aoff, &lp, &lpo);
    const bool dead = lp < SA_NEG_TEST;
    if (lane == 0) A.costs[b] = dead ? __builtin_inff() : (float)(-((double)lp + (double)lpo) * 0.6931471805599453);
    if (!WITH_GRAD) return;
    const int T_live = (dead || T <= 0) ? 0 : T;
    // each lane reads back only the checkpoint slots it wrote itself: program order suffices, no fence
    if (T_live > 0) ctc_wave_beta<R, SMALLK>(A, b, lane, L, T, lab, ring, occ, aoff, lp, lpo);
    for (int t = T_live; t < A.T_max; ++t) {  // padding rows / infeasible alignment: zero gradient
        float* g = A.grads + (long)b * A.sb + (long)t * A.st;
        for (int c = lane; c < A.K; c += 64) g[c] = 0.f;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------ host side
static inline int ctc_nchunks(int max_L) { return (max_L + 1 + 63) / 64; }

static inline int ctc_nbatch(int max_T) { return (max_T + kU - 1) / kU + 1; }

static size_t ctc_ws_layout(int max_T, int max_L, int K, int B, size_t* off_ly2, size_t* off_stash, size_t* off_lp,
                            size_t* off_goffs, size_t* off_flags, size_t* off_lsort) {
    const int nch = ctc_nchunks(max_L);
    size_t o = 0;
    *off_ly2 = o;   o += sa_align_up((size_t)B * sa_align_up((size_t)max_T * K, 4) * sizeof(float), 256);
    *off_stash = o; o += sa_align_up((size_t)B * max_T * 6 * nch * 64 * sizeof(float), 256);
    *off_lp = o;    o += sa_align_up((size_t)B * sizeof(float) * 2, 256);
    *off_goffs = o; o += sa_align_up((size_t)B * 2 * nch * ctc_nbatch(max_T) * sizeof(float), 256);
    *off_flags = o; o += sa_align_up((size_t)(B + 1) * sizeof(int), 256);  // + the gated launch's arrival counter
    *off_lsort = o; o += sa_align_up((size_t)B * (1 + nch * 64 + K) * sizeof(int), 256);
    return o;
}

extern "C" size_t sa_ctc_workspace_bytes(int max_T, int max_L, int alphabet_size, int minibatch) {
    if (max_T < 0 || max_L < 0 || alphabet_size <= 0 || minibatch <= 0) return 0;
    size_t a, b, c, d, e, f;
    return ctc_ws_layout(max_T > 0 ? max_T : 1, max_L, alphabet_size, minibatch, &a, &b, &c, &d, &e, &f);
}

// diagnostic (tests): byte offset, inside a workspace of sa_ctc_workspace_bytes(...) bytes, of the int[minibatch] flags the
// probability-domain pass leaves (0 = its result stands; else the log-domain kernels redid the utterance)
extern "C" size_t sa_ctc_flags_offset(int max_T, int max_L, int alphabet_size, int minibatch) {
    size_t a, b, c, d, e = 0, f;
    if (max_T < 0 || max_L < 0 || alphabet_size <= 0 || minibatch <= 0) return 0;
    ctc_ws_layout(max_T > 0 ? max_T : 1, max_L, alphabet_size, minibatch, &a, &b, &c, &d, &e, &f);
    return e;
}

template <bool WITH_BETA, bool LDS_EM, bool PROB>
static ctcStatus_t launch_alphabeta(const AbArgs& A, int B, int threads, size_t smem, hipStream_t stream) {
    const void* fn = (const void*)ctc_alphabeta_kernel<WITH_BETA, LDS_EM, PROB>;
    if (smem > 48 * 1024 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return CTC_STATUS_EXECUTION_FAILED;
    hipLaunchKernelGGL((ctc_alphabeta_kernel<WITH_BETA, LDS_EM, PROB>), dim3(B), dim3(threads), smem, stream, A);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
template <bool PROB>
static ctcStatus_t launch_ab_any(const AbArgs& A, int B, int threads, size_t smem, bool with_beta, bool lds_em,
                                 hipStream_t stream) {
    if (with_beta)
        return lds_em ? launch_alphabeta<true, true, PROB>(A, B, threads, smem, stream)
                      : launch_alphabeta<true, false, PROB>(A, B, threads, smem, stream);
    return lds_em ? launch_alphabeta<false, true, PROB>(A, B, threads, smem, stream)
                  : launch_alphabeta<false, false, PROB>(A, B, threads, smem, stream);
}

namespace {
// *out = scale * sum_b costs[b], summed in a fixed order (one wave; B is a batch size)
__global__ void ctc_cost_sum_kernel(const float* __restrict__ costs, int B, float scale, float* __restrict__ out) {
    float v = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) v += costs[b];
    v = sa_wave_sum_dpp(v);
    if (threadIdx.x == 0) *out = v * scale;
}
ctcStatus_t ctc_loss_impl(const float* acts, float* grads, long stride_t, long stride_b, const int* d_flat_labels,
                          const int* d_label_lengths, const int* d_input_lengths, int alphabet_size, int minibatch,
                          int max_T, int max_L, int blank_label, float* d_costs, float grad_scale, void* workspace,
                          size_t workspace_bytes, void* stream_, float* d_loss = nullptr, bool* loss_done = nullptr);
}  // namespace

extern "C" ctcStatus_t sa_ctc_loss(const float* acts, float* grads, long stride_t, long stride_b,
                                   const int* d_flat_labels, const int* d_label_lengths,
                                   const int* d_input_lengths, int alphabet_size, int minibatch, int max_T,
                                   int max_L, int blank_label, float* d_costs, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    return ctc_loss_impl(acts, grads, stride_t, stride_b, d_flat_labels, d_label_lengths, d_input_lengths, alphabet_size,
                         minibatch, max_T, max_L, blank_label, d_costs, 1.0f, workspace, workspace_bytes, stream_);
}

extern "C" ctcStatus_t sa_ctc_loss_reduced(const float* acts, float* grads, long stride_t, long stride_b,
                                           const int* d_flat_labels, const int* d_label_lengths,
                                           const int* d_input_lengths, int alphabet_size, int minibatch, int max_T,
                                           int max_L, int blank_label, float scale, float* d_costs, float* d_loss,
                                           void* workspace, size_t workspace_bytes, void* stream_) {
    if (!d_loss) return CTC_STATUS_INVALID_VALUE;
    bool folded = false;  // the latency regime's gated launch sums the costs itself (AbArgs::loss_out)
    const ctcStatus_t s = ctc_loss_impl(acts, grads, stride_t, stride_b, d_flat_labels, d_label_lengths, d_input_lengths,
                                        alphabet_size, minibatch, max_T, max_L, blank_label, d_costs, scale, workspace,
                                        workspace_bytes, stream_, d_loss, &folded);
    if (s != CTC_STATUS_SUCCESS) return s;
    if (folded) return CTC_STATUS_SUCCESS;
    hipLaunchKernelGGL(ctc_cost_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, (const float*)d_costs, minibatch,
                       scale, d_loss);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

// y[i] *= *d_factor, skipped entirely when the factor is exactly 1 (what autograd hands a root loss): the reduced loss's
// gradient is already scaled, so loss.backward() costs one early-out launch and no pass over the gradient
namespace {
__global__ __launch_bounds__(256) void scale_by_device_scalar_kernel(float* __restrict__ y, size_t n,
                                                                     const float* __restrict__ d_factor) {
    const float f = *d_factor;
    if (f == 1.0f) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] *= f;
}
}  // namespace
extern "C" ctcStatus_t sa_scale_by_device_scalar(float* y, size_t n, const float* d_factor, void* stream_) {
    SA_CLEAR_ERR();
    if (!y || !d_factor) return CTC_STATUS_INVALID_VALUE;
    if (n == 0) return CTC_STATUS_SUCCESS;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, y, n, d_factor);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

// ---- opt-in chain profiler (bench.py): device-clock span of the alpha / beta kernel of every latency-regime call ----
// After sa_ctc_profile_configure(1) every sa_ctc_loss* call that runs ctc_alphabeta_kernel (B < 512) hands it a slot
// {earliest workgroup entry, latest workgroup exit} in 100 MHz ticks (atomic min / max: the span of the LAUNCH, not of one
// workgroup); sa_ctc_profile_read(i, ..) copies slot i back (SYNC) together with the call's T and B.  At most
// kCtcProfSlots calls are recorded per configure.  Process-wide state, one device (like sa_gru_profile_*).
namespace {
constexpr int kCtcProfSlots = 256;
struct CtcProf {
    unsigned long long* dev = nullptr;
    int n = 0, on = 0;
    int T[kCtcProfSlots], B[kCtcProfSlots];
} g_ctc_prof;
}  // namespace
extern "C" int sa_ctc_profile_configure(int enable) {
    CtcProf& P = g_ctc_prof;
    P.on = 0; P.n = 0;
    if (!enable) return 0;
    if (!P.dev && hipMalloc((void**)&P.dev, (size_t)kCtcProfSlots * 2 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long init[2 * kCtcProfSlots];
    for (int i = 0; i < kCtcProfSlots; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
    if (hipMemcpy(P.dev, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) return -1;
    P.on = 1;
    return 0;
}
extern "C" int sa_ctc_profile_count(void) { return g_ctc_prof.n; }
extern "C" int sa_ctc_profile_read(int i, double* chain_us, int* T, int* B) {
    CtcProf& P = g_ctc_prof;
    if (!P.dev || i < 0 || i >= P.n || !chain_us) return -1;
    unsigned long long v[2];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(v, P.dev + 2 * i, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    *chain_us = v[1] > v[0] ? (double)(v[1] - v[0]) * 0.01 : 0.0;
    if (T) *T = P.T[i];
    if (B) *B = P.B[i];
    return 0;
}

namespace {
ctcStatus_t ctc_loss_impl(const float* acts, float* grads, long stride_t, long stride_b, const int* d_flat_labels,
                          const int* d_label_lengths, const int* d_input_lengths, int alphabet_size, int minibatch,
                          int max_T, int max_L, int blank_label, float* d_costs, float grad_scale, void* workspace,
                          size_t workspace_bytes, void* stream_, float* d_loss, bool* loss_done) {
    SA_CLEAR_ERR();
    if (!acts || !d_flat_labels || !d_label_lengths || !d_input_lengths || !d_costs || !workspace)
        return CTC_STATUS_INVALID_VALUE;
    if (alphabet_size <= 0 || minibatch <= 0 || max_T <= 0 || max_L < 0 || blank_label < 0 ||
        blank_label >= alphabet_size)
        return CTC_STATUS_INVALID_VALUE;
    const int nch = ctc_nchunks(max_L);
    if (nch > kMaxChunks) return CTC_STATUS_INVALID_VALUE;  // labels longer than 511: not supported
    if (workspace_bytes < sa_ctc_workspace_bytes(max_T, max_L, alphabet_size, minibatch))
        return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    const int K = alphabet_size, B = minibatch;
    size_t o_ly2, o_stash, o_lp, o_goffs, o_flags, o_lsort;
    ctc_ws_layout(max_T, max_L, K, B, &o_ly2, &o_stash, &o_lp, &o_goffs, &o_flags, &o_lsort);
    char* ws = (char*)workspace;

    AbArgs A;
    A.ly2 = (float*)(ws + o_ly2);
    A.labels = d_flat_labels;
    A.label_lens = d_label_lengths;
    A.in_lens = d_input_lengths;
    A.K = K; A.T_max = max_T; A.blank = blank_label; A.nchunks = nch; A.Ppad = nch * 64;
    A.hand_stride = (int)sa_align_up((size_t)max_T + kU + 1, 4);
    A.nbatch = ctc_nbatch(max_T);
    A.ly_sb = (long)sa_align_up((size_t)max_T * K, 4);
    A.stash = (float*)(ws + o_stash);
    A.goffs = (float*)(ws + o_goffs);
    A.logp2_out = (float*)(ws + o_lp);
    A.costs = d_costs;
    A.flags = (int*)(ws + o_flags);
    A.lsort = (int*)(ws + o_lsort);
    A.gate = 0;
    A.no_fast = 0;
    A.grads = grads; A.g_st = stride_t; A.g_sb = stride_b; A.gscale = grad_scale;
    A.dbg = sa_opt(SA_OPT_CTC_DBG) ? (unsigned long long*)((char*)workspace + o_goffs) : nullptr;  // overwrites goffs[0..2]: debug only
    A.prof = nullptr;
    A.loss_out = nullptr; A.loss_scale = grad_scale; A.done_ctr = (unsigned*)(ws + o_flags) + B;

    // K_A (log-softmax into the workspace); only != null: the utterances whose flag is set
    auto launch_ka = [&](const int* only) -> ctcStatus_t {
        if (K <= 64 && (long)B * max_T >= 256 * 1024) {  // K_A, one lane per row out of an LDS tile: the
            // throughput form; with fewer than ~4 workgroups per CU its serial 3 * K-step row loop is the slower one
            dim3 grid((max_T + kRowTile - 1) / kRowTile, B);
            hipLaunchKernelGGL(ctc_logsoftmax2_rows_kernel, grid, dim3(kRowTile), (size_t)kRowTile * (K | 1) * sizeof(float), stream,
                               acts, stride_t, stride_b, d_input_lengths, K, A.ly_sb, (float*)(ws + o_ly2), only);
            SA_CHECK_LAUNCH();
        } else {  // K_A, a lane group per row
            const int G = K <= 16 ? 16 : (K <= 32 ? 32 : 64);
            const int rows_per_block = 4 * (64 / G);
            dim3 grid((max_T + rows_per_block - 1) / rows_per_block, B);
            float* ly2 = (float*)(ws + o_ly2);
            if (G == 16)
                hipLaunchKernelGGL(ctc_logsoftmax2_kernel<16>, grid, dim3(256), 0, stream, acts, stride_t, stride_b,
                                   d_input_lengths, K, A.ly_sb, ly2, only);
            else if (G == 32)
                hipLaunchKernelGGL(ctc_logsoftmax2_kernel<32>, grid, dim3(256), 0, stream, acts, stride_t, stride_b,
                                   d_input_lengths, K, A.ly_sb, ly2, only);
            else
                hipLaunchKernelGGL(ctc_logsoftmax2_kernel<64>, grid, dim3(256), 0, stream, acts, stride_t, stride_b,
                                   d_input_lengths, K, A.ly_sb, ly2, only);
            SA_CHECK_LAUNCH();
        }
        return CTC_STATUS_SUCCESS;
    };
    // K_W: the throughput-regime kernel (one wave per utterance) once every CU holds several utterances
    {
        int R = 1;
        while (R * 64 < max_L + 1) R *= 2;
        const int KU = R <= 4 ? 8 : 4;
        const int nren = max_T / 32 + 2;
        const bool smallk = KU * K <= 512 && K <= 64;  // a batch of emission rows fits the 8 staging registers
        // (smallk: two batches of emissions -- 2 x 512 floats row-major in the log-domain kernel, 2 x kRingP class-major in
        // the probability-domain one -- and the sorted occupancies)
        const size_t wave_floats = sa_align_up((size_t)2 * (smallk ? kRingP : KU * K) + (smallk ? 64 * R + 1 : K) + nren, 4);
        const size_t wave_bytes = wave_floats * sizeof(float);
        const long wide_opt = sa_opt(SA_OPT_CTC_WIDE);
        const bool want = wide_opt >= 0 ? wide_opt == 1 : (B >= kWideMinBatch);
        const bool wide = want && wave_bytes <= 150 * 1024;
        // probability-domain pass first (gradient calls, K <= 64, R <= 4), the log-domain kernel behind it for flagged
        // utterances; option ctc.prob = 0: log domain only, =2: everything flagged (tests), =3: probability pass alone
        const long prob_opt = sa_opt(SA_OPT_CTC_PROB);
        const int prob = (wide && grads && smallk && R <= 4) ? (prob_opt >= 0 ? (int)prob_opt : 1) : 0;
        // ... and that pass normalises the activations itself while it stages them (RowStagerP): K_A then runs BEHIND it, for
        // the flagged utterances only (normally none: ~65 K workgroups that read one word and leave)
        // Measured (tools/ctc_b4096_time.py, one box): B = 4096 1.53 - 1.65 -> 1.37 - 1.49 ms per call, B = 1024 level, B = 512
        // 0.59 -> 0.62 (too few waves to hide the longer staging) -- from kDirectMinBatch utterances; option ctc.direct = 1 / 0 forces.
        const long direct_opt = sa_opt(SA_OPT_CTC_DIRECT);
        const bool direct = prob != 0 && grads != acts && (direct_opt >= 0 ? direct_opt == 1 : B >= kDirectMinBatch);
        if (!direct) {
            ctcStatus_t ks = launch_ka(nullptr);
            if (ks != CTC_STATUS_SUCCESS) return ks;
        }
        if (wide) {
            const int waves = 4 * wave_bytes <= 64 * 1024 ? 4 : 1;
            WaveArgs W;
            W.acts = direct ? acts : nullptr; W.ly2 = A.ly2; W.labels = d_flat_labels; W.label_lens = d_label_lengths; W.in_lens = d_input_lengths;
            W.K = K; W.T_max = max_T; W.blank = blank_label; W.B = B; W.nren = nren; W.nq = (max_T + KU - 1) / KU;
            W.wave_lds_floats = (int)wave_floats;
            W.ly_sb = A.ly_sb; W.stash = A.stash; W.costs = d_costs; W.grads = grads;
            W.st = stride_t; W.sb = stride_b; W.only = nullptr; W.gscale = grad_scale;
            const size_t smem = waves * wave_bytes;
            const dim3 grid((B + waves - 1) / waves), block(64 * waves);
            if (prob) {
                void (*pf)(WaveArgs, int*) = nullptr;
#define SA_PF(N_, U_) (R == 1 ? ctc_wave_p_kernel<1, N_, U_> : R == 2 ? ctc_wave_p_kernel<2, N_, U_> : ctc_wave_p_wide_kernel<4, N_, U_>)
                if (direct) pf = K <= 32 ? SA_PF(true, 4) : SA_PF(true, 8);
                else pf = K <= 32 ? SA_PF(false, 4) : SA_PF(false, 8);
#undef SA_PF
                if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)pf, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                            (int)smem) != hipSuccess)
                    return CTC_STATUS_EXECUTION_FAILED;
                hipLaunchKernelGGL(pf, grid, block, smem, stream, W, A.flags);
                SA_CHECK_LAUNCH();
                if (prob == 2 && hipMemsetAsync(A.flags, 1, (size_t)B * sizeof(int), stream) != hipSuccess)
                    return CTC_STATUS_MEMOPS_FAILED;
                if (prob == 3) return CTC_STATUS_SUCCESS;
                W.only = A.flags;
                if (direct) {
                    ctcStatus_t ks = launch_ka(A.flags);
                    if (ks != CTC_STATUS_SUCCESS) return ks;
                }
            }
#define SA_WIDE_LAUNCH(R_)                                                                                        \
    do {                                                                                                          \
        void (*fn)(WaveArgs) = grads ? (smallk ? ctc_wave_kernel<R_, true, true> : ctc_wave_kernel<R_, true, false>)    \
                                     : (smallk ? ctc_wave_kernel<R_, false, true> : ctc_wave_kernel<R_, false, false>); \
        if (smem > 48 * 1024 && hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                                    (int)smem) != hipSuccess)                                     \
            return CTC_STATUS_EXECUTION_FAILED;                                                                   \
        hipLaunchKernelGGL(fn, grid, block, smem, stream, W);                                                     \
    } while (0)
            if (R == 1) SA_WIDE_LAUNCH(1);
            else if (R == 2) SA_WIDE_LAUNCH(2);
            else if (R == 4) SA_WIDE_LAUNCH(4);
            else SA_WIDE_LAUNCH(8);
#undef SA_WIDE_LAUNCH
            SA_CHECK_LAUNCH();
            return CTC_STATUS_SUCCESS;
        }
    }
    {  // K_B
        // LDS of ctc_alphabeta_kernel by domain (see its layout); the emission copy joins when it fits
        auto ab_fixed_lds = [&](bool prob) {
            const size_t slots = (size_t)2 * (nch - 1);  // hand-off arrays: chunks with a consumer only (ctc_hand_slot)
            const size_t floats = prob ? slots * 2 * A.hand_stride + (size_t)2 * nch * 128
                                       : slots * A.hand_stride + ((slots * A.nbatch + 3) & ~(size_t)3) +
                                             (size_t)2 * nch * ((A.Ppad + 1 + 3) & ~3);
            return kAbSharedBytes + (floats + ((K + 1 + 3) & ~3)) * sizeof(float);
        };
        const size_t em_bytes = sa_align_up((size_t)(max_T + 2 * kU) * K * sizeof(float), 16);
        auto ab_lds_em = [&](bool prob) { return ab_fixed_lds(prob) + em_bytes <= 156 * 1024; };
        auto ab_smem = [&](bool prob) { return ab_fixed_lds(prob) + (ab_lds_em(prob) ? em_bytes : 0); };
        if (ab_smem(false) > 160 * 1024) return CTC_STATUS_INVALID_VALUE;
        const int threads = 64 * nch * (grads ? 2 : 1);
        // probability-domain chain first (option ctc.prob = 0: log domain only; =2: flag every utterance, which exercises the
        // hand-over in tests), then the log-domain kernels for whatever it flagged (see ctc_chain_p).  A score-only call
        // has no rows to check: it stays in the log domain.
        const long prob_opt = sa_opt(SA_OPT_CTC_PROB);
        int prob = !grads ? 0 : (prob_opt >= 0 ? (int)prob_opt : 1);
        // the probability-domain chain needs its emissions in LDS to pay (its hand-off arrays are twice the log domain's:
        // from three chunks at T = 1000 they do not fit beside them) -- then the log-domain kernels run alone
        if (prob && (!ab_lds_em(true) || ab_smem(true) > 160 * 1024) && !(prob_opt >= 2)) prob = 0;
        ctcStatus_t s;
        const dim3 ggrid((max_T + 3) / 4, B);
        const size_t gsmem = 4 * (size_t)(A.Ppad + 1) * sizeof(float);
        if (g_ctc_prof.on && g_ctc_prof.n < kCtcProfSlots) {  // the first alpha / beta launch of this call is the one stamped
            CtcProf& Q = g_ctc_prof;
            A.prof = Q.dev + 2 * Q.n; Q.T[Q.n] = max_T; Q.B[Q.n] = B; ++Q.n;
        }
        if (prob) {
            if (ab_smem(true) > 160 * 1024) return CTC_STATUS_INVALID_VALUE;
            s = launch_ab_any<true>(A, B, threads, ab_smem(true), true, ab_lds_em(true), stream);
            A.prof = nullptr;
            if (s != CTC_STATUS_SUCCESS) return s;
            hipLaunchKernelGGL(ctc_grad_kernel<true>, ggrid, dim3(256), gsmem, stream, A, grads, stride_t, stride_b);
            SA_CHECK_LAUNCH();
            if (prob == 2 && hipMemsetAsync(A.flags, 1, (size_t)B * sizeof(int), stream) != hipSuccess)
                return CTC_STATUS_MEMOPS_FAILED;
            A.gate = 1;
            if (prob == 3) return CTC_STATUS_SUCCESS;  // debug: the probability-domain pass alone, flags left for inspection
            if (d_loss && loss_done) { A.loss_out = d_loss; *loss_done = true; }  // the gated launch below folds the costs
        }
        s = launch_ab_any<false>(A, B, threads, ab_smem(false), grads != nullptr, ab_lds_em(false), stream);
        if (s != CTC_STATUS_SUCCESS) return s;
        if (grads && !A.gate) {  // K_C (the hand-over pass writes its rows itself)
            hipLaunchKernelGGL(ctc_grad_kernel<false>, ggrid, dim3(256), gsmem, stream, A, grads, stride_t, stride_b);
            SA_CHECK_LAUNCH();
        }
    }
    return CTC_STATUS_SUCCESS;
}
}  // namespace

// ----------------------------------------------------------------------------------- the warp-ctc-shaped entry points
extern "C" int get_warpctc_version(void) { return 2; }

extern "C" const char* ctcGetStatusString(ctcStatus_t status) {
    switch (status) {
        case CTC_STATUS_SUCCESS: return "no error";
        case CTC_STATUS_MEMOPS_FAILED: return "device memory operation failed";
        case CTC_STATUS_INVALID_VALUE: return "invalid value";
        case CTC_STATUS_EXECUTION_FAILED: return "execution failed";
        case CTC_STATUS_UNKNOWN_ERROR:
        default: return "unknown error";
    }
}

static bool host_shape(const int* label_lengths, const int* input_lengths, int B, int* max_T, int* max_L,
                       long* total_L) {
    *max_T = 0; *max_L = 0; *total_L = 0;
    for (int b = 0; b < B; ++b) {
        if (label_lengths[b] < 0 || input_lengths[b] < 0) return false;
        if (input_lengths[b] > *max_T) *max_T = input_lengths[b];
        if (label_lengths[b] > *max_L) *max_L = label_lengths[b];
        *total_L += label_lengths[b];
    }
    return true;
}

extern "C" ctcStatus_t get_workspace_size(const int* label_lengths, const int* input_lengths, int alphabet_size,
                                          int minibatch, ctcOptions options, size_t* size_bytes) {
    SA_CLEAR_ERR();
    if (!label_lengths || !input_lengths || !size_bytes || alphabet_size <= 0 || minibatch <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if (options.loc != CTC_GPU) return CTC_STATUS_EXECUTION_FAILED;
    int max_T, max_L;
    long total_L;
    if (!host_shape(label_lengths, input_lengths, minibatch, &max_T, &max_L, &total_L))
        return CTC_STATUS_INVALID_VALUE;
    if (ctc_nchunks(max_L) > kMaxChunks) return CTC_STATUS_INVALID_VALUE;
    if (max_T < 1) max_T = 1;
    *size_bytes = sa_ctc_workspace_bytes(max_T, max_L, alphabet_size, minibatch) +
                  sa_align_up((size_t)(total_L + 1) * sizeof(int), 256) +
                  2 * sa_align_up((size_t)minibatch * sizeof(int), 256) +
                  sa_align_up((size_t)minibatch * sizeof(float), 256);
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t compute_ctc_loss(const float* activations, float* gradients, const int* flat_labels,
                                        const int* label_lengths, const int* input_lengths, int alphabet_size,
                                        int minibatch, float* costs, void* workspace, ctcOptions options) {
    SA_CLEAR_ERR();
    if (!activations || !flat_labels || !label_lengths || !input_lengths || !costs || !workspace ||
        alphabet_size <= 0 || minibatch <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if (options.loc != CTC_GPU) return CTC_STATUS_EXECUTION_FAILED;
    const int blank = options.blank_label;
    if (blank < 0 || blank >= alphabet_size) return CTC_STATUS_INVALID_VALUE;
    int max_T, max_L;
    long total_L;
    if (!host_shape(label_lengths, input_lengths, minibatch, &max_T, &max_L, &total_L))
        return CTC_STATUS_INVALID_VALUE;
    for (long i = 0; i < total_L; ++i)
        if (flat_labels[i] < 0 || flat_labels[i] >= alphabet_size || flat_labels[i] == blank)
            return CTC_STATUS_INVALID_VALUE;
    if (ctc_nchunks(max_L) > kMaxChunks) return CTC_STATUS_INVALID_VALUE;
    if (max_T < 1) max_T = 1;
    hipStream_t stream = (hipStream_t)options.stream;
    const size_t core = sa_ctc_workspace_bytes(max_T, max_L, alphabet_size, minibatch);
    char* ws = (char*)workspace;
    int* d_labels = (int*)(ws + core);
    int* d_llen = (int*)((char*)d_labels + sa_align_up((size_t)(total_L + 1) * sizeof(int), 256));
    int* d_ilen = (int*)((char*)d_llen + sa_align_up((size_t)minibatch * sizeof(int), 256));
    float* d_costs = (float*)((char*)d_ilen + sa_align_up((size_t)minibatch * sizeof(int), 256));
    if (total_L > 0 &&
        hipMemcpyAsync(d_labels, flat_labels, total_L * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    if (hipMemcpyAsync(d_llen, label_lengths, minibatch * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(d_ilen, input_lengths, minibatch * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    // warp-ctc layout: (T, B, V) row-major
    ctcStatus_t s = sa_ctc_loss(activations, gradients, (long)minibatch * alphabet_size, (long)alphabet_size,
                                d_labels, d_llen, d_ilen, alphabet_size, minibatch, max_T, max_L, blank, d_costs,
                                workspace, core, stream);
    if (s != CTC_STATUS_SUCCESS) return s;
    if (hipMemcpyAsync(costs, d_costs, minibatch * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    if (hipStreamSynchronize(stream) != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;
    return CTC_STATUS_SUCCESS;
}
