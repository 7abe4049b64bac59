// transducer.hip -- RNN-Transducer loss forward (alpha) / backward (beta) / gradient for gfx950.
//
// Replaces the `transducer` extension the reference calls at /root/reference/speech/models/transducer_model.py:50-51
// (transducer.functions.transducer.TransducerLoss; out-of-tree, un-pinned: see include/speech_amd.h section 7 for the
// provenance; the algorithm is Graves 2012, eqs. 16-20).  Input: the LOG-softmax lattice (B, T, U+1, K) the
// model produces (:76-77); output: per-utterance cost -log p and the gradient with respect to the lattice entries.
//
// Design:
//   T_A  tr_gather      every cell (t, u) in parallel: the two arc weights that matter, lp[t,u,blank] and
//                       lp[t,u,y[u]], are pulled out of the K-wide rows into two DIAGONAL-MAJOR planes
//                       [d = t + u][u] (log2 domain).  All cells of an anti-diagonal are independent, so this is
//                       the layout in which a wave reads one contiguous row per step.
//   T_B  tr_alphabeta   one workgroup per utterance, wave 0 = alpha, wave 1 = beta (concurrent: the serial chain is
//                       T + U steps, not 2 (T + U)).  Lane i owns R contiguous label positions u; a step is one
//                       anti-diagonal:   alpha: a(u) <- lse(a(u) + blank(d-1,u), [a(u-1) + label(d-1,u-1)])
//                                        beta:  b(u) <- lse(b(u) + blank(d,u),   b(u+1) + label(d,u))
//                       with ONE DPP wave shift for the lane-boundary neighbour.  Same machinery as the CTC kernels:
//                       finite sentinel for log 0, exact integer-offset renormalisation every 8 steps, arc weights
//                       prefetched a batch of 8 diagonals ahead, unconditional VMEM accesses in the loop.
//   T_C  tr_grad        every cell in parallel: the two non-zero gradient entries of the cell's row,
//                       -exp2(alpha + w + beta_next - log2 p), scattered into the zero-filled dense gradient.
#include "common.h"

namespace {

constexpr int kTU = 8;  // diagonals per prefetch batch
constexpr int kPS = 3;  // log2 of the renormalisation period: states drift ~5 log2 units per diagonal, so every 8
                        // diagonals keeps them below ~64 where fp32 resolves 4e-6 (32 diagonals gave 3.4e-5 errors)

struct TrArgs {
    const float* lp;        // (B, T_max, U1_max, K)
    const int* labels;
    const int* label_lens;
    const int* in_lens;
    int K, T_max, U1_max, blank, B;
    int Up;                 // padded row width of the diagonal-major planes: 64 * R
    int D;                  // diagonals per utterance: T_max + U1_max - 1, rounded up to a multiple of kTU, + kTU
    float* wB;              // [B][D][Up]  log2 lp[t,u,blank]   (NEG where the blank arc does not exist: t = T-1)
    float* wL;              // [B][D][Up]  log2 lp[t,u,y[u]]    (NEG where u >= U)
    float* alpha;           // [B][D][Up]
    float* beta;            // [B][D][Up]
    float* offs;            // [B][2][(D >> kPS) + 2]: integer offsets of alpha / beta per 2^kPS-diagonal period
    float* endB;            // [B] log2 lp[T-1, U, blank]
    float* logp2;           // [B][2]: log2 p = hat + offset
    float* costs;
};

__device__ __forceinline__ float tr_lse2(float a, float b) {
    const float m = fmaxf(a, b);
    return m + sa_log2(sa_exp2(a - m) + sa_exp2(b - m));  // both operands <= m: the sum is in [1, 2]
}

// ---------------------------------------------------------------------------------------------------------- T_A
// grid (ceil(Up / 64), T_max, B), 64 threads: thread = label position u of one time step.
__global__ __launch_bounds__(64) void tr_gather_kernel(TrArgs A) {
    const int b = blockIdx.z, t = blockIdx.y;
    const int u = blockIdx.x * 64 + threadIdx.x;
    const int T = A.in_lens[b], U = A.label_lens[b];
    if (u >= A.Up || t >= T) return;  // rows of later diagonals were pre-filled with the sentinel
    int loff = 0;
    for (int i = 0; i < b; ++i) loff += A.label_lens[i];
    float vb = SA_NEG, vl = SA_NEG;
    if (u <= U) {
        const float* row = A.lp + (((long)b * A.T_max + t) * A.U1_max + u) * A.K;
        const float lb = fmaxf(row[A.blank] * SA_LOG2E, SA_NEG);
        if (t < T - 1) vb = lb;
        if (t == T - 1 && u == U) A.endB[b] = lb;
        if (u < U) vl = fmaxf(row[A.labels[loff + u]] * SA_LOG2E, SA_NEG);
    }
    const long o = ((long)b * A.D + (t + u)) * A.Up + u;
    A.wB[o] = vb;
    A.wL[o] = vl;
}

__global__ void tr_fill_kernel(float* p, long n, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------------- T_B
// The T loop runs over FULL batches of kTU diagonals with no branch around any VMEM access (loads of the next batch's
// arcs, fire-and-forget stores of the states): the compiler can then wait for exactly the prefetched loads
// (vmcnt(N)) instead of vmcnt(0), which would put the HBM latency of the batch's stores on every batch.  The ragged
// last batch is a separate block after the loop; the per-period offsets go to LDS and are dumped once at the end.
template <int R, bool WITH_BETA>
__global__ __launch_bounds__(128) void tr_alphabeta_kernel(TrArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = A.in_lens[b], U = A.label_lens[b];
    const int nd = T + U;  // diagonals 0 .. T+U-1
    const int noff = (A.D >> kPS) + 2;
    const long base = (long)b * A.D * A.Up + lane * R;
    const float* wB = A.wB + base;
    const float* wL = A.wL + base;
    float* s_offs = reinterpret_cast<float*>(smem_raw) + dir * noff;
    for (int i = lane; i < noff; i += 64) s_offs[i] = 0.f;

    float st[R];
    float off = 0.f;
    auto renorm = [&](int period) {
        float m = SA_NEG;
#pragma unroll
        for (int r = 0; r < R; ++r) m = fmaxf(m, st[r]);
        m = sa_wave_max_dpp(m);
        const float dlt = m > SA_NEG_TEST ? __builtin_rintf(m) : 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) st[r] -= dlt;
        off += dlt;
        if (lane == 0) s_offs[period] = off;
    };
    float pb[kTU][R], pl[kTU][R];

    if (dir == 0) {
        // alpha[0,0] = 0 on diagonal 0; step s = 1 .. nd-1 produces diagonal s from diagonal s-1's values and arcs
        float* out = A.alpha + base;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            st[r] = (lane * R + r == 0) ? 0.f : SA_NEG;
            if (WITH_BETA) out[r] = st[r];
        }
        auto issue = [&](int d0) {  // arcs leaving diagonals d0 .. d0+kTU-1 (the planes are padded: no bounds test)
#pragma unroll
            for (int k = 0; k < kTU; ++k)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    pb[k][r] = wB[(long)(d0 + k) * A.Up + r];
                    pl[k][r] = wL[(long)(d0 + k) * A.Up + r];
                }
        };
        auto step = [&](int s, const float (&cb)[R], const float (&cl)[R]) {  // s = destination diagonal
            if ((s & ((1 << kPS) - 1)) == 0) renorm(s >> kPS);
            float viaL[R];  // a(u) + label arc: consumed by position u + 1
#pragma unroll
            for (int r = 0; r < R; ++r) viaL[r] = st[r] + cl[r];
            const float edge = sa_wave_shr1(viaL[R - 1], SA_NEG);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float fromL = r == 0 ? edge : viaL[r - 1];
                st[r] = fmaxf(tr_lse2(st[r] + cb[r], fromL), SA_NEG);
            }
            if (WITH_BETA) {
                float* o = out + (long)s * A.Up;
#pragma unroll
                for (int r = 0; r < R; ++r) o[r] = st[r];
            }
        };
        issue(0);
        __builtin_amdgcn_s_waitcnt(0);
        int s0 = 1;
        for (; s0 + kTU <= nd; s0 += kTU) {
            float cb[kTU][R], cl[kTU][R];
#pragma unroll
            for (int k = 0; k < kTU; ++k)
#pragma unroll
                for (int r = 0; r < R; ++r) { cb[k][r] = pb[k][r]; cl[k][r] = pl[k][r]; }
            issue(s0 - 1 + kTU);
#pragma unroll
            for (int k = 0; k < kTU; ++k) step(s0 + k, cb[k], cl[k]);
        }
#pragma unroll
        for (int k = 0; k < kTU; ++k)
            if (s0 + k < nd) step(s0 + k, pb[k], pl[k]);
        // log2 p = alpha[T-1, U] + blank(T-1, U): position U on the last diagonal
        float f = SA_NEG;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (lane * R + r == U) f = st[r];
        f = sa_wave_max_dpp(f);
        if (lane == 0) {
            const float lp = f + A.endB[b];
            const bool dead = lp < SA_NEG_TEST;
            A.logp2[2 * b] = dead ? SA_NEG : lp;
            A.logp2[2 * b + 1] = dead ? 0.f : off;
            A.costs[b] = dead ? __builtin_inff() : (float)(-((double)lp + (double)off) * 0.6931471805599453);
        }
    } else if (WITH_BETA) {
        // beta[T-1,U] = blank(T-1,U) on the last diagonal; the step to diagonal d uses diagonal d's own arcs
        float* out = A.beta + base;
        const float eB = A.endB[b];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            st[r] = (lane * R + r == U) ? eB : SA_NEG;
            out[(long)(nd - 1) * A.Up + r] = st[r];
        }
        auto issue = [&](int dhi) {  // arcs of diagonals dhi, dhi-1, .. (clamped at 0: the extra rows are never used)
#pragma unroll
            for (int k = 0; k < kTU; ++k)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const long row = (long)max(dhi - k, 0) * A.Up + r;
                    pb[k][r] = wB[row];
                    pl[k][r] = wL[row];
                }
        };
        int cur_period = (nd - 1) >> kPS;
        auto step = [&](int d, const float (&cb)[R], const float (&cl)[R]) {
            if ((d >> kPS) != cur_period) {  // entering a new 2^kPS-diagonal period: renormalise, record its offset
                cur_period = d >> kPS;
                renorm(cur_period);
            }
            const float edge = sa_wave_shl1(st[0], SA_NEG);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float right = r == R - 1 ? edge : st[r + 1];
                st[r] = fmaxf(tr_lse2(st[r] + cb[r], right + cl[r]), SA_NEG);
            }
            float* o = out + (long)d * A.Up;
#pragma unroll
            for (int r = 0; r < R; ++r) o[r] = st[r];
        };
        issue(nd - 2);
        __builtin_amdgcn_s_waitcnt(0);
        int d0 = nd - 2;
        for (; d0 - kTU + 1 >= 0; d0 -= kTU) {
            float cb[kTU][R], cl[kTU][R];
#pragma unroll
            for (int k = 0; k < kTU; ++k)
#pragma unroll
                for (int r = 0; r < R; ++r) { cb[k][r] = pb[k][r]; cl[k][r] = pl[k][r]; }
            issue(d0 - kTU);
#pragma unroll
            for (int k = 0; k < kTU; ++k) step(d0 - k, cb[k], cl[k]);
        }
#pragma unroll
        for (int k = 0; k < kTU; ++k)
            if (d0 - k >= 0) step(d0 - k, pb[k], pl[k]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float* offs = A.offs + ((long)b * 2 + dir) * noff;
    for (int i = lane; i < noff; i += 64) offs[i] = s_offs[i];
}

// ---------------------------------------------------------------------------------------------------------- T_C
// grid (ceil(Up / 64), T_max, B), 64 threads: one cell per thread; the dense gradient was zero-filled before.
__global__ __launch_bounds__(64) void tr_grad_kernel(TrArgs A, float* __restrict__ grads) {
    const int b = blockIdx.z, t = blockIdx.y;
    const int u = blockIdx.x * 64 + threadIdx.x;
    const int T = A.in_lens[b], U = A.label_lens[b];
    if (t >= T || u > U) return;
    const float lp = A.logp2[2 * b], lpo = A.logp2[2 * b + 1];
    if (lp < SA_NEG_TEST) return;  // no alignment: zero gradient
    int loff = 0;
    for (int i = 0; i < b; ++i) loff += A.label_lens[i];
    const int noff = (A.D >> kPS) + 2;
    const float* oa = A.offs + (long)b * 2 * noff;
    const float* ob = oa + noff;
    const int d = t + u;
    const long o = ((long)b * A.D + d) * A.Up + u;
    const float a = A.alpha[o];
    const float ao = oa[d >> kPS];
    float* g = grads + (((long)b * A.T_max + t) * A.U1_max + u) * A.K;
    if (t < T - 1) {
        const float io = ao + ob[(d + 1) >> kPS] - lpo;  // integers: exact
        g[A.blank] = -sa_exp2((a + A.wB[o] + A.beta[o + A.Up] - lp) + io);
    } else if (u == U) {
        g[A.blank] = -sa_exp2((a + A.endB[b] - lp) + (ao - lpo));
    }
    if (u < U) {
        const float io = ao + ob[(d + 1) >> kPS] - lpo;
        g[A.labels[loff + u]] = -sa_exp2((a + A.wL[o] + A.beta[o + A.Up + 1] - lp) + io);
    }
}

// ---------------------------------------------------------------------------------------------------------- T_D
// Static beam search over one utterance's lattice (the call at transducer_model.py:98; the callee is in the un-vendored
// package, restated in the header's section 7): the search never re-runs the prediction network, so a hypothesis with u
// labels reads row u of the lattice.  Step i of T+U-2 extends every beam entry (hyp, score), u = len(hyp), t = i - u,
// by a blank (t < T-1: same hyp) or by any label (u < U-1); equal hypotheses merge by log-sum-exp; the best `beam`
// survive a stable descending sort.  Scores are double precision with double exp / log, as the Python original's.
// One wave per utterance, lane = class (K <= 64), candidates c[entry][class] in LDS.  Hypotheses are nodes of a
// prefix tree (parent, symbol); a hypothesis can be reached twice in one step only as "stay on A" + "extend A's parent
// by A's last symbol", which is found by looking A's parent up in the beam -- no hash table.  The dict's insertion
// order (beam order, then class order) is the candidate's sequence number e * K + v; a merged key keeps the smaller.
constexpr int kDecMaxBeam = 16;

struct DecArgs {
    const float* lp;        // (B, T_max, U1_max, K)
    const int* T_arr;       // [B] frames to search
    const int* U_arr;       // [B] lattice rows to search (label length + 1)
    int K, T_max, U1_max, beam, blank, max_nodes;
    int* node_parent;       // [B][max_nodes]
    int* node_sym;          // [B][max_nodes]
    int* out_labels;        // [B][U1_max]
    int* out_lens;          // [B]
    double* out_scores;     // [B]
};

// the prefix tree lives in global memory and is written by lane 0, read by every lane: device-scope accesses
// (they bypass the per-CU L1, which a plain load after another lane's store could hit stale)
__device__ __forceinline__ int dec_ld(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dec_st(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double dec_lse(double a, double b) {
    const double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

__global__ __launch_bounds__(64) void tr_decode_static_kernel(DecArgs A) {
    __shared__ double c[kDecMaxBeam][64];      // candidate scores of this step (-inf: none)
    __shared__ int exist[kDecMaxBeam][64];     // candidate is an existing node (its id) or -1
    __shared__ int b_node[2][kDecMaxBeam], b_len[2][kDecMaxBeam];
    __shared__ double b_score[2][kDecMaxBeam];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int T = A.T_arr[b], U = A.U_arr[b], K = A.K;
    const float* lp = A.lp + (long)b * A.T_max * A.U1_max * K;
    int* parent = A.node_parent + (long)b * A.max_nodes;
    int* sym = A.node_sym + (long)b * A.max_nodes;
    const double NINF = -__builtin_inf();
    int nbeam = 1, nnodes = 1, cur = 0;
    if (lane == 0) { b_node[0][0] = 0; b_len[0][0] = 0; b_score[0][0] = 0.0; dec_st(parent, -1); dec_st(sym, -1); }
    __syncthreads();

    for (int i = 0; i < T + U - 2; ++i) {
        // (1) candidates: lane = class v
        for (int e = 0; e < nbeam; ++e) {
            const int u = b_len[cur][e], t = i - u;
            double v = NINF;
            if (t >= 0 && t <= T - 1 && lane < K) {
                const bool ok = lane == A.blank ? (t < T - 1) : (u < U - 1);
                if (ok) v = b_score[cur][e] + (double)lp[((long)t * A.U1_max + u) * K + lane];
            }
            c[e][lane] = v;
            exist[e][lane] = -1;
        }
        __syncthreads();
        // (2) merges: "stay on A" with "extend parent(A) by sym(A)"   (uniform control flow, lane 0 writes)
        for (int a = 0; a < nbeam; ++a) {
            const int na = b_node[cur][a];
            const int pa = dec_ld(parent + na), sa = dec_ld(sym + na);
            if (pa < 0) continue;
            int bidx = -1;
            for (int e = 0; e < nbeam; ++e)
                if (b_node[cur][e] == pa) bidx = e;
            if (bidx < 0) continue;
            if (lane == 0) {
                const double ext = c[bidx][sa], stay = c[a][A.blank];
                if (ext != NINF) {
                    if (stay == NINF) {
                        exist[bidx][sa] = na;  // the extension re-creates the existing hypothesis A
                    } else {
                        const double m = dec_lse(stay, ext);  // two contributions at most: order-independent
                        const bool ext_first = bidx * K + sa < a * K + A.blank;
                        if (ext_first) { c[bidx][sa] = m; exist[bidx][sa] = na; c[a][A.blank] = NINF; }
                        else { c[a][A.blank] = m; c[bidx][sa] = NINF; }
                    }
                }
            }
            __syncthreads();
        }
        __syncthreads();
        // (3) stable top-`beam`: repeated wave arg-max over (score desc, sequence number asc)
        const int nxt = cur ^ 1;
        int nnew = 0;
        for (int w = 0; w < A.beam; ++w) {
            double best = NINF;
            int bseq = 0x7fffffff;
            for (int e = 0; e < nbeam; ++e) {
                const double v = c[e][lane];
                if (v > best) { best = v; bseq = e * K + lane; }  // e ascending: the first maximum has the smaller seq
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64);
                const int os = __shfl_xor(bseq, o, 64);
                if (ob > best || (ob == best && os < bseq)) { best = ob; bseq = os; }
            }
            if (best == NINF) break;  // uniform: every lane holds the same winner
            const int e = bseq / K, v = bseq - e * K;
            if (lane == 0) {
                int node, len;
                if (v == A.blank) { node = b_node[cur][e]; len = b_len[cur][e]; }
                else {
                    len = b_len[cur][e] + 1;
                    node = exist[e][v];
                    if (node < 0) {
                        node = nnodes;
                        dec_st(parent + node, b_node[cur][e]);
                        dec_st(sym + node, v);
                    }
                }
                b_node[nxt][nnew] = node; b_len[nxt][nnew] = len; b_score[nxt][nnew] = best;
                c[e][v] = NINF;
            }
            if (v != A.blank && exist[e][v] < 0) ++nnodes;  // uniform (LDS value read by all lanes before lane 0 moves on)
            ++nnew;
            __syncthreads();
        }
        nbeam = nnew;
        cur = nxt;
        __syncthreads();
        if (nbeam == 0) break;
    }
    if (lane == 0) {
        int n = 0;
        double score = NINF;
        if (nbeam > 0) {
            int node = b_node[cur][0];
            n = b_len[cur][0];
            score = b_score[cur][0] + (double)lp[((long)(T - 1) * A.U1_max + (U - 1)) * K + A.blank];
            int* out = A.out_labels + (long)b * A.U1_max;
            for (int k = n - 1; k >= 0; --k) { out[k] = dec_ld(sym + node); node = dec_ld(parent + node); }
        }
        A.out_lens[b] = n;
        A.out_scores[b] = score;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------ host side
static inline int tr_R(int max_U1) {
    int R = 1;
    while (64 * R < max_U1) R *= 2;
    return R;
}
static inline int tr_D(int max_T, int max_U1) { return (int)sa_align_up((size_t)max_T + max_U1 - 1, kTU) + 2 * kTU; }

static size_t tr_layout(int max_T, int max_U1, int B, size_t o[8]) {
    const size_t plane = sa_align_up((size_t)B * tr_D(max_T, max_U1) * 64 * tr_R(max_U1) * sizeof(float), 256);
    size_t p = 0;
    for (int i = 0; i < 4; ++i) { o[i] = p; p += plane; }
    o[4] = p; p += sa_align_up((size_t)B * 2 * ((tr_D(max_T, max_U1) >> kPS) + 2) * sizeof(float), 256);
    o[5] = p; p += sa_align_up((size_t)B * sizeof(float), 256);
    o[6] = p; p += sa_align_up((size_t)B * 2 * sizeof(float), 256);
    return p;
}

extern "C" size_t sa_transducer_workspace_bytes(int max_T, int max_U1, int minibatch) {
    if (max_T <= 0 || max_U1 <= 0 || minibatch <= 0 || max_U1 > 512) return 0;
    size_t o[8];
    return tr_layout(max_T, max_U1, minibatch, o);
}

extern "C" ctcStatus_t sa_transducer_loss(const float* log_probs, float* grads, const int* d_flat_labels,
                                          const int* d_label_lengths, const int* d_input_lengths, int alphabet_size,
                                          int minibatch, int max_T, int max_U1, int blank_label, float* d_costs,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    if (!log_probs || !d_flat_labels || !d_label_lengths || !d_input_lengths || !d_costs || !workspace)
        return CTC_STATUS_INVALID_VALUE;
    if (alphabet_size <= 0 || minibatch <= 0 || max_T <= 0 || max_U1 <= 0 || max_U1 > 512 || blank_label < 0 ||
        blank_label >= alphabet_size)
        return CTC_STATUS_INVALID_VALUE;
    if (workspace_bytes < sa_transducer_workspace_bytes(max_T, max_U1, minibatch)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    size_t o[8];
    tr_layout(max_T, max_U1, minibatch, o);
    char* ws = (char*)workspace;
    const int R = tr_R(max_U1);
    TrArgs A;
    A.lp = log_probs; A.labels = d_flat_labels; A.label_lens = d_label_lengths; A.in_lens = d_input_lengths;
    A.K = alphabet_size; A.T_max = max_T; A.U1_max = max_U1; A.blank = blank_label; A.B = minibatch;
    A.Up = 64 * R; A.D = tr_D(max_T, max_U1);
    A.wB = (float*)(ws + o[0]); A.wL = (float*)(ws + o[1]); A.alpha = (float*)(ws + o[2]); A.beta = (float*)(ws + o[3]);
    A.offs = (float*)(ws + o[4]); A.endB = (float*)(ws + o[5]); A.logp2 = (float*)(ws + o[6]);
    A.costs = d_costs;

    const long plane = (long)minibatch * A.D * A.Up;
    hipLaunchKernelGGL(tr_fill_kernel, dim3(1024), dim3(256), 0, stream, A.wB, 2 * plane, SA_NEG);  // wB and wL
    hipLaunchKernelGGL(tr_fill_kernel, dim3(64), dim3(256), 0, stream, A.endB, (long)minibatch, SA_NEG);
    SA_CHECK_LAUNCH();
    const dim3 cells((A.Up + 63) / 64, max_T, minibatch);
    hipLaunchKernelGGL(tr_gather_kernel, cells, dim3(64), 0, stream, A);
    SA_CHECK_LAUNCH();
    const size_t off_smem = (size_t)2 * ((A.D >> kPS) + 2) * sizeof(float);
#define SA_TR_LAUNCH(R_)                                                                                            \
    do {                                                                                                            \
        if (grads) hipLaunchKernelGGL((tr_alphabeta_kernel<R_, true>), dim3(minibatch), dim3(128), off_smem, stream, A); \
        else hipLaunchKernelGGL((tr_alphabeta_kernel<R_, false>), dim3(minibatch), dim3(64), off_smem, stream, A);  \
    } while (0)
    if (R == 1) SA_TR_LAUNCH(1);
    else if (R == 2) SA_TR_LAUNCH(2);
    else if (R == 4) SA_TR_LAUNCH(4);
    else SA_TR_LAUNCH(8);
#undef SA_TR_LAUNCH
    SA_CHECK_LAUNCH();
    if (grads) {
        if (hipMemsetAsync(grads, 0, (size_t)minibatch * max_T * max_U1 * alphabet_size * sizeof(float), stream) !=
            hipSuccess)
            return CTC_STATUS_MEMOPS_FAILED;
        hipLaunchKernelGGL(tr_grad_kernel, cells, dim3(64), 0, stream, A, grads);
        SA_CHECK_LAUNCH();
    }
    return CTC_STATUS_SUCCESS;
}

extern "C" size_t sa_transducer_decode_workspace_bytes(int max_T, int max_U1, int minibatch, int beam_size) {
    if (max_T <= 0 || max_U1 <= 0 || minibatch <= 0 || beam_size <= 0 || beam_size > kDecMaxBeam) return 0;
    const size_t max_nodes = (size_t)(max_T + max_U1) * beam_size + 2;
    return 2 * sa_align_up((size_t)minibatch * max_nodes * sizeof(int), 256);
}

extern "C" ctcStatus_t sa_transducer_decode_static(const float* log_probs, const int* d_T, const int* d_U1,
                                                   int alphabet_size, int minibatch, int max_T, int max_U1,
                                                   int beam_size, int blank_label, int* d_out_labels, int* d_out_lens,
                                                   double* d_out_scores, void* workspace, size_t workspace_bytes,
                                                   void* stream) {
    SA_CLEAR_ERR();
    if (!log_probs || !d_T || !d_U1 || !d_out_labels || !d_out_lens || !d_out_scores || !workspace)
        return CTC_STATUS_INVALID_VALUE;
    if (alphabet_size <= 0 || alphabet_size > 64 || minibatch <= 0 || max_T <= 0 || max_U1 <= 0 || beam_size <= 0 ||
        beam_size > kDecMaxBeam || blank_label < 0 || blank_label >= alphabet_size)
        return CTC_STATUS_INVALID_VALUE;
    const size_t need = sa_transducer_decode_workspace_bytes(max_T, max_U1, minibatch, beam_size);
    if (workspace_bytes < need) return CTC_STATUS_INVALID_VALUE;
    DecArgs A;
    A.lp = log_probs; A.T_arr = d_T; A.U_arr = d_U1; A.K = alphabet_size; A.T_max = max_T; A.U1_max = max_U1;
    A.beam = beam_size; A.blank = blank_label; A.max_nodes = (max_T + max_U1) * beam_size + 2;
    A.node_parent = (int*)workspace;
    A.node_sym = (int*)((char*)workspace + need / 2);
    A.out_labels = d_out_labels; A.out_lens = d_out_lens; A.out_scores = d_out_scores;
    hipLaunchKernelGGL(tr_decode_static_kernel, dim3(minibatch), dim3(64), 0, (hipStream_t)stream, A);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
