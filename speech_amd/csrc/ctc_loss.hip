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
//                          serial chain is T steps, not 2T) and stash alpha/beta planes to HBM.
//   K_C  ctc_grad          all rows in parallel again: occupancy gamma = exp2(alpha + beta - ly2 - log2 p),
//                          blank states by a wave reduction, label states by LDS float atomics,
//                          grad = softmax - occupancy written with the caller's strides.
// log(0) is the finite sentinel SA_NEG, so no inf/NaN guards sit on the dependent chain.
#include "common.h"

namespace {

constexpr int kU = 8;          // steps per hand-off batch (producer lead, emission prefetch depth)
constexpr int kMaxChunks = 8;  // chunks per direction: labels up to 64 * 8 - 1 = 511 per utterance

// ---------------------------------------------------------------------------------------------------------- K_A
// G lanes cooperate on one row (G = 16, 32 or 64); a wave handles 64 / G rows at a time.
template <int G>
__global__ __launch_bounds__(256) void ctc_logsoftmax2_kernel(const float* __restrict__ acts, long st, long sb,
                                                              const int* __restrict__ in_lens, int K, int B, int T,
                                                              float* __restrict__ ly2) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int sub = lane / G, gl = lane % G;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    const long rows = (long)B * T;
    for (long r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
        const long r = r0 + sub;
        const bool live = r < rows;
        const int b = live ? (int)(r / T) : 0;
        const int t = live ? (int)(r % T) : 0;
        const bool act = live && t < in_lens[b];
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
            float* o = ly2 + r * K;
            for (int k = gl; k < K; k += G) o[k] = fmaxf((a[k] - m) * SA_LOG2E - lz, SA_NEG);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- K_B
struct AbShared {
    int prog[2][kMaxChunks];  // steps completed by (direction, chunk)
    float fin[2];             // alpha[T-1, 2L], alpha[T-1, 2L-1]
    int label_off;
    int timeout;
};

__device__ __forceinline__ float lse2_1p(float a, float b) {  // log2(2^a + 2^b)
    const float m = fmaxf(a, b);
    return m + sa_log2(1.0f + sa_exp2(fminf(a, b) - m));
}
__device__ __forceinline__ float lse3_1p(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    const float md = __builtin_amdgcn_fmed3f(a, b, c);
    const float mn = fminf(fminf(a, b), c);
    return m + sa_log2(1.0f + sa_exp2(md - m) + sa_exp2(mn - m));
}

template <bool WITH_BETA>
__global__ __launch_bounds__(1024) void ctc_alphabeta_kernel(const float* __restrict__ ly2,
                                                             const int* __restrict__ labels,
                                                             const int* __restrict__ label_lens,
                                                             const int* __restrict__ in_lens, int K, int T_max,
                                                             int blank, int nchunks, int Ppad, int hand_stride,
                                                             float* __restrict__ stash, float* __restrict__ logp2_out,
                                                             float* __restrict__ costs) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AbShared* sh = reinterpret_cast<AbShared*>(smem_raw);
    float* hand_all = reinterpret_cast<float*>(smem_raw + 128);  // [2][nchunks][hand_stride]

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = wave / nchunks;          // 0 = alpha (forward in time), 1 = beta (backward in time)
    const int chunk = wave - dir * nchunks;  // which 64 pairs
    const int L = label_lens[b];
    const int T = in_lens[b];

    if (wave == 0) {  // flat-label offset of this utterance
        int acc = 0;
        for (int i = lane; i < b; i += 64) acc += label_lens[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            sh->label_off = acc;
            sh->timeout = 0;
            sh->fin[0] = SA_NEG;
            sh->fin[1] = SA_NEG;
        }
    }
    float* hand_w = hand_all + (long)(dir * nchunks + chunk) * hand_stride;  // what this chunk publishes
    if (lane == 0) {
        sh->prog[dir][chunk] = 0;
        hand_w[0] = SA_NEG;
    }
    __syncthreads();

    const int* lab = labels + sh->label_off;
    const int j = chunk * 64 + lane;  // pair index: (blank_j, label_j) for alpha, (label_{j-1}, blank_j) for beta
    const int own = dir == 0 ? j : j - 1;
    const bool own_ok = own >= 0 && own < L;
    const int own_lab = own_ok ? lab[own] : blank;
    const bool skip = (j >= 1 && j <= L - 1) ? (lab[j] != lab[j - 1]) : false;
    const bool in_lattice = j <= L;

    // producer / consumer wiring of the chunk pipeline
    const int prod_chunk = dir == 0 ? chunk - 1 : chunk + 1;
    const bool has_prod = prod_chunk >= 0 && prod_chunk < nchunks;
    const bool has_cons = dir == 0 ? (chunk + 1 < nchunks) : (chunk > 0);
    const float* hand_r = hand_all + (long)(dir * nchunks + (has_prod ? prod_chunk : 0)) * hand_stride;
    const int edge_lane = dir == 0 ? 63 : 0;

    float Bst = (dir == 0 ? (j == 0) : (j == L)) ? 0.0f : SA_NEG;
    float Lst = SA_NEG;

    // Running pointers: time moves forward for alpha, backward for beta.
    const int t0 = dir == 0 ? 0 : (T > 0 ? T - 1 : 0);
    const long dK = dir == 0 ? (long)K : -(long)K;
    const long dS = dir == 0 ? 4L * Ppad : -4L * Ppad;
    const float* lyb = ly2 + (long)b * T_max * K;
    const float* erow = lyb + (long)t0 * K;  // emission row of the next step to prefetch
    float* pB = nullptr;
    float* pL = nullptr;
    if (WITH_BETA) {
        // stash planes per (b, t): [alpha blank | alpha label | beta blank | beta label], Ppad floats each.
        // beta's label state belongs to pair j-1; lanes without a label state dump into a never-read slot
        // (label index Ppad-1 cannot exist because Ppad >= L+1).
        float* base = stash + ((long)b * T_max + t0) * 4 * Ppad;
        pB = base + (dir == 0 ? 0 : 2) * Ppad + j;
        pL = base + (dir == 0 ? 1 : 3) * Ppad + (own >= 0 ? own : Ppad - 1);
    }

    float el[kU], eb[kU];
    auto load_emissions = [&](int r0, float* pel, float* peb) {
        if (r0 + kU <= T) {  // full batch: no clamping
#pragma unroll
            for (int k = 0; k < kU; ++k) {
                peb[k] = erow[blank];
                pel[k] = erow[own_lab];
                erow += dK;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kU; ++k) {
                const bool ok = r0 + k < T;
                peb[k] = ok ? erow[blank] : 0.f;
                pel[k] = ok ? erow[own_lab] : 0.f;
                if (ok) erow += dK;
            }
        }
    };
    load_emissions(0, el, eb);

    auto do_step = [&](float e_l_raw, float e_b, float h, int r) {
        const float n = dir == 0 ? sa_wave_shr1(Lst, h) : sa_wave_shl1(Lst, h);
        const float e_l = own_ok ? e_l_raw : SA_NEG;
        const float nB = e_b + lse2_1p(Bst, n);
        const float nL = e_l + lse3_1p(Lst, Bst, skip ? n : SA_NEG);
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
    for (int r0 = 0; r0 < T; r0 += kU) {
        float nel[kU], neb[kU];
        load_emissions(r0 + kU, nel, neb);  // prefetch the next batch's emissions

        float hv[kU];
        if (has_prod) {
            const int need = min(r0 + kU, T);
            if (avail < need) {
                int spins = 0;
                while ((avail = __hip_atomic_load(&sh->prog[dir][prod_chunk], __ATOMIC_ACQUIRE,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 24)) {  // bounded: never hang the GPU on a protocol bug
                        sh->timeout = 1;
                        break;
                    }
                }
            }
            const float4 h0 = *reinterpret_cast<const float4*>(hand_r + r0);
            const float4 h1 = *reinterpret_cast<const float4*>(hand_r + r0 + 4);
            hv[0] = h0.x; hv[1] = h0.y; hv[2] = h0.z; hv[3] = h0.w;
            hv[4] = h1.x; hv[5] = h1.y; hv[6] = h1.z; hv[7] = h1.w;
        } else {
#pragma unroll
            for (int k = 0; k < kU; ++k) hv[k] = SA_NEG;
        }

        if (r0 + kU <= T) {
#pragma unroll
            for (int k = 0; k < kU; ++k) do_step(el[k], eb[k], hv[k], r0 + k);
        } else {
#pragma unroll
            for (int k = 0; k < kU; ++k)
                if (r0 + k < T) do_step(el[k], eb[k], hv[k], r0 + k);
        }
        if (has_cons && lane == 0)
            __hip_atomic_store(&sh->prog[dir][chunk], min(r0 + kU, T), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 0; k < kU; ++k) { el[k] = nel[k]; eb[k] = neb[k]; }
    }

    if (dir == 0) {
        if (j == L) sh->fin[0] = Bst;
        if (j == L - 1) sh->fin[1] = Lst;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lp = lse2_1p(sh->fin[0], sh->fin[1]);
        const bool dead = lp < SA_NEG_TEST;
        logp2_out[b] = dead ? SA_NEG : lp;
        float c = dead ? __builtin_inff() : -lp * SA_LN2;
        if (sh->timeout) c = __builtin_nanf("");
        costs[b] = c;
    }
}

// ---------------------------------------------------------------------------------------------------------- K_C
constexpr int kGradRowsPerBlock = 16;

__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ ly2, const float* __restrict__ stash,
                                                       const float* __restrict__ logp2,
                                                       const int* __restrict__ labels,
                                                       const int* __restrict__ label_lens,
                                                       const int* __restrict__ in_lens, int K, int T_max, int blank,
                                                       int nchunks, int Ppad, float* __restrict__ grads, long st,
                                                       long sb) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* occ_all = reinterpret_cast<float*>(smem_raw);  // [4 waves][K]
    __shared__ int s_off;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int L = label_lens[b];
    const int T = in_lens[b];
    if (wave == 0) {
        int acc = 0;
        for (int i = lane; i < b; i += 64) acc += label_lens[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) s_off = acc;
    }
    __syncthreads();
    const int* lab = labels + s_off;
    int mylab[kMaxChunks];
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int j = c * 64 + lane;
        mylab[c] = (c < nchunks && j < L) ? lab[j] : -1;
    }
    const float lp = logp2[b];
    const bool dead = lp < SA_NEG_TEST;
    float* occ = occ_all + wave * K;

    const int t_end = min(T_max, (int)(blockIdx.x + 1) * kGradRowsPerBlock);
    for (int t = blockIdx.x * kGradRowsPerBlock + wave; t < t_end; t += 4) {
        float* g = grads + (long)b * sb + (long)t * st;
        if (t >= T || dead) {
            for (int k = lane; k < K; k += 64) g[k] = 0.f;
            continue;
        }
        const float* row = ly2 + ((long)b * T_max + t) * K;
        const float* sp = stash + ((long)b * T_max + t) * 4 * Ppad;
        for (int k = lane; k < K; k += 64) occ[k] = 0.f;
        __threadfence_block();
        const float lyblank = row[blank];
        float accB = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) {
            if (c < nchunks) {
                const int j = c * 64 + lane;
                if (j <= L) accB += sa_exp2(sp[j] + sp[2 * Ppad + j] - lyblank - lp);
                if (mylab[c] >= 0) {
                    const float gm = sa_exp2(sp[Ppad + j] + sp[3 * Ppad + j] - row[mylab[c]] - lp);
                    atomicAdd(&occ[mylab[c]], gm);  // LDS float atomic; one wave owns this row
                }
            }
        }
        accB = sa_wave_sum(accB);
        __threadfence_block();
        for (int k = lane; k < K; k += 64) {
            const float y = sa_exp2(row[k]);
            const float o = (k == blank) ? accB : occ[k];
            g[k] = y - o;
        }
        __threadfence_block();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------ host side
static inline int ctc_nchunks(int max_L) { return (max_L + 1 + 63) / 64; }

static size_t ctc_ws_layout(int max_T, int max_L, int K, int B, size_t* off_ly2, size_t* off_stash, size_t* off_lp,
                            size_t* off_ints) {
    const int nch = ctc_nchunks(max_L);
    size_t o = 0;
    *off_ly2 = o;   o += sa_align_up((size_t)B * max_T * K * sizeof(float), 256);
    *off_stash = o; o += sa_align_up((size_t)B * max_T * 4 * nch * 64 * sizeof(float), 256);
    *off_lp = o;    o += sa_align_up((size_t)B * sizeof(float) * 2, 256);
    *off_ints = o;  // scratch for the warp-ctc-shaped entry: labels + 2 length vectors (sized by the caller there)
    return o;
}

extern "C" size_t sa_ctc_workspace_bytes(int max_T, int max_L, int alphabet_size, int minibatch) {
    if (max_T < 0 || max_L < 0 || alphabet_size <= 0 || minibatch <= 0) return 0;
    size_t a, b, c, d;
    return ctc_ws_layout(max_T > 0 ? max_T : 1, max_L, alphabet_size, minibatch, &a, &b, &c, &d);
}

extern "C" ctcStatus_t sa_ctc_loss(const float* acts, float* grads, long stride_t, long stride_b,
                                   const int* d_flat_labels, const int* d_label_lengths,
                                   const int* d_input_lengths, int alphabet_size, int minibatch, int max_T,
                                   int max_L, int blank_label, float* d_costs, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
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
    const int K = alphabet_size, B = minibatch, Ppad = nch * 64;
    size_t o_ly2, o_stash, o_lp, o_ints;
    ctc_ws_layout(max_T, max_L, K, B, &o_ly2, &o_stash, &o_lp, &o_ints);
    char* ws = (char*)workspace;
    float* ly2 = (float*)(ws + o_ly2);
    float* stash = (float*)(ws + o_stash);
    float* logp2 = (float*)(ws + o_lp);

    {  // K_A
        const long rows = (long)B * max_T;
        const int G = K <= 16 ? 16 : (K <= 32 ? 32 : 64);
        const long waves = (rows + (64 / G) - 1) / (64 / G);
        int grid = (int)((waves + 3) / 4);
        if (grid > 4096) grid = 4096;
        if (grid < 1) grid = 1;
        if (G == 16)
            hipLaunchKernelGGL(ctc_logsoftmax2_kernel<16>, dim3(grid), dim3(256), 0, stream, acts, stride_t,
                               stride_b, d_input_lengths, K, B, max_T, ly2);
        else if (G == 32)
            hipLaunchKernelGGL(ctc_logsoftmax2_kernel<32>, dim3(grid), dim3(256), 0, stream, acts, stride_t,
                               stride_b, d_input_lengths, K, B, max_T, ly2);
        else
            hipLaunchKernelGGL(ctc_logsoftmax2_kernel<64>, dim3(grid), dim3(256), 0, stream, acts, stride_t,
                               stride_b, d_input_lengths, K, B, max_T, ly2);
        SA_CHECK_LAUNCH();
    }
    {  // K_B
        const int hand_stride = (int)sa_align_up((size_t)max_T + kU + 1, 4);
        const size_t smem = 128 + (size_t)2 * nch * hand_stride * sizeof(float);
        if (smem > 160 * 1024) return CTC_STATUS_INVALID_VALUE;
        if (grads) {
            if (smem > 48 * 1024)
                if (hipFuncSetAttribute((const void*)ctc_alphabeta_kernel<true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                    return CTC_STATUS_EXECUTION_FAILED;
            hipLaunchKernelGGL(ctc_alphabeta_kernel<true>, dim3(B), dim3(64 * nch * 2), smem, stream, ly2,
                               d_flat_labels, d_label_lengths, d_input_lengths, K, max_T, blank_label, nch, Ppad,
                               hand_stride, stash, logp2, d_costs);
        } else {
            if (smem > 48 * 1024)
                if (hipFuncSetAttribute((const void*)ctc_alphabeta_kernel<false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                    return CTC_STATUS_EXECUTION_FAILED;
            hipLaunchKernelGGL(ctc_alphabeta_kernel<false>, dim3(B), dim3(64 * nch), smem, stream, ly2,
                               d_flat_labels, d_label_lengths, d_input_lengths, K, max_T, blank_label, nch, Ppad,
                               hand_stride, stash, logp2, d_costs);
        }
        SA_CHECK_LAUNCH();
    }
    if (grads) {  // K_C
        dim3 grid((max_T + kGradRowsPerBlock - 1) / kGradRowsPerBlock, B);
        hipLaunchKernelGGL(ctc_grad_kernel, grid, dim3(256), 4 * (size_t)K * sizeof(float), stream, ly2, stash,
                           logp2, d_flat_labels, d_label_lengths, d_input_lengths, K, max_T, blank_label, nch, Ppad,
                           grads, stride_t, stride_b);
        SA_CHECK_LAUNCH();
    }
    return CTC_STATUS_SUCCESS;
}

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
