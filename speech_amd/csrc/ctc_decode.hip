// ctc_decode.hip -- CTC decoding on the device.
//
// sa_ctc_beam_decode: the prefix beam search of /root/reference/speech/models/ctc_decoder.py:38-113, one wave per
// utterance, T sequential steps.  CTC.infer (ctc_model.py:55-60) calls it with beam_size = 1 on softmax outputs copied
// to the host; here it consumes logits (softmax of ctc_model.py:30-31 fused) or probabilities on the device.
// It reproduces the reference's arithmetic and ordering, not just its math:
//   * scores are float32; logsumexp (ctc_decoder.py:27-36) is  f32(a_max) + f32(log_d(sum_d exp_d(f32(a - a_max))))
//     with the sum taken in argument order in double (what NumPy-2 scalar promotion does to the reference's code);
//   * the per-step candidate set is the reference's dict: one "stay" entry per beam prefix plus one entry per
//     (prefix, non-blank symbol); an extension that equals another beam prefix is merged into that prefix's entry,
//     with the two updates applied in beam order (ctc_decoder.py:71 inner loop);
//   * ties sort by first-touch order -- vocab-major, beam-minor (ctc_decoder.py:65,71) -- under a stable descending
//     sort (ctc_decoder.py:107-110).
// Prefixes are nodes of a trie (parent, symbol) made canonical by a hash table, so "same prefix" is "same node id".
//
//
// Round 3 (latency: M-DEC beam 1 was 3.0 ms for (32, 498, 29), as long as the encoder forward it follows):
//   * the per-frame log-probabilities no longer sit on the T-step chain: ctc_logprob_kernel computes all B x T rows in
//     parallel (one wave per row, the SAME arithmetic and reduction order as before, so the values are bit-identical)
//     into the workspace, and a decode step just reads its row (requested a step ahead);
//   * beam_size = 1 (what CTC.infer asks for, ctc_model.py:59) has its own kernel: with one beam entry the dict of a
//     step is "stay" + one extension per symbol, nothing ever merges, the survivor is the arg-max under the reference's
//     tie order, and a prefix is just the labels appended so far -- no trie, no hash table, no LDS, the state in
//     registers; a log-sum-exp costs ONE double exp and one double log (the largest term's exp(0) = 1 and the -inf
//     terms' 0 are exact, so the reference's three-term sums are reproduced bit for bit);
//   * the general kernel runs 4 waves per utterance: the W x S candidates of a step (1 .. 3 double log-sum-exps each)
//     are spread over 256 threads instead of walked four at a time by one wave; selection and the trie stay on wave 0.
//
// sa_ctc_greedy_decode: argmax per frame + CTC.max_decode collapse (ctc_model.py:62-70), one wave per utterance.
#include "common.h"

namespace {

#define NEG_INF_F (-__builtin_inff())

// ctc_decoder.py:27-36 on up to three arguments, in order.  Arguments are float32 values or -inf.
// A term equal to the maximum contributes exp(0.0) = 1.0 exactly and is not sent through exp (round 3: one double exp
// fewer per call on the T-step chain; the value is unchanged bit for bit -- the terms are still added in argument order).
__device__ __forceinline__ double ref_lse_term(float x, float m) {
    return (x == NEG_INF_F) ? 0.0 : (x == m ? 1.0 : exp((double)(x - m)));
}
__device__ __forceinline__ float ref_lse3(float a, float b, float c, int n) {
    float m = a;
    if (n > 1 && b > m) m = b;
    if (n > 2 && c > m) m = c;
    if (m == NEG_INF_F) return NEG_INF_F;
    double tot = 0.0;
    tot += ref_lse_term(a, m);
    if (n > 1) tot += ref_lse_term(b, m);
    if (n > 2) tot += ref_lse_term(c, m);
    return m + (float)log(tot);
}

// ctc_decoder.py:27-36 on two arguments: identical in value to ref_lse3(-inf, x, y, 3) and to ref_lse3(x, y, ., 2) --
// the term of the maximum is exp(0.0) = 1.0 exactly, a -inf term contributes 0.0 exactly, and the double sum of the two
// remaining terms does not depend on their order -- with one exp instead of two or three.
__device__ __forceinline__ float ref_lse2(float x, float y) {
    const float m = fmaxf(x, y), lo = fminf(x, y);
    if (m == NEG_INF_F) return NEG_INF_F;
    const double e = (lo == NEG_INF_F) ? 0.0 : exp((double)(lo - m));
    return m + (float)log(1.0 + e);
}

// log-probabilities of every frame, one wave per (utterance, frame) row: ctc_decoder.py:52 after ctc_model.py:30-31.
// grid: ceil(B * T_max / 4) blocks of 4 waves.  lp: (B, T_max, S).
__global__ __launch_bounds__(256) void ctc_logprob_kernel(const float* __restrict__ in, long st, long sb,
                                                          const int* __restrict__ in_lens, int S, int B, int T_max,
                                                          int is_logits, float* __restrict__ lp) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long)B * T_max) return;
    const int b = (int)(r / T_max), t = (int)(r - (long)b * T_max);
    if (t >= in_lens[b]) return;
    const float* row = in + (long)b * sb + (long)t * st;
    float* out = lp + r * S;
    if (is_logits) {
        float m = -3.0e38f;
        for (int s = lane; s < S; s += 64) m = fmaxf(m, row[s]);
        m = sa_wave_max(m);
        float z = 0.f;
        for (int s = lane; s < S; s += 64) z += expf(row[s] - m);
        z = sa_wave_sum(z);
        for (int s = lane; s < S; s += 64) out[s] = logf(expf(row[s] - m) / z);
    } else {
        for (int s = lane; s < S; s += 64) out[s] = logf(row[s]);  // log(0) = -inf, as np.log
    }
}

struct BeamArgs {
    const float* in;    // the log-probabilities (B, T_max, S) written by ctc_logprob_kernel
    long st, sb;
    const int* in_lens;
    int S, B, T_max, W, blank, is_logits;
    int* out_labels;   // (B, T_max)
    int* out_lens;     // (B)
    float* out_nll;    // (B) or null
    // per-utterance workspace
    int* node_parent;  // [B][max_nodes]
    int* node_sym;     // [B][max_nodes]
    unsigned long long* hkeys;  // [B][hsize]   (0 = empty)
    int* hvals;        // [B][hsize]
    int trie_in_lds;   // the trie and its hash table fit the workgroup's LDS (the common case): no global atomics, no
                       // dependent global loads on the T-step chain; else they live in the workspace arrays above
    int max_nodes, hsize;
};

__device__ __forceinline__ unsigned hash_u64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (unsigned)k;
}

// beam_size = 1 (CTC.infer): one wave per utterance, lane = symbol (S <= 64), state in registers.  Mirrors
// ctc_beam_kernel for W = 1 term by term: the blank lane carries the prefix's "stay" entry (p_b from the blank, p_nb
// from repeating the last symbol), every other lane the extension by its symbol (from p_b only when it repeats the last
// symbol); nothing can merge (the beam holds no other prefix); first-touch order of an entry = 2 s (its symbol's turn in
// the vocab-major loop), of the stay entry also 2 last + 1 (the repeat-merge) -- ctc_decoder.py:65-103.
__global__ __launch_bounds__(64) void ctc_beam1_kernel(BeamArgs A) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int S = A.S, T = A.in_lens[b];
    const float* lp = A.in + (long)b * A.sb;
    int* out = A.out_labels + (long)b * A.T_max;
    float pb = 0.0f, pnb = NEG_INF_F;
    int last = -1, len = 0;
    const bool live = lane < S;
    // (r6) the rows EIGHT steps ahead, in a register ring: one step ahead (rounds 3 - 5) made a step as long as a load's
    // round trip -- 1.46 us where its arithmetic is ~0.3 (CTC.infer at B = 1 spent a tenth of the call here)
    constexpr int kAhead = 8;
    float ring[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) ring[u] = T > 0 ? lp[(long)min(u, T - 1) * A.st + (live ? lane : 0)] : 0.f;
    for (int t0 = 0; t0 < T; t0 += kAhead) {
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int t = t0 + u;
        if (t >= T) break;   // wave-uniform
        const float p = live ? ring[u] : NEG_INF_F;
        // unconditional load at a clamped address, the RAW value into the ring (a load under a branch, or a select on its
        // result, makes every step wait for the load it has just issued); rows beyond T - 1 are never consumed
        ring[u] = lp[(long)min(t + kAhead, T - 1) * A.st + (live ? lane : 0)];
        float npb = NEG_INF_F, npnb = NEG_INF_F, score = NEG_INF_F, ord = 3.0e38f;
        // log-probability of the prefix's last symbol (`last` is wave-uniform: v_readlane, outside divergent flow)
        const float q = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), last >= 0 ? last : 0));
        // lse(pb + p, pnb + p) is the blank lane's new p_b AND every other lane's extension: once, outside the divergent flow
        // (the two branches each carried their own double-precision exp + log: three in series per step, now two)
        const float both = ref_lse2(pb + p, pnb + p);
        if (live) {
            ord = (float)(2 * lane);
            if (lane == A.blank) {
                npb = both;
                if (last >= 0) {
                    npnb = pnb + q;                       // ref_lse3(-inf, merge, ., 2) = merge exactly
                    ord = fminf(ord, (float)(2 * last + 1));
                }
                score = ref_lse2(npb, npnb);
            } else {
                npnb = (lane != last) ? both : pb + p;   // ctc_decoder.py:88-96
                score = npnb;                             // ref_lse3(-inf, npnb, ., 2) = npnb exactly
            }
        }
        // stable descending arg-max: score first, then the smallest first-touch order (unique per entry)
        const float wbest = sa_wave_max_dpp(score);
        const bool tie = live && score == wbest;
        const float word = -sa_wave_max_dpp(tie ? -ord : -3.0e38f);
        const unsigned long long win = __ballot(tie && ord == word);
        const int w = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(win | (1ULL << 63)));
        pb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, npb), w));
        pnb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, npnb), w));
        if (w != A.blank) {
            last = w;
            if (lane == 0 && len < A.T_max) out[len] = w;
            ++len;
        }
      }
    }
    if (lane == 0) {
        A.out_lens[b] = len < A.T_max ? len : A.T_max;
        if (A.out_nll) A.out_nll[b] = -ref_lse2(pb, pnb);
    }
}

__host__ __device__ inline size_t beam_lds_bytes_dev(int S, int W) { return (size_t)(S + 9 * W + 5 * W * S + 8) * sizeof(float); }

// 4 waves per utterance: the candidate phase on all 256 threads, everything else on wave 0
// The stable descending selection of the top W of a step's candidates (ctc_decoder.py:107-110) by ONE wave with the
// candidates in registers: lane l holds c = l, l + 64, ... (NPL of them), sorted once by (score descending, first-touch
// order ascending: the order is unique per live candidate, so this is a total order; void candidates sink to the end).
// A round then compares only the lanes' HEADS: two wave reductions (the best score; the smallest order among its
// holders) and a pop in the winner's lane -- nothing on the W-round chain goes through LDS (the rounds of the LDS form
// further down cost 1 us each: eight dependent LDS round trips and the winner's copies per round).  Same comparisons as
// that form, hence the same survivors in the same order.  Returns in lane r < nsel the candidate of rank r.
template <int NPL>
__device__ __forceinline__ int beam_select_regs(const float* c_score, const float* c_ord, const int* c_state, int ncand,
                                                int W, int lane, int& nsel) {
    float sc[NPL], od[NPL];
    int ix[NPL];
    int nlive = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int c = lane + 64 * j;
        const bool live = c < ncand && c_state[c < ncand ? c : 0] != 0;
        sc[j] = live ? c_score[c] : NEG_INF_F;
        od[j] = live ? c_ord[c] : 3.0e38f;
        ix[j] = c;
        nlive += live ? 1 : 0;
    }
    nlive = (int)sa_wave_sum((float)nlive);
    nsel = min(W, nlive);
    auto cmpx = [&](int a, int b) {  // after: slot a holds the better of the two
        const bool swap = sc[b] > sc[a] || (sc[b] == sc[a] && od[b] < od[a]);
        const float s0 = swap ? sc[b] : sc[a], s1 = swap ? sc[a] : sc[b];
        const float o0 = swap ? od[b] : od[a], o1 = swap ? od[a] : od[b];
        const int i0 = swap ? ix[b] : ix[a], i1 = swap ? ix[a] : ix[b];
        sc[a] = s0; sc[b] = s1; od[a] = o0; od[b] = o1; ix[a] = i0; ix[b] = i1;
    };
    if constexpr (NPL == 4) {
        cmpx(0, 1); cmpx(2, 3); cmpx(0, 2); cmpx(1, 3); cmpx(1, 2);
    } else {  // 19-comparator network for 8
        cmpx(0, 1); cmpx(2, 3); cmpx(4, 5); cmpx(6, 7);
        cmpx(0, 2); cmpx(1, 3); cmpx(4, 6); cmpx(5, 7);
        cmpx(1, 2); cmpx(5, 6); cmpx(0, 4); cmpx(3, 7);
        cmpx(1, 5); cmpx(2, 6);
        cmpx(1, 4); cmpx(3, 6);
        cmpx(2, 4); cmpx(3, 5);
        cmpx(3, 4);
    }
    int mine = -1;
    for (int r = 0; r < nsel; ++r) {
        const float wbest = sa_wave_max_dpp(sc[0]);
        const bool tie = sc[0] == wbest && od[0] < 3.0e38f;
        bool win = tie;
        if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(tie)) != 1) {  // several heads hold the best score (rare: exact
            const float word = -sa_wave_max_dpp(tie ? -od[0] : -3.0e38f);   // float ties, or only -inf left): the order decides
            win = tie && od[0] == word;
        }
        const unsigned long long wmask = __builtin_amdgcn_ballot_w64(win);
        const int wl = __builtin_ctzll(wmask | (1ULL << 63));
        const int widx = __builtin_amdgcn_readlane(ix[0], wl);
        if (lane == r) mine = widx;
        if (win) {  // pop
#pragma unroll
            for (int j = 0; j + 1 < NPL; ++j) { sc[j] = sc[j + 1]; od[j] = od[j + 1]; ix[j] = ix[j + 1]; }
            sc[NPL - 1] = NEG_INF_F; od[NPL - 1] = 3.0e38f;
        }
    }
    return mine;
}

__global__ __launch_bounds__(256) void ctc_beam_kernel(BeamArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int S = A.S, W = A.W;
    const int ncand_max = W * S;
    float* lp = reinterpret_cast<float*>(smem_raw);             // [S]
    float* b_pb = lp + S;                                       // [W] beam: p_blank
    float* b_pnb = b_pb + W;                                    // [W] beam: p_non_blank
    int* b_node = reinterpret_cast<int*>(b_pnb + W);            // [W]
    int* b_last = b_node + W;                                   // [W] last symbol, -1 for the empty prefix
    int* b_ip = b_last + W;                                     // [W] beam index of the parent prefix, -1 if absent
    float* n_pb = reinterpret_cast<float*>(b_ip + W);           // [W] next beam
    float* n_pnb = n_pb + W;
    int* n_node = reinterpret_cast<int*>(n_pnb + W);
    int* n_last = n_node + W;
    float* c_pb = reinterpret_cast<float*>(n_last + W);         // [W*S] candidates
    float* c_pnb = c_pb + ncand_max;
    float* c_score = c_pnb + ncand_max;
    float* c_ord = c_score + ncand_max;                         // first-touch order (exact small integers in fp32)
    int* c_state = reinterpret_cast<int*>(c_ord + ncand_max);   // 0 = void, 1 = live, 2 = taken
    int* node_count = c_state + ncand_max;                      // [1]
    int* s_nsel = node_count + 1;                               // [1] survivors of this step (written by wave 0)

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = tid < 64;
    const int T = A.in_lens[b];
    int* node_parent = A.node_parent + (long)b * A.max_nodes;
    int* node_sym = A.node_sym + (long)b * A.max_nodes;
    unsigned long long* hkeys = A.hkeys + (long)b * A.hsize;
    int* hvals = A.hvals + (long)b * A.hsize;
    if (A.trie_in_lds) {
        hkeys = reinterpret_cast<unsigned long long*>(smem_raw + ((beam_lds_bytes_dev(S, W) + 15) & ~(size_t)15));
        hvals = reinterpret_cast<int*>(hkeys + A.hsize);
        node_parent = hvals + A.hsize;
        node_sym = node_parent + A.max_nodes;
        for (int k = tid; k < A.hsize; k += 256) hkeys[k] = 0ULL;
    }
    const unsigned hmask = (unsigned)A.hsize - 1;

    if (tid == 0) {
        b_pb[0] = 0.0f; b_pnb[0] = NEG_INF_F; b_node[0] = 0; b_last[0] = -1;
        node_parent[0] = -1; node_sym[0] = -1;
        *node_count = 1;
    }
    int nb = 1;  // beam entries in use (block-uniform)
    __syncthreads();

    // a frame's log-probabilities (computed by ctc_logprob_kernel: ctc_decoder.py:52 after ctc_model.py:30-31) are
    // requested a step ahead: the global load is not on the step's chain (S <= 256: one value per thread)
    const float* rows = A.in + (long)b * A.sb;
    float lp_next = (tid < S && T > 0) ? rows[tid] : 0.f;
    for (int t = 0; t < T; ++t) {
        const float* row = rows + (long)t * A.st;
        if (S <= 256) {
            if (tid < S) lp[tid] = lp_next;
            if (tid < S && t + 1 < T) lp_next = row[A.st + tid];
        } else {
            for (int s = tid; s < S; s += 256) lp[s] = row[s];
        }
        // ---- where is each beam prefix's parent in the beam?
        if (tid < nb) {
            const int par = node_parent[b_node[lane]];
            int ip = -1;
            for (int k = 0; k < nb; ++k)
                if (b_node[k] == par) ip = k;
            b_ip[lane] = ip;
        }
        __syncthreads();

        // ---- candidates: slot (i, s); the blank slot of beam entry i carries its "stay" entry
        const int ncand = nb * S;
        for (int c = tid; c < ncand; c += 256) {
            const int i = c / S, s = c - i * S;
            const float pb = b_pb[i], pnb = b_pnb[i];
            const int last = b_last[i];
            float npb, npnb, ord;
            int state = 1;
            if (s == A.blank) {
                const float p = lp[s];
                npb = ref_lse3(NEG_INF_F, pb + p, pnb + p, 3);
                npnb = NEG_INF_F;
                ord = (float)((s * W + i) * 2);
                if (last >= 0) {  // non-empty prefix: repeat-merge and (if the parent is in the beam) parent extension
                    const float q = lp[last];
                    const int ip = b_ip[i];
                    float vpar_b = 0.f, vpar_nb = 0.f;
                    int par_n = 0;
                    if (ip >= 0) {
                        vpar_b = b_pb[ip] + q;
                        vpar_nb = b_pnb[ip] + q;
                        par_n = (last != b_last[ip]) ? 3 : 2;  // ctc_decoder.py:88-96
                    }
                    const float merge = pnb + q;
                    if (ip >= 0 && ip < i) {
                        npnb = ref_lse3(NEG_INF_F, vpar_b, vpar_nb, par_n);
                        npnb = ref_lse3(npnb, merge, 0.f, 2);
                    } else {
                        npnb = ref_lse3(NEG_INF_F, merge, 0.f, 2);
                        if (ip >= 0) npnb = ref_lse3(npnb, vpar_b, vpar_nb, par_n);
                    }
                    // first touch of this dict entry
                    float o2 = (float)((last * W + i) * 2 + 1);
                    if (ip >= 0) o2 = fminf(o2, (float)((last * W + ip) * 2));
                    ord = fminf(ord, o2);
                }
            } else {
                // does (prefix_i + s) equal another beam prefix?  then that prefix's stay entry owns the update
                bool merged = false;
                for (int k = 0; k < nb; ++k)
                    if (b_ip[k] == i && b_last[k] == s) merged = true;
                if (merged) state = 0;
                const float p = lp[s];
                npb = NEG_INF_F;
                npnb = (s != last) ? ref_lse3(NEG_INF_F, pb + p, pnb + p, 3) : ref_lse3(NEG_INF_F, pb + p, 0.f, 2);
                ord = (float)((s * W + i) * 2);
            }
            c_pb[c] = npb; c_pnb[c] = npnb;
            c_score[c] = ref_lse3(npb, npnb, 0.f, 2);
            c_ord[c] = ord;
            c_state[c] = state;
        }
        __syncthreads();

        // ---- stable descending selection of the top W (ctc_decoder.py:107-110): wave 0
        if (wave0 && ncand <= 512) {
            int nsel;
            const int mine = ncand <= 256 ? beam_select_regs<4>(c_score, c_ord, c_state, ncand, W, lane, nsel)
                                          : beam_select_regs<8>(c_score, c_ord, c_state, ncand, W, lane, nsel);
            if (lane == 0) *s_nsel = nsel;
            if (lane < nsel) {  // the survivors' entries: all ranks at once
                const int bidx = mine;
                c_state[bidx] = 2;
                const int i = bidx / S, sy = bidx - i * S;
                n_pb[lane] = c_pb[bidx]; n_pnb[lane] = c_pnb[bidx];
                if (sy == A.blank) {
                    n_node[lane] = b_node[i]; n_last[lane] = b_last[i];
                } else {
                    n_node[lane] = -(i * S + sy) - 1;  // resolved to a trie node below
                    n_last[lane] = sy;
                }
            }
        } else if (wave0) {
        int nlive = 0;
        for (int c = lane; c < ncand; c += 64) nlive += c_state[c] != 0;
        nlive = (int)sa_wave_sum((float)nlive);
        const int nsel = min(W, nlive);
        if (lane == 0) *s_nsel = nsel;
        for (int r = 0; r < nsel; ++r) {
            float best = NEG_INF_F, bord = 3.0e38f;
            int bidx = -1;
            for (int c = lane; c < ncand; c += 64) {
                if (c_state[c] != 1) continue;
                const float sc = c_score[c], od = c_ord[c];
                if (bidx < 0 || sc > best || (sc == best && od < bord)) { best = sc; bord = od; bidx = c; }
            }
            // wave argmax: score first, then the smallest first-touch order (unique per live candidate)
            const bool have = bidx >= 0;
            const float wbest = sa_wave_max_dpp(have ? best : NEG_INF_F);
            const bool tie = have && best == wbest;
            const float word = -sa_wave_max_dpp(tie ? -bord : -3.0e38f);
            if (tie && bord == word) {
                c_state[bidx] = 2;
                const int i = bidx / S, s = bidx - i * S;
                n_pb[r] = c_pb[bidx]; n_pnb[r] = c_pnb[bidx];
                if (s == A.blank) {
                    n_node[r] = b_node[i]; n_last[r] = b_last[i];
                } else {
                    n_node[r] = -(i * S + s) - 1;  // resolved to a trie node below
                    n_last[r] = s;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are visible to its next reads
            __builtin_amdgcn_wave_barrier();
        }
        }
        __syncthreads();
        const int nsel = *s_nsel;
        // ---- canonical trie nodes for the surviving extensions (hash table keyed by (parent node, symbol))
        if (tid < nsel && n_node[tid] < 0) {
            const int code = -(n_node[lane] + 1);
            const int i = code / S, s = code - i * S;
            const int par = b_node[i];
            const unsigned long long key = ((unsigned long long)(unsigned)(par + 1) << 32) | (unsigned)(s + 1);
            unsigned slot = hash_u64(key) & hmask;
            int id = -1;
            for (int probe = 0; probe <= (int)hmask; ++probe) {
                const unsigned long long old = atomicCAS(&hkeys[slot], 0ULL, key);
                if (old == 0ULL) {  // new prefix
                    id = atomicAdd(node_count, 1);
                    if (id < A.max_nodes) { node_parent[id] = par; node_sym[id] = s; }
                    hvals[slot] = id;
                    break;
                }
                if (old == key) { id = hvals[slot]; break; }
                slot = (slot + 1) & hmask;
            }
            n_node[lane] = id;
        }
        __syncthreads();
        if (tid < nsel) {
            b_pb[lane] = n_pb[lane]; b_pnb[lane] = n_pnb[lane];
            b_node[lane] = n_node[lane]; b_last[lane] = n_last[lane];
        }
        nb = nsel;
        __syncthreads();
    }

    // ---- best prefix: walk the trie back to the root
    if (tid == 0) {
        int node = b_node[0];
        int len = 0;
        for (int n = node; n > 0 && len < A.T_max; n = node_parent[n]) ++len;
        int* out = A.out_labels + (long)b * A.T_max;
        int pos = len;
        for (int n = node; n > 0 && pos > 0; n = node_parent[n]) out[--pos] = node_sym[n];
        A.out_lens[b] = len;
        if (A.out_nll) A.out_nll[b] = -ref_lse3(b_pb[0], b_pnb[0], 0.f, 2);
    }
}

// one wave per utterance: argmax per frame (first maximal index), keep p iff p != blank and p != previous frame's p
__global__ __launch_bounds__(64) void ctc_greedy_kernel(const float* __restrict__ in, long st, long sb,
                                                        const int* __restrict__ in_lens, int S, int T_max, int blank,
                                                        int* __restrict__ out_labels, int* __restrict__ out_lens) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int T = in_lens[b];
    int* out = out_labels + (long)b * T_max;
    int base = 0, carry = -1;  // carry: label of the frame before this chunk (-1: none)
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        int arg = -1;
        if (t < T) {
            const float* row = in + (long)b * sb + (long)t * st;
            float best = row[0];
            arg = 0;
            for (int s = 1; s < S; ++s) {
                const float v = row[s];
                if (v > best) { best = v; arg = s; }
            }
        }
        int prev = __shfl_up(arg, 1, 64);
        if (lane == 0) prev = carry;
        const bool keep = t < T && arg != blank && arg != prev;
        const unsigned long long mask = __ballot(keep);
        if (keep) out[base + __popcll(mask & ((1ULL << lane) - 1ULL))] = arg;
        base += __popcll(mask);
        carry = __shfl(arg, 63, 64);
    }
    if (lane == 0) out_lens[b] = base;
}

size_t beam_lds_bytes(int S, int W) { return beam_lds_bytes_dev(S, W); }

}  // namespace

static void beam_ws_layout(int max_T, int S, int B, int W, int* max_nodes, int* hsize, size_t* o_par, size_t* o_sym,
                           size_t* o_keys, size_t* o_vals, size_t* o_lp, size_t* total) {
    *max_nodes = max_T * W + 2;
    int h = 64;
    while (h < 2 * (*max_nodes)) h <<= 1;
    *hsize = h;
    size_t o = 0;
    *o_keys = o; o += sa_align_up((size_t)B * h * sizeof(unsigned long long), 256);  // first: one memset clears it
    *o_vals = o; o += sa_align_up((size_t)B * h * sizeof(int), 256);
    *o_par = o;  o += sa_align_up((size_t)B * (*max_nodes) * sizeof(int), 256);
    *o_sym = o;  o += sa_align_up((size_t)B * (*max_nodes) * sizeof(int), 256);
    *o_lp = o;   o += sa_align_up((size_t)B * max_T * S * sizeof(float), 256);  // log-probabilities of every frame
    *total = o;
}

extern "C" size_t sa_ctc_beam_workspace_bytes(int max_T, int alphabet_size, int minibatch, int beam_size) {
    if (max_T <= 0 || alphabet_size <= 0 || minibatch <= 0 || beam_size <= 0) return 0;
    int mn, hs;
    size_t a, b, c, d, e, total;
    beam_ws_layout(max_T, alphabet_size, minibatch, beam_size, &mn, &hs, &a, &b, &c, &d, &e, &total);
    return total;
}

extern "C" ctcStatus_t sa_ctc_beam_decode(const float* in, long stride_t, long stride_b, const int* d_input_lengths,
                                          int alphabet_size, int minibatch, int max_T, int beam_size,
                                          int blank_label, int input_is_logits, int* d_out_labels, int* d_out_lens,
                                          float* d_out_nll, void* workspace, size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    if (!in || !d_input_lengths || !d_out_labels || !d_out_lens || !workspace) return CTC_STATUS_INVALID_VALUE;
    if (alphabet_size <= 0 || minibatch <= 0 || max_T <= 0 || beam_size <= 0 || beam_size > 64 || blank_label < 0 ||
        blank_label >= alphabet_size)
        return CTC_STATUS_INVALID_VALUE;
    const size_t lds = beam_lds_bytes(alphabet_size, beam_size);
    if (lds > 150 * 1024) return CTC_STATUS_INVALID_VALUE;  // beam_size * alphabet too large for one workgroup
    if ((long)alphabet_size * beam_size * 2 + 2 * beam_size >= (1 << 24)) return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    BeamArgs A;
    size_t o_par, o_sym, o_keys, o_vals, o_lp, total;
    beam_ws_layout(max_T, alphabet_size, minibatch, beam_size, &A.max_nodes, &A.hsize, &o_par, &o_sym, &o_keys, &o_vals,
                   &o_lp, &total);
    if (workspace_bytes < total) return CTC_STATUS_INVALID_VALUE;
    char* ws = (char*)workspace;
    // every frame's log-probabilities, all rows in parallel: off the decode kernels' T-step chain
    float* lpbuf = (float*)(ws + o_lp);
    hipLaunchKernelGGL(ctc_logprob_kernel, dim3((unsigned)(((long)minibatch * max_T + 3) / 4)), dim3(256), 0, stream, in,
                       stride_t, stride_b, d_input_lengths, alphabet_size, minibatch, max_T, input_is_logits, lpbuf);
    A.in = lpbuf; A.st = alphabet_size; A.sb = (long)max_T * alphabet_size; A.in_lens = d_input_lengths;
    A.S = alphabet_size; A.B = minibatch; A.T_max = max_T; A.W = beam_size; A.blank = blank_label;
    A.is_logits = input_is_logits;
    A.out_labels = d_out_labels; A.out_lens = d_out_lens; A.out_nll = d_out_nll;
    A.node_parent = (int*)(ws + o_par); A.node_sym = (int*)(ws + o_sym);
    A.hkeys = (unsigned long long*)(ws + o_keys); A.hvals = (int*)(ws + o_vals);
    if (beam_size == 1 && alphabet_size <= 64) {  // CTC.infer: no trie, no hash table
        hipLaunchKernelGGL(ctc_beam1_kernel, dim3(minibatch), dim3(64), 0, stream, A);
        SA_CHECK_LAUNCH();
        return CTC_STATUS_SUCCESS;
    }
    // the prefix trie + its hash table in LDS when they fit (T' = 498, beam 8: 128 KB): a survivor's node is found / made
    // with LDS atomics and a prefix's parent read from LDS, instead of global atomics and dependent global loads per step
    const size_t trie = (size_t)A.hsize * 12 + (size_t)A.max_nodes * 8;
    const size_t lds_all = ((lds + 15) & ~(size_t)15) + trie;
    A.trie_in_lds = lds_all <= 150 * 1024 ? 1 : 0;
    size_t lds_req = A.trie_in_lds ? lds_all : lds;
    if (A.trie_in_lds && lds_req > 48 * 1024 &&
        hipFuncSetAttribute((const void*)ctc_beam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_req) !=
            hipSuccess) {
        // a device (or a carve-out) that does not grant the LDS-resident trie: the global-memory trie is the same search
        (void)hipGetLastError();
        A.trie_in_lds = 0;
        lds_req = lds;
    }
    if (!A.trie_in_lds &&
        hipMemsetAsync(A.hkeys, 0, (size_t)minibatch * A.hsize * sizeof(unsigned long long), stream) != hipSuccess)
        return CTC_STATUS_MEMOPS_FAILED;
    if (!A.trie_in_lds && lds_req > 48 * 1024 &&
        hipFuncSetAttribute((const void*)ctc_beam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_req) !=
            hipSuccess)
        return CTC_STATUS_EXECUTION_FAILED;
    hipLaunchKernelGGL(ctc_beam_kernel, dim3(minibatch), dim3(256), lds_req, stream, A);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" ctcStatus_t sa_ctc_greedy_decode(const float* in, long stride_t, long stride_b, const int* d_input_lengths,
                                            int alphabet_size, int minibatch, int max_T, int blank_label,
                                            int* d_out_labels, int* d_out_lens, void* stream_) {
    SA_CLEAR_ERR();
    if (!in || !d_input_lengths || !d_out_labels || !d_out_lens || alphabet_size <= 0 || minibatch <= 0 || max_T <= 0)
        return CTC_STATUS_INVALID_VALUE;
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3(minibatch), dim3(64), 0, (hipStream_t)stream_, in, stride_t, stride_b,
                       d_input_lengths, alphabet_size, max_T, blank_label, d_out_labels, d_out_lens);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
