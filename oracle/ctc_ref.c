/*
 * oracle/ctc_ref.c -- CPU restatement of the CTC loss (forward alpha, backward beta, gradient
 * w.r.t. un-normalised logits).  TEST INFRASTRUCTURE ONLY: imported by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the product path.
 *
 * What it restates.  The reference calls an out-of-tree native extension at
 *   /root/reference/speech/models/ctc_model.py:9      import functions.ctc as ctc
 *   /root/reference/speech/models/ctc_model.py:38-39  loss_fn = ctc.CTCLoss(); loss_fn(out, y, x_lens, y_lens)
 * which is github.com/awni/warp-ctc, cloned UNPINNED at build time
 * (/root/reference/Makefile:4-7) and absent from /root/reference.  Its published algorithm
 * (Graves et al. 2006; the forward-variable formulation the reference's decoder docstring
 * cites, ctc_decoder.py:11-13; restated in SURVEY.md Appendix A) is what this file implements,
 * constrained by the call-site facts: logits are un-normalised and batch-first (ctc_model.py:29-32,36),
 * blank is the LAST class (ctc_model.py:18), labels are a flat int32 vector (ctc_model.py:47-48),
 * every act_len equals the padded length (ctc_model.py:43-45).
 *
 * PARITY UNPINNED: no file under /root/reference holds a golden loss or gradient for this path
 * (tests/ctc_test.py:26 only checks that the call returns).  The restatement is instead pinned by
 * (i) brute-force path enumeration on tiny lattices, (ii) fp64 central finite differences and
 * (iii) torch.nn.functional.ctc_loss on CPU (tests/test_oracle_ctc.py).
 *
 * Conventions (all explicit parameters):
 *   acts[t*stride_t + b*stride_b + k]   k in [0,K), K = |V|+1
 *   blank index `blank`; per-utterance cost = -log p(l|x); no batch reduction here.
 *   infeasible alignment (T < L + repeats) or p underflowing to 0: cost = +inf, gradient = 0.
 *   rows t >= act_lens[b] get zero gradient.
 *
 * ctc_ref_f64: everything in double (the checker).
 * ctc_ref_f32: float log-space arithmetic, OpenMP over utterances -- mirrors the threading model
 *              of warp-ctc's CPU path (one task per utterance); used as the timed CPU baseline.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEG_INF (-INFINITY)

static inline double lse2d(double a, double b) {
    if (a == NEG_INF) return b;
    if (b == NEG_INF) return a;
    double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}
static inline double lse3d(double a, double b, double c) { return lse2d(lse2d(a, b), c); }

static inline float lse2f(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    float m = a > b ? a : b;
    return m + logf(expf(a - m) + expf(b - m));
}

/* One utterance, double precision.  acts/grads point at (t=0, b, k=0). */
static double ctc_one_f64(const float* acts, double* grads, long stride_t, const int* lab, int L, int T,
                          int K, int blank) {
    const int S = 2 * L + 1;
    double* ly = (double*)malloc(sizeof(double) * (size_t)T * K);
    double* alpha = (double*)malloc(sizeof(double) * (size_t)T * S);
    double* beta = (double*)malloc(sizeof(double) * (size_t)T * S);
    int* ext = (int*)malloc(sizeof(int) * S);
    for (int s = 0; s < S; ++s) ext[s] = (s & 1) ? lab[s >> 1] : blank;

    /* log-softmax, max-shifted (Appendix A: ly = log y) */
    for (int t = 0; t < T; ++t) {
        const float* a = acts + (long)t * stride_t;
        double m = a[0];
        for (int k = 1; k < K; ++k) m = a[k] > m ? a[k] : m;
        double z = 0;
        for (int k = 0; k < K; ++k) z += exp((double)a[k] - m);
        double lz = m + log(z);
        for (int k = 0; k < K; ++k) ly[(size_t)t * K + k] = (double)a[k] - lz;
    }
    for (size_t i = 0; i < (size_t)T * S; ++i) alpha[i] = beta[i] = NEG_INF;

    double logp = NEG_INF;
    if (T > 0) {
        alpha[0] = ly[blank];
        if (S > 1) alpha[1] = ly[ext[1]];
        for (int t = 1; t < T; ++t) {
            for (int s = 0; s < S; ++s) {
                double v = alpha[(size_t)(t - 1) * S + s];
                if (s >= 1) v = lse2d(v, alpha[(size_t)(t - 1) * S + s - 1]);
                if (s >= 2 && ext[s] != blank && ext[s] != ext[s - 2])
                    v = lse2d(v, alpha[(size_t)(t - 1) * S + s - 2]);
                alpha[(size_t)t * S + s] = (v == NEG_INF) ? NEG_INF : v + ly[(size_t)t * K + ext[s]];
            }
        }
        logp = alpha[(size_t)(T - 1) * S + S - 1];
        if (S > 1) logp = lse2d(logp, alpha[(size_t)(T - 1) * S + S - 2]);

        /* beta includes the emission at t (Appendix A convention) */
        beta[(size_t)(T - 1) * S + S - 1] = ly[(size_t)(T - 1) * K + blank];
        if (S > 1) beta[(size_t)(T - 1) * S + S - 2] = ly[(size_t)(T - 1) * K + ext[S - 2]];
        for (int t = T - 2; t >= 0; --t) {
            for (int s = 0; s < S; ++s) {
                double v = beta[(size_t)(t + 1) * S + s];
                if (s + 1 < S) v = lse2d(v, beta[(size_t)(t + 1) * S + s + 1]);
                if (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != ext[s])
                    v = lse2d(v, beta[(size_t)(t + 1) * S + s + 2]);
                beta[(size_t)t * S + s] = (v == NEG_INF) ? NEG_INF : v + ly[(size_t)t * K + ext[s]];
            }
        }
    }

    if (grads) {
        double* acc = (double*)malloc(sizeof(double) * K);
        for (int t = 0; t < T; ++t) {
            double* g = grads + (long)t * stride_t;
            if (logp == NEG_INF) {
                for (int k = 0; k < K; ++k) g[k] = 0.0;
                continue;
            }
            for (int k = 0; k < K; ++k) acc[k] = NEG_INF;
            for (int s = 0; s < S; ++s)
                acc[ext[s]] = lse2d(acc[ext[s]], alpha[(size_t)t * S + s] + beta[(size_t)t * S + s]);
            for (int k = 0; k < K; ++k) {
                double l = ly[(size_t)t * K + k];
                double y = exp(l);
                double occ = (acc[k] == NEG_INF) ? 0.0 : exp(acc[k] - l - logp);
                g[k] = y - occ;
            }
        }
        free(acc);
    }
    free(ly); free(alpha); free(beta); free(ext);
    return (logp == NEG_INF) ? INFINITY : -logp;
}

/* costs[B] double; grads (same strides as acts, double) may be NULL.  labels flat. Returns 0 / nonzero on bad args. */
int ctc_ref_f64(const float* acts, long stride_t, long stride_b, const int* labels, const int* label_lens,
                const int* act_lens, int K, int B, int blank, int maxT, double* costs, double* grads) {
    if (!acts || !labels || !label_lens || !act_lens || !costs || K <= 0 || B <= 0 || blank < 0 || blank >= K)
        return 2;
    long off = 0;
    for (int b = 0; b < B; ++b) {
        int T = act_lens[b], L = label_lens[b];
        if (T < 0 || T > maxT || L < 0) return 2;
        for (int i = 0; i < L; ++i)
            if (labels[off + i] < 0 || labels[off + i] >= K || labels[off + i] == blank) return 2;
        double* g = grads ? grads + (long)b * stride_b : NULL;
        costs[b] = ctc_one_f64(acts + (long)b * stride_b, g, stride_t, labels + off, L, T, K, blank);
        if (g)
            for (int t = T; t < maxT; ++t)
                for (int k = 0; k < K; ++k) g[(long)t * stride_t + k] = 0.0;
        off += L;
    }
    return 0;
}

/* One utterance, float log-space (the arithmetic class of warp-ctc's CPU path). */
static float ctc_one_f32(const float* acts, float* grads, long stride_t, const int* lab, int L, int T, int K,
                         int blank) {
    const int S = 2 * L + 1;
    float* ly = (float*)malloc(sizeof(float) * (size_t)T * K);
    float* alpha = (float*)malloc(sizeof(float) * (size_t)T * S);
    float* bet = (float*)malloc(sizeof(float) * 2 * (size_t)S);
    int* ext = (int*)malloc(sizeof(int) * S);
    for (int s = 0; s < S; ++s) ext[s] = (s & 1) ? lab[s >> 1] : blank;
    for (int t = 0; t < T; ++t) {
        const float* a = acts + (long)t * stride_t;
        float m = a[0];
        for (int k = 1; k < K; ++k) m = a[k] > m ? a[k] : m;
        float z = 0;
        for (int k = 0; k < K; ++k) z += expf(a[k] - m);
        float lz = m + logf(z);
        for (int k = 0; k < K; ++k) ly[(size_t)t * K + k] = a[k] - lz;
    }
    float logp = -INFINITY;
    if (T > 0) {
        for (int s = 0; s < S; ++s) alpha[s] = -INFINITY;
        alpha[0] = ly[blank];
        if (S > 1) alpha[1] = ly[ext[1]];
        for (int t = 1; t < T; ++t) {
            const float* ap = alpha + (size_t)(t - 1) * S;
            float* an = alpha + (size_t)t * S;
            const float* l = ly + (size_t)t * K;
            /* reachability window (Appendix A): s <= 2t+1 and s >= S - 2(T-t) */
            int lo = S - 2 * (T - t); if (lo < 0) lo = 0;
            int hi = 2 * t + 1; if (hi > S - 1) hi = S - 1;
            for (int s = 0; s < lo; ++s) an[s] = -INFINITY;
            for (int s = hi + 1; s < S; ++s) an[s] = -INFINITY;
            for (int s = lo; s <= hi; ++s) {
                float v = ap[s];
                if (s >= 1) v = lse2f(v, ap[s - 1]);
                if (s >= 2 && (s & 1) && ext[s] != ext[s - 2]) v = lse2f(v, ap[s - 2]);
                an[s] = (v == -INFINITY) ? v : v + l[ext[s]];
            }
        }
        logp = alpha[(size_t)(T - 1) * S + S - 1];
        if (S > 1) logp = lse2f(logp, alpha[(size_t)(T - 1) * S + S - 2]);
    }
    if (grads) {
        float* acc = (float*)malloc(sizeof(float) * K);
        if (logp == -INFINITY) {
            for (int t = 0; t < T; ++t)
                for (int k = 0; k < K; ++k) grads[(long)t * stride_t + k] = 0.f;
        } else {
            float* bn = bet;       /* beta[t+1] */
            float* bc = bet + S;   /* beta[t]   */
            for (int t = T - 1; t >= 0; --t) {
                const float* l = ly + (size_t)t * K;
                if (t == T - 1) {
                    for (int s = 0; s < S; ++s) bc[s] = -INFINITY;
                    bc[S - 1] = l[blank];
                    if (S > 1) bc[S - 2] = l[ext[S - 2]];
                } else {
                    for (int s = 0; s < S; ++s) {
                        float v = bn[s];
                        if (s + 1 < S) v = lse2f(v, bn[s + 1]);
                        if (s + 2 < S && (s & 1) && ext[s + 2] != ext[s]) v = lse2f(v, bn[s + 2]);
                        bc[s] = (v == -INFINITY) ? v : v + l[ext[s]];
                    }
                }
                for (int k = 0; k < K; ++k) acc[k] = -INFINITY;
                const float* at = alpha + (size_t)t * S;
                for (int s = 0; s < S; ++s) acc[ext[s]] = lse2f(acc[ext[s]], at[s] + bc[s]);
                float* g = grads + (long)t * stride_t;
                for (int k = 0; k < K; ++k) {
                    float y = expf(l[k]);
                    float occ = (acc[k] == -INFINITY) ? 0.f : expf(acc[k] - l[k] - logp);
                    g[k] = y - occ;
                }
                float* tmp = bn; bn = bc; bc = tmp;
            }
        }
        free(acc);
    }
    free(ly); free(alpha); free(bet); free(ext);
    return (logp == -INFINITY) ? INFINITY : -logp;
}

int ctc_ref_f32(const float* acts, long stride_t, long stride_b, const int* labels, const int* label_lens,
                const int* act_lens, int K, int B, int blank, int maxT, float* costs, float* grads,
                int num_threads) {
    if (!acts || !labels || !label_lens || !act_lens || !costs || K <= 0 || B <= 0 || blank < 0 || blank >= K)
        return 2;
    long* offs = (long*)malloc(sizeof(long) * (B + 1));
    offs[0] = 0;
    for (int b = 0; b < B; ++b) offs[b + 1] = offs[b] + label_lens[b];
    int bad = 0;
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad)
#endif
    for (int b = 0; b < B; ++b) {
        int T = act_lens[b], L = label_lens[b];
        if (T < 0 || T > maxT || L < 0) { bad |= 1; continue; }
        float* g = grads ? grads + (long)b * stride_b : NULL;
        costs[b] = ctc_one_f32(acts + (long)b * stride_b, g, stride_t, labels + offs[b], L, T, K, blank);
        if (g)
            for (int t = T; t < maxT; ++t)
                for (int k = 0; k < K; ++k) g[(long)t * stride_t + k] = 0.f;
    }
    free(offs);
    return bad ? 2 : 0;
}

int ctc_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
