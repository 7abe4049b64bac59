/* oracle/transducer_ref.c -- CPU restatement of the RNN-Transducer loss (TEST INFRASTRUCTURE ONLY: tests/,
 * __graft_entry__.smoke(), tools' cpu legs; the product path never links or calls this).
 *
 * PARITY UNPINNED.  The reference takes this loss from an un-vendored, un-pinned dependency:
 *   /root/reference/Makefile:11          git clone github.com/awni/transducer (no commit / tag)
 *   /root/reference/speech/models/transducer_model.py:10-11   import transducer.decoders / transducer.functions.transducer
 *   /root/reference/speech/models/transducer_model.py:50-51   loss_fn = transducer.TransducerLoss(); loss_fn(out, y, x_lens, y_lens)
 * and no test under /root/reference/tests imports the Transducer model, so there is no golden vector.  What follows is
 * the published algorithm (A. Graves, "Sequence Transduction with Recurrent Neural Networks", 2012, eqs. 16-20), fixed
 * by the call-site facts: `out` is a LOG-softmax lattice (B, T, U+1, V+1) (transducer_model.py:76-77), blank = V (the
 * last class, :32 and :98), labels flat int32, x_lens all equal to the padded T' (:81-83).  It is pinned here by
 * brute-force path enumeration and fp64 finite differences (tests/test_oracle_transducer.py).
 *
 *   alpha[0,0] = 0;  alpha[t,u] = lse(alpha[t-1,u] + lp[t-1,u,blank], alpha[t,u-1] + lp[t,u-1,y[u-1]])
 *   log p = alpha[T-1,U] + lp[T-1,U,blank]
 *   beta[T-1,U] = lp[T-1,U,blank];  beta[t,u] = lse(beta[t+1,u] + lp[t,u,blank], beta[t,u+1] + lp[t,u,y[u]])
 *   cost = -log p;  d cost / d lp[t,u,blank] = -exp(alpha[t,u] + lp[t,u,blank] + beta[t+1,u] - log p)   (t < T-1)
 *                                              -exp(alpha[T-1,U] + lp[T-1,U,blank] - log p) = -1          (t = T-1, u = U)
 *                   d cost / d lp[t,u,y[u]]  = -exp(alpha[t,u] + lp[t,u,y[u]] + beta[t,u+1] - log p)      (u < U)
 * every other entry of the gradient is 0 (the lattice entries are treated as free variables, as a loss layered on
 * torch's log_softmax must).
 */
#include <math.h>
#include <stdlib.h>

static double lse2(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

/* lp: float32 (B, T_max, U1_max, K) contiguous.  labels flat.  grads: float64, same shape as lp, or NULL.
 * returns 0, or -1 on invalid arguments. */
int transducer_ref_f64(const float* lp, int B, int T_max, int U1_max, int K, const int* flat_labels,
                       const int* label_lens, const int* input_lens, int blank, double* costs, double* grads) {
    if (!lp || !flat_labels || !label_lens || !input_lens || !costs || B <= 0 || T_max <= 0 || U1_max <= 0 || K <= 0 ||
        blank < 0 || blank >= K)
        return -1;
    long* loff = (long*)malloc(sizeof(long) * (size_t)(B + 1));
    loff[0] = 0;
    for (int b = 0; b < B; ++b) {
        if (label_lens[b] < 0 || label_lens[b] + 1 > U1_max || input_lens[b] < 1 || input_lens[b] > T_max) {
            free(loff);
            return -1;
        }
        loff[b + 1] = loff[b] + label_lens[b];
    }
    const long cell = K, row = (long)U1_max * K, utt = (long)T_max * row;
    if (grads)
        for (long i = 0; i < (long)B * utt; ++i) grads[i] = 0.0;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        const int T = input_lens[b], U = label_lens[b], U1 = U + 1;
        const int* y = flat_labels + loff[b];
        const float* L = lp + (long)b * utt;
        double* al = (double*)malloc(sizeof(double) * (size_t)T * U1);
        double* be = (double*)malloc(sizeof(double) * (size_t)T * U1);
        for (int t = 0; t < T; ++t)
            for (int u = 0; u < U1; ++u) {
                double v;
                if (t == 0 && u == 0) v = 0.0;
                else {
                    const double a = t > 0 ? al[(t - 1) * U1 + u] + L[(t - 1) * row + u * cell + blank] : -INFINITY;
                    const double c = u > 0 ? al[t * U1 + u - 1] + L[t * row + (u - 1) * cell + y[u - 1]] : -INFINITY;
                    v = lse2(a, c);
                }
                al[t * U1 + u] = v;
            }
        const double logp = al[(T - 1) * U1 + U] + L[(T - 1) * row + U * cell + blank];
        costs[b] = -logp;
        if (grads) {
            for (int t = T - 1; t >= 0; --t)
                for (int u = U; u >= 0; --u) {
                    double v;
                    if (t == T - 1 && u == U) v = L[t * row + u * cell + blank];
                    else {
                        const double a = t < T - 1 ? be[(t + 1) * U1 + u] + L[t * row + u * cell + blank] : -INFINITY;
                        const double c = u < U ? be[t * U1 + u + 1] + L[t * row + u * cell + y[u]] : -INFINITY;
                        v = lse2(a, c);
                    }
                    be[t * U1 + u] = v;
                }
            double* G = grads + (long)b * utt;
            if (logp > -INFINITY)
                for (int t = 0; t < T; ++t)
                    for (int u = 0; u < U1; ++u) {
                        const double a = al[t * U1 + u];
                        if (t < T - 1)
                            G[t * row + u * cell + blank] = -exp(a + L[t * row + u * cell + blank] + be[(t + 1) * U1 + u] - logp);
                        else if (u == U)
                            G[t * row + u * cell + blank] = -exp(a + L[t * row + u * cell + blank] - logp);
                        if (u < U)
                            G[t * row + u * cell + y[u]] += -exp(a + L[t * row + u * cell + y[u]] + be[t * U1 + u + 1] - logp);
                    }
        }
        free(al);
        free(be);
    }
    free(loff);
    return 0;
}
