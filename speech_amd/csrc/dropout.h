// dropout.h -- the Bernoulli masks of the encoder's dropout, generated INSIDE the kernels that produce or consume the
// masked tensor (no mask tensor in memory, no extra pass over the activations).
//
// Reference: nn.Dropout(p) after every conv ReLU (/root/reference/speech/models/model.py:25-27) and
// nn.GRU(dropout=p) between the layers of the stack (model.py:35-39); every shipped config trains with p = 0.2 .. 0.5
// (examples/timit/ctc_config.json:20).  torch draws those masks from its own generator; which elements are dropped is
// not part of any contract, only the distribution is -- here element `idx` of masked tensor `stream` of a forward pass
// keyed by `seed` is kept iff word (idx & 3) of Philox4x32-10(counter = {idx >> 2, stream}, key = seed) is >= p * 2^32,
// and a kept element is scaled by 1 / (1 - p).  A counter-based generator makes the mask a pure function of
// (seed, stream, idx): the forward kernel that writes an element and the backward kernel that routes its gradient
// recompute the same bit, and the tests restate it in NumPy and compare bit for bit (tests/test_host_dropout.py).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

struct SaDrop {
    uint32_t k0, k1;   // Philox key = the 64-bit seed
    uint32_t thresh;   // keep iff word >= thresh;  thresh = floor(p * 2^32), 0 = dropout off
    float scale;       // 1 / (1 - p)
    __host__ __device__ bool on() const { return thresh != 0u; }
};

static inline SaDrop sa_drop_make(float p, unsigned long long seed) {
    SaDrop d;
    d.k0 = (uint32_t)seed; d.k1 = (uint32_t)(seed >> 32);
    if (!(p > 0.f)) { d.thresh = 0u; d.scale = 1.f; return d; }
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    if (d.thresh == 0u) d.thresh = 1u;
    d.scale = (float)(1.0 / (1.0 - (double)p));
    return d;
}
static inline bool sa_drop_valid(float p) { return p >= 0.f && p < 1.f; }

// Philox4x32-10 (Salmon et al., SC'11): ten rounds of two 32 x 32 -> 64 bit multiplies and a word permutation.
__host__ __device__ __forceinline__ void sa_philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// The four words covering elements 4 * (idx >> 2) .. + 3 of masked tensor `stream`.
__host__ __device__ __forceinline__ void sa_drop_words(const SaDrop& d, uint32_t stream, uint64_t idx, uint32_t (&w)[4]) {
    const uint64_t q = idx >> 2;
    w[0] = (uint32_t)q; w[1] = (uint32_t)(q >> 32); w[2] = stream; w[3] = 0u;
    sa_philox4x32_10(w, d.k0, d.k1);
}

// Factor applied to element idx: 0 (dropped) or 1 / (1 - p) (kept).
__host__ __device__ __forceinline__ float sa_drop_factor(const SaDrop& d, uint32_t stream, uint64_t idx) {
    uint32_t w[4];
    sa_drop_words(d, stream, idx, w);
    const uint32_t sel = (uint32_t)idx & 3u;
    const uint32_t v = sel == 0u ? w[0] : (sel == 1u ? w[1] : (sel == 2u ? w[2] : w[3]));
    return v >= d.thresh ? d.scale : 0.f;
}

// Sub-streams of one forward pass: conv layer i masks stream i, GRU layer l (its OUTPUT, l < L-1) stream 64 + l.
#define SA_DROP_STREAM_CONV 0u
#define SA_DROP_STREAM_GRU 64u
