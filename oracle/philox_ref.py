"""
oracle/philox_ref.py -- NumPy restatement of the dropout masks the HIP kernels generate (speech_amd/csrc/dropout.h).
TEST INFRASTRUCTURE ONLY (imported by tests/; never by the product path).

What is restated: the reference applies nn.Dropout(p) behind every conv ReLU and nn.GRU(dropout=p) between the GRU
layers (/root/reference/speech/models/model.py:25-27,35-39) with masks drawn from torch's generator.  Which elements
are dropped is not part of any contract -- the distribution is (Bernoulli keep probability 1 - p, kept elements scaled
by 1 / (1 - p)) -- so the HIP path defines its masks as a pure function of (seed, mask stream, element index):

    keep(idx)  <=>  Philox4x32-10(counter = [idx >> 2 (lo), idx >> 34 (hi), stream, 0], key = [seed lo, seed hi])[idx & 3]
                    >= floor(p * 2^32)

This module computes the same integers (pinned to the Random123 known-answer vectors of Philox4x32-10 in
tests/test_host_dropout.py, and compared bit for bit with the device in tests/test_gpu_dropout.py) so that the CPU
oracles (oracle/torch_ref.py, oracle/encoder_np.py) can be run on exactly the masks the kernels used.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over counters (uint32 arrays of one shape); k0, k1 Python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, p1 & MASK32, n2, p0 & MASK32
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def threshold(p):
    """floor(p * 2^32) as the kernels compute it (double arithmetic), at least 1 when p > 0."""
    if not p > 0:
        return 0
    t = float(np.float32(p)) * 4294967296.0
    t = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    return max(t, 1)


def scale(p):
    return np.float32(1.0 / (1.0 - float(np.float32(p)))) if p > 0 else np.float32(1.0)


def words(n, seed, stream, idx0=0):
    """The mask words of elements idx0 .. idx0 + n - 1 of mask stream `stream` (uint32, length n)."""
    idx = np.arange(idx0, idx0 + n, dtype=np.uint64)
    q = idx >> np.uint64(2)
    w = philox4x32_10(q & MASK32, q >> np.uint64(32), np.full(n, stream, np.uint64), np.zeros(n, np.uint64),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    sel = (idx & np.uint64(3)).astype(np.int64)
    return np.choose(sel, w)


def mask(n, p, seed, stream, idx0=0):
    """float32 factors: 0 (dropped) or 1 / (1 - p) (kept)."""
    if not p > 0:
        return np.ones(n, np.float32)
    return np.where(words(n, seed, stream, idx0) >= np.uint32(threshold(p)), scale(p), np.float32(0)).astype(np.float32)


# mask streams of one forward pass (speech_amd/ops.py DROP_STREAM_*)
STREAM_CONV, STREAM_GRU = 0, 64


def encoder_masks(p, seed, conv_shapes, gru_shape, layers):
    """The masks of one forward pass of the encoder, in the layouts the CPU oracles use:
    conv[i] (B, O, T', F') -- the kernels index the NCHW offset; gru[l] (B, T', D*H), l < layers - 1 -- the kernels
    index the offset in the TIME-MAJOR (T', B, D*H) array, so the mask is generated there and transposed."""
    conv = [mask(int(np.prod(s)), p, seed, STREAM_CONV + i).reshape(s) for i, s in enumerate(conv_shapes)]
    B, Tp, DH = gru_shape
    gru = [mask(Tp * B * DH, p, seed, STREAM_GRU + l).reshape(Tp, B, DH).transpose(1, 0, 2).copy()
           for l in range(layers - 1)]
    return {"conv": conv, "gru": gru}
