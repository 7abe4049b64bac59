"""HIP CTC decoders (speech_amd.decoder -> sa_ctc_beam_decode / sa_ctc_greedy_decode) against
  * the live-reference fixtures tests/golden/decoder.npz (labels must match EXACTLY), and
  * oracle/decoder_ref.py (bit-exact restatement of ctc_decoder.py) on fresh seeded batches."""
import os

import numpy as np
import pytest
import torch

from oracle import decoder_ref

pytestmark = pytest.mark.gpu


def softmax32(z):
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


def test_golden_fixtures_exact(golden_dir):
    from speech_amd import decoder
    z = np.load(os.path.join(golden_dir, "decoder.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    checked = 0
    for n in names:
        beam, blank = [int(v) for v in z[n + ".meta"]]
        probs = z[n + ".probs"]
        if probs.dtype != np.float32:  # the reference's float64 demo vector: compare on its float32 cast
            probs = probs.astype(np.float32)
            want, want_nll = decoder_ref.decode(probs, beam, blank)
        else:
            want, want_nll = tuple(int(v) for v in z[n + ".labels"]), float(z[n + ".nll"])
        got, nll = decoder.decode(probs, beam_size=beam, blank=blank)
        assert tuple(got) == tuple(want), n
        if np.isfinite(want_nll):
            assert abs(nll - float(want_nll)) <= 1e-5 * max(1.0, abs(float(want_nll))), n
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("beam", [1, 8])
def test_m_dec_batch_matches_restatement(beam):
    # SURVEY 8d M-DEC shapes (T'=498, S=29, blank=28), batch trimmed so the Python oracle finishes in seconds
    from speech_amd import decoder
    rng = np.random.RandomState(2017 + beam)
    B = 6 if beam == 1 else 3
    probs = softmax32(4.0 * rng.randn(B, 498, 29))
    got, nll = decoder.beam_decode(torch.from_numpy(probs).cuda(), beam_size=beam, blank=28)
    nll = nll.cpu().numpy()
    for b in range(B):
        want, want_nll = decoder_ref.decode(probs[b], beam, 28)
        assert got[b] == tuple(want), b
        assert abs(nll[b] - float(want_nll)) <= 1e-5 * abs(float(want_nll))


def test_ragged_lengths_and_small_alphabets():
    from speech_amd import decoder
    rng = np.random.RandomState(5)
    probs = softmax32(3.0 * rng.randn(5, 70, 7))
    lens = [70, 1, 33, 64, 65]
    got, _ = decoder.beam_decode(torch.from_numpy(probs).cuda(), beam_size=4, blank=0, lengths=lens)
    for b, n in enumerate(lens):
        assert got[b] == tuple(decoder_ref.decode(probs[b, :n], 4, 0)[0])


def test_fused_softmax_path_agrees_with_probability_path():
    from speech_amd import decoder
    rng = np.random.RandomState(9)
    logits = (5.0 * rng.randn(4, 120, 29)).astype(np.float32)
    a, _ = decoder.beam_decode(torch.from_numpy(logits).cuda(), beam_size=1, blank=28, input_is_logits=True)
    for b in range(4):
        assert a[b] == tuple(decoder_ref.decode(softmax32(logits[b]), 1, 28)[0])


def test_greedy_kats_and_random():
    from speech_amd import decoder
    # /root/reference/tests/ctc_test.py:31-43, as one-hot frames
    for pre, post in [([1, 2, 2, 0, 0, 0, 2, 1], [1, 2, 2, 1]), ([2, 2, 2], [2]), ([0, 0, 0], [])]:
        x = np.zeros((1, len(pre), 3), dtype=np.float32)
        x[0, np.arange(len(pre)), pre] = 1.0
        assert decoder.greedy_decode(torch.from_numpy(x).cuda(), blank=0) == [tuple(post)]
    rng = np.random.RandomState(1)
    x = rng.randn(7, 300, 29).astype(np.float32)
    lens = [300, 299, 64, 65, 1, 128, 200]
    got = decoder.greedy_decode(torch.from_numpy(x).cuda(), blank=28, lengths=lens)
    for b, n in enumerate(lens):
        assert list(got[b]) == decoder_ref.greedy(x[b, :n], 28)


def test_thirty_random_small_shapes_match_the_restatement():
    """A fuzz over what the fixed cases do not enumerate: 1 .. 9 utterances of 1 .. 60 frames (ragged), alphabets of 2 .. 40, beams
    1 .. 16 (beam 1 = the one-wave kernel with its row ring and shared log-sum-exp, r6), blank first or last, peaked and flat
    rows: labels identical to oracle/decoder_ref.py, scores to 1e-5."""
    from speech_amd import decoder
    rng = np.random.RandomState(930)
    for i in range(30):
        B, T, S = int(rng.randint(1, 10)), int(rng.randint(1, 61)), int(rng.randint(2, 41))
        beam = int(rng.choice([1, 1, 2, 3, 5, 8, 16]))
        blank = 0 if rng.randint(2) else S - 1
        probs = softmax32(float(rng.choice([0.5, 2.0, 6.0])) * rng.randn(B, T, S))
        lens = [int(rng.randint(1, T + 1)) for _ in range(B)]
        got, nll = decoder.beam_decode(torch.from_numpy(probs).cuda(), beam_size=beam, blank=blank, lengths=lens)
        nll = nll.cpu().numpy()
        for b in range(B):
            want, want_nll = decoder_ref.decode(probs[b, :lens[b]], beam, blank)
            assert got[b] == tuple(want), (i, b, B, T, S, beam, blank)
            if np.isfinite(want_nll):
                assert abs(nll[b] - float(want_nll)) <= 1e-5 * max(1.0, abs(float(want_nll))), (i, b)
