"""oracle/decoder_ref.py against the live-reference fixtures (tests/golden/decoder.npz, made by oracle/gen_golden.py)
and the reference's own greedy KATs (/root/reference/tests/ctc_test.py:31-43)."""
import os

import numpy as np
import pytest

from oracle import decoder_ref


def _cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "decoder.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    return z, names


def test_prefix_beam_matches_reference_fixtures(golden_dir):
    z, names = _cases(golden_dir)
    assert len(names) >= 20
    for n in names:
        beam, blank = [int(v) for v in z[n + ".meta"]]
        labels, nll = decoder_ref.decode(z[n + ".probs"], beam_size=beam, blank=blank)
        assert list(labels) == list(z[n + ".labels"]), n
        assert float(nll) == float(z[n + ".nll"]), n  # bit-exact score, not just labels


def test_demo_vector_is_the_reference_main(golden_dir):
    # ctc_decoder.py:115-126: seed 3, T=50, S=20, beam 10, blank 0
    z, _ = _cases(golden_dir)
    np.random.seed(3)
    probs = np.random.rand(50, 20)
    probs = probs / np.sum(probs, axis=1, keepdims=True)
    assert np.array_equal(probs, z["demo.probs"])
    labels, _ = decoder_ref.decode(probs)
    assert list(labels) == list(z["demo.labels"])


@pytest.mark.parametrize("pre,post", [([1, 2, 2, 0, 0, 0, 2, 1], [1, 2, 2, 1]), ([2, 2, 2], [2]), ([0, 0, 0], [])])
def test_max_decode_kats(pre, post):
    assert decoder_ref.max_decode(pre, 0) == post


def test_beam1_differs_from_greedy_sometimes():
    # SURVEY finding 4: CTC.infer (beam 1 prefix search) is not argmax-collapse.
    rng = np.random.RandomState(0)
    diff = 0
    for _ in range(10):
        z = rng.randn(60, 11)
        p = np.exp(z) / np.exp(z).sum(1, keepdims=True)
        a, _ = decoder_ref.decode(p.astype(np.float32), 1, 10)
        diff += list(a) != decoder_ref.greedy(p, 10)
    assert diff > 0
