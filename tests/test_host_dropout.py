"""CPU checks of the dropout-mask restatement (oracle/philox_ref.py) and of the masked form of the torch CPU oracle.

The HIP kernels generate their Bernoulli masks with Philox4x32-10 (speech_amd/csrc/dropout.h) instead of torch's
generator (the reference: nn.Dropout / nn.GRU(dropout=p), /root/reference/speech/models/model.py:25-27,38); the NumPy
restatement is pinned here to the generator's published known-answer vectors (Random123 kat_vectors, philox4x32 10
rounds) and compared bit for bit with the device in tests/test_gpu_dropout.py."""
import numpy as np
import torch

from oracle import philox_ref


def _hex(t):
    return ["%08x" % int(x[0]) for x in t]


def test_philox4x32_10_known_answers():
    f = philox_ref.philox4x32_10
    assert _hex(f([0], [0], [0], [0], 0, 0)) == ["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]
    ff = [0xffffffff]
    assert _hex(f(ff, ff, ff, ff, 0xffffffff, 0xffffffff)) == ["408f276d", "41c83b0e", "a20bc7c6", "6d5451fd"]
    assert _hex(f([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)) == \
        ["d16cfe09", "94fdcceb", "5001e420", "24126ea1"]


def test_mask_distribution_scale_and_windows():
    for p in (0.2, 0.4, 0.5):
        m = philox_ref.mask(400000, p, seed=2017, stream=3)
        kept = m > 0
        assert abs(kept.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / m.size)
        assert np.all(m[kept] == np.float32(1.0 / (1.0 - float(np.float32(p)))))
    # a window of the index space is the same mask (the kernels mask chunks of a tensor with idx0 offsets)
    full = philox_ref.mask(1000, 0.3, 99, 64)
    np.testing.assert_array_equal(philox_ref.mask(137, 0.3, 99, 64, idx0=501), full[501:638])
    # streams and seeds decorrelate
    a, b = philox_ref.mask(100000, 0.5, 1, 0), philox_ref.mask(100000, 0.5, 1, 1)
    assert abs(((a > 0) == (b > 0)).mean() - 0.5) < 0.01
    assert np.all(philox_ref.mask(10, 0.0, 1, 0) == 1.0)


def test_masked_torch_oracle_equals_the_reference_modules_without_dropout():
    """masks of ones through TorchRefCTC.encode(masks=...) == the nn.Sequential / nn.GRU path in eval mode: the masked
    form restates the same computation (per-layer GRU calls on the shared parameters)."""
    from oracle.torch_ref import TorchRefCTC
    for bi in (False, True):
        cfg = {"dropout": 0.3, "encoder": {"conv": [[8, 5, 8, 2], [8, 3, 4, 1]],
                                           "rnn": {"dim": 16, "layers": 3, "bidirectional": bi}}}
        torch.manual_seed(0)
        m = TorchRefCTC(40, 10, cfg).eval()
        x = torch.randn(3, 50, 40)
        with torch.no_grad():
            c1 = m.conv[0](x.unsqueeze(1))
            c2 = m.conv[3](torch.relu(c1))
            D = 2 if bi else 1
            ones = {"conv": [np.ones(c1.shape, np.float32), np.ones(c2.shape, np.float32)],
                    "gru": [np.ones((3, c2.shape[2], 16 * D), np.float32)] * 2}
            assert torch.equal(m(x), m(x, ones))
            # and a real mask changes the result only through the masked activations
            masks = philox_ref.encoder_masks(0.3, 7, [tuple(c1.shape), tuple(c2.shape)], (3, c2.shape[2], 16 * D), 3)
            y = m(x, masks)
            assert torch.isfinite(y).all() and not torch.equal(y, m(x))


def test_dropout_key_differs_per_rank_and_generators_stay_in_step(monkeypatch):
    """every data-parallel rank seeds torch's generator from the same config seed and indexes its masks by LOCAL
    position -- without the rank in the key, utterance b of every rank would get the same mask.  ops.new_dropout_seed mixes
    the rank in; the DRAW itself stays the same on every rank (one draw per forward pass, also on a rank that skips the pass:
    Model.skipped_step), so the generators stay in lock-step."""
    import torch
    from speech_amd import ops
    keys = {}
    for rank in (0, 1, 2):
        monkeypatch.setenv("WORLD_SIZE", "3")
        monkeypatch.setenv("RANK", str(rank))
        torch.manual_seed(2017)
        keys[rank] = [ops.new_dropout_seed() for _ in range(3)]
        keys[rank].append(int(torch.randint(0, 2 ** 62, (1,)).item()))  # the generator's NEXT draw: same on every rank
    assert keys[0][:3] != keys[1][:3] and keys[1][:3] != keys[2][:3]
    assert all(a != b for a, b in zip(keys[0][:3], keys[1][:3]))
    assert keys[0][3] == keys[1][3] == keys[2][3]
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.delenv("RANK")
    torch.manual_seed(2017)
    assert [ops.new_dropout_seed() for _ in range(3)] == keys[0][:3]   # a single process: the plain draw
