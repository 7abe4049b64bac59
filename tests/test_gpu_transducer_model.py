"""speech_amd.models.Transducer (the reference's transducer_model.py on the HIP ops) against
  * the lattice the LIVE reference produced (tests/golden/transducer_tiny.npz) -- pins everything except the loss,
  * oracle/torch_ref.TorchRefTransducer in fp64 + the fp64 C loss -- loss value and every parameter gradient,
  * oracle/transducer_ref.decode_static -- the static beam search (exact labels).
The loss and the decoder themselves are PARITY UNPINNED (un-vendored awni/transducer; see oracle/transducer_ref.c)."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref, transducer_ref as R

pytestmark = pytest.mark.gpu

CFG = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 11, 2]], "rnn": {"dim": 16, "bidirectional": True, "layers": 2}},
       "decoder": {"embedding_dim": 12, "layers": 2}}


def fixture():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "transducer_tiny.npz"))


def build(g, flatten=False):
    from speech_amd.models import Transducer
    m = Transducer(40, 10, CFG)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")})
    m = m.cuda()
    if flatten:
        m.flatten_parameters_()
    return m


def test_lattice_matches_live_reference():
    g = fixture()
    m = build(g)
    m.set_eval()
    with torch.no_grad():
        out = m.forward_impl(torch.from_numpy(g["x"]), torch.from_numpy(g["y_mat"]))
    assert out.shape == g["out"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=2e-4, atol=2e-5)


def test_loss_and_parameter_gradients_match_fp64_reference():
    g = fixture()
    B = g["x"].shape[0]
    ref = torch_ref.TorchRefTransducer(40, 10, CFG).double()
    ref.load_state_dict({k[len("param."):]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith("param.")})
    ref.train()
    Tp = 29  # ceil((61 - 5 + 1) / 2)
    al = np.full(B, Tp, np.int32)
    loss_ref = torch_ref.transducer_loss(ref, torch.from_numpy(g["x"]).double(), torch.from_numpy(g["y_mat"]),
                                         g["labels_flat"], al, g["label_lens"])
    loss_ref.backward()

    for flatten in (False, True):
        m = build(g, flatten)
        m.set_train()
        inputs = tuple(g["x"][b] for b in range(B))
        off = np.concatenate([[0], np.cumsum(g["label_lens"])])
        labels = tuple(g["labels_flat"][off[b]:off[b + 1]] for b in range(B))
        m.zero_grad(set_to_none=True)
        loss = m.loss((inputs, labels))
        loss.backward()
        assert abs(float(loss.item()) - float(loss_ref.item())) < 1e-4 * abs(float(loss_ref.item()))
        want = dict(ref.named_parameters())
        for name, p in m.named_parameters():
            w = want[name].grad.numpy()
            got = p.grad.cpu().numpy()
            assert np.abs(got - w).max() < 1e-3 * max(np.abs(w).max(), 1e-3), name
        if flatten:
            # every gradient landed in its slot of the flat buffer (what the fused clip+SGD and the all-reduce read)
            for name, p in m.named_parameters():
                assert torch.equal(p._grad_slot, p.grad), name


def test_decode_static_matches_restatement_and_infer_runs():
    from speech_amd import transducer as tr
    rng = np.random.RandomState(2017)
    for T, U, K, scale, beam, blank in [(12, 5, 6, 1.0, 1, 5), (12, 5, 6, 1.0, 4, 5), (30, 9, 29, 3.0, 4, 28),
                                        (30, 9, 29, 3.0, 8, 28), (20, 7, 11, 2.0, 16, 0), (1, 1, 4, 1.0, 2, 3),
                                        (5, 1, 4, 1.0, 2, 3), (1, 4, 5, 1.0, 3, 4), (40, 12, 29, 6.0, 2, 28)]:
        z = torch.from_numpy((scale * rng.randn(T, U, K)).astype(np.float32))
        lp = torch.log_softmax(z, dim=2).numpy()
        want, wscore = R.decode_static(lp, beam_size=beam, blank=blank)
        got, gscore = tr.decode_static(lp, beam_size=beam, blank=blank)
        assert got == want, (T, U, K, beam, got, want)
        assert abs(gscore - wscore) < 1e-9 * max(1.0, abs(wscore))
    # batched, ragged rows / frames
    z = torch.from_numpy((3.0 * rng.randn(4, 25, 8, 11)).astype(np.float32))
    lp = torch.log_softmax(z, dim=3)
    u1, t = [8, 5, 3, 1], [25, 20, 25, 9]
    hyps, scores = tr.decode_static_batch(lp.cuda(), u1, t, beam_size=4, blank=10)
    for b in range(4):
        want, ws = R.decode_static(lp[b, :t[b], :u1[b]].numpy(), beam_size=4, blank=10)
        assert hyps[b] == want and abs(float(scores[b]) - ws) < 1e-9 * max(1.0, abs(ws))
    g = fixture()
    m = build(g)
    m.set_eval()
    B = g["x"].shape[0]
    off = np.concatenate([[0], np.cumsum(g["label_lens"])])
    labels = tuple(g["labels_flat"][off[b]:off[b + 1]] for b in range(B))
    preds = m.infer((tuple(g["x"][b] for b in range(B)), labels))
    for b in range(B):
        want, _ = R.decode_static(g["out"][b, :, :len(labels[b]) + 1], beam_size=4, blank=10)
        assert tuple(preds[b]) == want


def test_import_shims():
    import transducer.decoders as td
    import transducer.functions.transducer as tf
    from speech.models import Transducer  # noqa: F401
    assert tf.TransducerLoss.__module__ == "speech_amd.transducer" and callable(td.decode_static)


def test_training_mode_dropout_runs_on_the_library_masks():
    """dropout != 0 (examples/timit/transducer_config.json:20 ships 0.5): encoder masks inside the HIP kernels, the
    two-layer prediction network's inter-layer mask through the library's element-wise form.  Finite loss / gradients
    in training mode, a reproducible pass under torch.manual_seed, and eval mode unaffected by the setting."""
    from speech_amd.models import Transducer
    g = fixture()
    cfg = dict(CFG, dropout=0.5)
    m = Transducer(40, 10, cfg)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")})
    m = m.cuda()
    x, y_mat = torch.from_numpy(g["x"]), torch.from_numpy(g["y_mat"])
    m.set_eval()
    with torch.no_grad():
        np.testing.assert_allclose(m.forward_impl(x, y_mat).cpu().numpy(), g["out"], rtol=2e-4, atol=2e-5)
    m.set_train()
    torch.manual_seed(3)
    a = m.forward_impl(x, y_mat)
    torch.manual_seed(3)
    b = m.forward_impl(x, y_mat)
    c = m.forward_impl(x, y_mat)
    assert torch.equal(a, b) and not torch.equal(a, c)
    a.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
