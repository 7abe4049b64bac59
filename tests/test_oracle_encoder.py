"""oracle/encoder_np.py against the live-reference fixtures (tests/golden/encoder_*.npz: the reference's own
Model/CTC classes run on CPU by oracle/gen_golden.py) and against torch.nn float64 autograd."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import encoder_np as E

CFGS = {
    "encoder_tiny": (40, 10, {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]],
                                                         "rnn": {"dim": 16, "bidirectional": False, "layers": 1}}}),
    "encoder_bi2": (40, 12, {"dropout": 0.0, "encoder": {"conv": [[8, 5, 11, 2], [8, 3, 7, 1]],
                                                        "rnn": {"dim": 24, "bidirectional": True, "layers": 2}}}),
    "encoder_uni3": (80, 28, {"dropout": 0.0, "encoder": {"conv": [[16, 5, 32, 2]],
                                                         "rnn": {"dim": 32, "bidirectional": False, "layers": 3}}}),
}


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    P = {k[len("param."):]: z[k] for k in z.files if k.startswith("param.")}
    return z, P


@pytest.mark.parametrize("name", sorted(CFGS))
def test_forward_matches_reference_fixture(golden_dir, name):
    F, V, cfg = CFGS[name]
    z, P = load_case(golden_dir, name)
    logits, cache = E.model_fwd(P, z["x"], cfg, dtype=np.float64)
    assert logits.shape == z["logits"].shape
    # reference ran in float32 on CPU; the fp64 restatement must agree to fp32 round-off
    np.testing.assert_allclose(logits, z["logits"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(cache["enc"], z["enc"], rtol=2e-5, atol=2e-6)
    assert logits.shape[1] == int(z["t_out"]) == E.conv_out_size(z["x"].shape[1], cfg["encoder"]["conv"], 0)
    assert int(z["f_out"]) == E.conv_out_size(F, cfg["encoder"]["conv"], 1)


def test_shapes_of_the_reference_tests(golden_dir):
    # /root/reference/tests/ctc_test.py:19-24 and model_test.py:18-23: (4,48,11) and (4,48,16)
    z, _ = load_case(golden_dir, "encoder_tiny")
    assert z["logits"].shape == (4, 48, 11) and z["enc"].shape == (4, 48, 16)


def test_state_dict_names(golden_dir):
    F, V, cfg = CFGS["encoder_bi2"]
    _, P = load_case(golden_dir, "encoder_bi2")
    mine = E.init_params(F, V, cfg)
    assert set(mine) == set(P)
    for k in P:
        assert mine[k].shape == P[k].shape, k


def _torch_model(P, cfg, F, V):
    convs, in_c = [], 1
    for o, h, w, s in cfg["encoder"]["conv"]:
        convs += [nn.Conv2d(in_c, o, (h, w), stride=(s, s)), nn.ReLU()]
        in_c = o
    r = cfg["encoder"]["rnn"]

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Sequential(*convs)
            self.rnn = nn.GRU(in_c * E.conv_out_size(F, cfg["encoder"]["conv"], 1), r["dim"], r["layers"],
                              batch_first=True, bidirectional=r["bidirectional"])
            self.fc = nn.Module()
            self.fc.fc = nn.Linear(r["dim"], V + 1)

        def forward(self, x):
            a = self.conv(x.unsqueeze(1)).transpose(1, 2).contiguous()
            b, t, f, c = a.size()
            o, _ = self.rnn(a.view(b, t, f * c))
            if r["bidirectional"]:
                h = o.size(-1) // 2
                o = o[:, :, :h] + o[:, :, h:]
            return self.fc.fc(o)

    m = M().double()
    m.load_state_dict({k: torch.tensor(np.asarray(v, dtype=np.float64)) for k, v in P.items()})
    return m


@pytest.mark.parametrize("name", sorted(CFGS))
def test_backward_matches_torch_float64(golden_dir, name):
    F, V, cfg = CFGS[name]
    z, P = load_case(golden_dir, name)
    rng = np.random.RandomState(5)
    logits, cache = E.model_fwd(P, z["x"], cfg, dtype=np.float64)
    dl = rng.randn(*logits.shape)
    G = E.model_bwd(cache, dl)
    m = _torch_model(P, cfg, F, V)
    out = m(torch.tensor(z["x"], dtype=torch.float64))
    np.testing.assert_allclose(out.detach().numpy(), logits, rtol=1e-12, atol=1e-12)
    (out * torch.tensor(dl)).sum().backward()
    for k, p in m.named_parameters():
        np.testing.assert_allclose(G[k], p.grad.numpy(), rtol=1e-9, atol=1e-11, err_msg=k)


def test_clip_and_sgd_matches_torch():
    rng = np.random.RandomState(0)
    P = {"a": rng.randn(7, 5), "b": rng.randn(11)}
    G = {"a": 100 * rng.randn(7, 5), "b": 100 * rng.randn(11)}
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}
    for k in tp:
        tp[k].grad = torch.tensor(G[k])
    opt = torch.optim.SGD(tp.values(), lr=1e-3, momentum=0.0)
    norm = torch.nn.utils.clip_grad_norm_(tp.values(), 200)
    opt.step()
    newP, total = E.clip_and_sgd(P, G, 1e-3, 200.0)
    assert abs(total - float(norm)) < 1e-9 * total and total > 200
    for k in P:
        np.testing.assert_allclose(newP[k], tp[k].detach().numpy(), rtol=1e-12)


@pytest.mark.parametrize("name", sorted(CFGS))
def test_torch_ref_port_matches_reference_fixture(golden_dir, name):
    """oracle/torch_ref.py (the timed CPU baseline) reproduces the live reference's logits bit-for-bit class."""
    from oracle.torch_ref import TorchRefCTC
    F, V, cfg = CFGS[name]
    z, P = load_case(golden_dir, name)
    m = TorchRefCTC(F, V, cfg)
    m.load_state_dict({k: torch.tensor(v) for k, v in P.items()})
    m.eval()
    with torch.no_grad():
        out = m(torch.tensor(z["x"]))
    np.testing.assert_allclose(out.numpy(), z["logits"], rtol=1e-5, atol=1e-6)


def test_torch_ref_train_step_runs_and_decreases_loss():
    from oracle.torch_ref import TorchRefCTC, train_step
    F, V, cfg = CFGS["encoder_tiny"]
    torch.manual_seed(0)
    m = TorchRefCTC(F, V, cfg)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    rng = np.random.RandomState(0)
    x = torch.tensor(rng.randn(2, 60, F).astype(np.float32))
    labs = rng.randint(0, V, 10).astype(np.int32)
    l0, _ = train_step(m, opt, x, labs, np.array([5, 5], np.int32))
    for _ in range(5):
        l1, gn = train_step(m, opt, x, labs, np.array([5, 5], np.int32))
    assert np.isfinite(l1) and l1 < l0 and gn > 0
