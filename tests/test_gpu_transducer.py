"""HIP RNN-Transducer loss (speech_amd.transducer -> sa_transducer_loss) and the joint-network kernels against the CPU
oracle (oracle/transducer_ref.c, fp64; PARITY UNPINNED against the un-vendored awni/transducer, pinned by enumeration
and finite differences in tests/test_oracle_transducer.py).

Tolerances: cost rtol 1e-5; gradient entries are probabilities in [0, 1] formed as exp2(alpha + w + beta - log2 p) with
alpha, beta renormalised every 8 diagonals, so their absolute error is a few ulp of the spread of the states on one
diagonal: atol = max(2e-5, 8 * 2^-24 * |log2 p|max) -- the rule tests/test_gpu_ctc.py uses."""
import numpy as np
import pytest
import torch

from oracle import transducer_ref as R

pytestmark = pytest.mark.gpu


def grad_atol(costs):
    c = np.asarray(costs)
    c = c[np.isfinite(c)]
    return max(2e-5, 8.0 * 2.0 ** -24 * (np.abs(c).max() if c.size else 0.0) / np.log(2.0))


def lattice(seed, B, T, U1, K, scale=1.0):
    rng = np.random.RandomState(seed)
    z = (scale * rng.randn(B, T, U1, K)).astype(np.float32)
    return torch.log_softmax(torch.from_numpy(z), dim=3).numpy()


def make(seed, B, T, Umax, K, ragged=True):
    rng = np.random.RandomState(seed)
    ll = (rng.randint(0, Umax + 1, B) if ragged else np.full(B, Umax)).astype(np.int32)
    ll[rng.randint(B)] = Umax
    al = (rng.randint(max(T // 2, 1), T + 1, B) if ragged else np.full(B, T)).astype(np.int32)
    al[rng.randint(B)] = T
    labs = np.concatenate([rng.randint(0, K - 1, l) for l in ll] + [np.zeros(0, int)]).astype(np.int32)
    return labs, al, ll


def run(lp, labs, al, ll, blank=None, want_grad=True):
    from speech_amd.transducer import transducer_loss_raw
    c, g = transducer_loss_raw(torch.from_numpy(lp).cuda(), torch.from_numpy(labs), torch.from_numpy(al),
                               torch.from_numpy(ll), blank=blank, want_grad=want_grad)
    torch.cuda.synchronize()
    return c.cpu().numpy(), (g.cpu().numpy() if g is not None else None)


@pytest.mark.parametrize("B,T,Umax,K,scale", [
    (1, 1, 0, 3, 1.0), (1, 1, 2, 4, 1.0), (2, 2, 1, 3, 1.0), (3, 5, 3, 5, 1.0),
    (4, 48, 20, 11, 1.0),       # the reference's test shapes (tests/shared.py)
    (5, 60, 63, 29, 1.0),       # U+1 = 64: one value per lane
    (3, 40, 64, 29, 1.0),       # U+1 = 65: two per lane
    (2, 30, 200, 12, 1.0),      # four per lane
    (2, 150, 100, 29, 4.0),     # peaky lattice
    (2, 20, 300, 6, 1.0),       # eight per lane
])
def test_matches_oracle(B, T, Umax, K, scale):
    lp = lattice(B * 100 + T, B, T, Umax + 1, K, scale)
    labs, al, ll = make(T + Umax, B, T, Umax, K)
    c, g = run(lp, labs, al, ll)
    co, go = R.transducer_loss(lp, labs, al, ll)
    np.testing.assert_allclose(c, co, rtol=1e-5)
    assert np.isfinite(g).all()
    assert np.abs(g - go).max() < grad_atol(co), (np.abs(g - go).max(), grad_atol(co))
    assert np.array_equal(g == 0, np.abs(go) < 1e-300) or np.abs(g[go == 0]).max() == 0
    c2, g2 = run(lp, labs, al, ll, want_grad=False)
    assert g2 is None and np.array_equal(c, c2)


def test_blank_zero_and_full_size():
    lp = lattice(5, 3, 30, 9, 7)
    rng = np.random.RandomState(6)
    ll = np.array([8, 3, 0], np.int32)
    labs = rng.randint(1, 7, int(ll.sum())).astype(np.int32)
    al = np.array([30, 22, 30], np.int32)
    c, g = run(lp, labs, al, ll, blank=0)
    co, go = R.transducer_loss(lp, labs, al, ll, blank=0)
    np.testing.assert_allclose(c, co, rtol=1e-5)
    assert np.abs(g - go).max() < grad_atol(co)
    # S-LIBRI-sized lattice rows (T' = 498, U + 1 = 101, V + 1 = 29), small batch so the oracle finishes in seconds
    lp = lattice(2017, 4, 498, 101, 29)
    labs, al, ll = make(9, 4, 498, 100, 29, ragged=False)
    c, g = run(lp, labs, al, ll)
    co, go = R.transducer_loss(lp, labs, al, ll)
    np.testing.assert_allclose(c, co, rtol=1e-5)
    assert np.abs(g - go).max() < grad_atol(co), (np.abs(g - go).max(), grad_atol(co))
    # every alignment crosses each anti-diagonal exactly once: occupancies of a diagonal sum to 1
    occ = -g[0].sum(axis=2)
    for d in (0, 57, 300, 597):
        s = sum(occ[t, d - t] for t in range(498) if 0 <= d - t <= 100)
        assert abs(s - 1.0) < 1e-4


def test_module_reduction_and_backward():
    from speech_amd.transducer import TransducerLoss
    lp = lattice(7, 3, 12, 6, 5)
    labs, al, ll = make(8, 3, 12, 5, 5)
    x = torch.from_numpy(lp).cuda().requires_grad_(True)
    loss = TransducerLoss()(x, torch.IntTensor(labs), torch.IntTensor(al), torch.IntTensor(ll))
    assert loss.shape == (1,)
    loss.backward()
    co, go = R.transducer_loss(lp, labs, al, ll)
    assert abs(float(loss.item()) - co.sum() / 3) < 1e-5 * co.sum()
    assert np.abs(x.grad.cpu().numpy() - go / 3).max() < grad_atol(co)


def test_joint_pieces_match_torch():
    from speech_amd import transducer as tr
    torch.manual_seed(3)
    B, T, U1, H, K, V, E = 2, 7, 5, 24, 9, 8, 12
    xa = torch.randn(B, T, H, dtype=torch.float64)
    ya = torch.randn(B, U1, H, dtype=torch.float64)
    w = torch.randn(B, T, U1, H, dtype=torch.float64)
    xa_r, ya_r = xa.clone().requires_grad_(True), ya.clone().requires_grad_(True)
    (torch.relu(xa_r.unsqueeze(2) + ya_r.unsqueeze(1)) * w).sum().backward()
    xa_g, ya_g = xa.float().cuda().requires_grad_(True), ya.float().cuda().requires_grad_(True)
    z = tr.JointFunction.apply(xa_g, ya_g)
    (z * w.float().cuda()).sum().backward()
    np.testing.assert_allclose(z.detach().cpu().numpy(), torch.relu(xa.unsqueeze(2) + ya.unsqueeze(1)).numpy(),
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(xa_g.grad.cpu().numpy(), xa_r.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ya_g.grad.cpu().numpy(), ya_r.grad.numpy(), rtol=1e-5, atol=1e-5)
    # log_softmax
    x = torch.randn(B * T, U1, K, dtype=torch.float64) * 3
    wy = torch.randn_like(x)
    xr = x.clone().requires_grad_(True)
    (torch.log_softmax(xr, dim=2) * wy).sum().backward()
    xg = x.float().cuda().requires_grad_(True)
    y = tr.LogSoftmaxFunction.apply(xg)
    (y * wy.float().cuda()).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), torch.log_softmax(x, dim=2).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    xw = torch.randn(3, 100, dtype=torch.float64) * 2   # K > 64: strided lanes
    yw = tr.LogSoftmaxFunction.apply(xw.float().cuda())
    np.testing.assert_allclose(yw.cpu().numpy(), torch.log_softmax(xw, dim=1).numpy(), rtol=1e-5, atol=1e-5)
    # embedding
    table = torch.randn(V, E, dtype=torch.float64)
    idx = torch.randint(0, V, (B, U1))
    we = torch.randn(B, U1, E, dtype=torch.float64)
    tr_ = table.clone().requires_grad_(True)
    (torch.nn.functional.embedding(idx, tr_) * we).sum().backward()
    tg = table.float().cuda().requires_grad_(True)
    e = tr.EmbeddingFunction.apply(idx, tg)
    (e * we.float().cuda()).sum().backward()
    np.testing.assert_allclose(e.detach().cpu().numpy(), table[idx].numpy(), rtol=1e-6)
    np.testing.assert_allclose(tg.grad.cpu().numpy(), tr_.grad.numpy(), rtol=1e-5, atol=1e-5)
    # embedding gradient over several 256-index scan chunks, a table wider than one pass of the block, unused rows
    for (Vb, Eb, nb) in ((28, 512, 3200), (7, 1100, 700), (300, 36, 257)):
        table = torch.randn(Vb, Eb, dtype=torch.float64)
        idx = torch.randint(0, Vb - 1, (nb // 10, 10))   # the last row never occurs: its gradient must be zero
        we = torch.randn(nb // 10, 10, Eb, dtype=torch.float64)
        tr_ = table.clone().requires_grad_(True)
        (torch.nn.functional.embedding(idx, tr_) * we).sum().backward()
        tg = table.float().cuda().requires_grad_(True)
        (tr.EmbeddingFunction.apply(idx, tg) * we.float().cuda()).sum().backward()
        np.testing.assert_allclose(tg.grad.cpu().numpy(), tr_.grad.numpy(), rtol=1e-4, atol=1e-4)
        assert float(tg.grad[Vb - 1].abs().max()) == 0.0
    # prediction GRU against nn.GRU
    gru = torch.nn.GRU(E, H, num_layers=2, batch_first=True).double()
    xin = torch.randn(B, U1, E, dtype=torch.float64)
    wo = torch.randn(B, U1, H, dtype=torch.float64)
    xr = xin.clone().requires_grad_(True)
    (gru(xr)[0] * wo).sum().backward()
    ps = []
    for l in range(2):
        ps += [getattr(gru, "%s_l%d" % (n, l)).detach().float().cuda().requires_grad_(True)
               for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    xg = xin.float().cuda().requires_grad_(True)
    out = tr.GRUStackFunction.apply(xg, H, *ps)
    (out * wo.float().cuda()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), gru(xin)[0].detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-4)
    for p, n in zip(ps, [(n, l) for l in range(2) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]):
        want = getattr(gru, "%s_l%d" % n).grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), want, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(want).max()))
