"""Pins for oracle/transducer_ref.c (PARITY UNPINNED against the un-vendored awni/transducer, see its header):
brute-force path enumeration, fp64 finite differences, structural properties."""
import numpy as np

from oracle import transducer_ref as R


def lattice(seed, B, T, U1, K, scale=1.0):
    rng = np.random.RandomState(seed)
    z = scale * rng.randn(B, T, U1, K)
    z = z - z.max(axis=3, keepdims=True)
    return (z - np.log(np.exp(z).sum(axis=3, keepdims=True))).astype(np.float32)


def test_matches_path_enumeration():
    rng = np.random.RandomState(1)
    for T, U, K in [(1, 0, 3), (1, 2, 4), (2, 1, 3), (3, 2, 4), (4, 3, 5), (5, 1, 3), (3, 4, 6)]:
        lp = lattice(10 * T + U, 1, T, U + 1, K)
        y = rng.randint(0, K - 1, U).astype(np.int32)
        c, _ = R.transducer_loss(lp, y, [T], [U])
        want = -R.enumerate_log_prob(lp[0].astype(np.float64), list(y), K - 1)
        assert abs(c[0] - want) < 1e-9 * max(1.0, abs(want)), (T, U, c[0], want)


def test_gradient_by_finite_differences_and_structure():
    B, T, U1, K = 3, 5, 4, 5
    lp = lattice(3, B, T, U1, K)
    ll = np.array([3, 1, 2], np.int32)
    al = np.array([5, 4, 3], np.int32)
    y = np.random.RandomState(4).randint(0, K - 1, int(ll.sum())).astype(np.int32)
    c, g = R.transducer_loss(lp, y, al, ll)
    # cells outside an utterance's (T_b, U_b + 1) window and classes other than blank / the cell's label get 0
    off = np.concatenate([[0], np.cumsum(ll)])
    for b in range(B):
        assert np.all(g[b, al[b]:] == 0) and np.all(g[b, :, ll[b] + 1:] == 0)
        for t in range(al[b]):
            for u in range(ll[b] + 1):
                keep = {K - 1} | ({int(y[off[b] + u])} if u < ll[b] else set())
                for k in range(K):
                    if k not in keep:
                        assert g[b, t, u, k] == 0
        # total flow: every alignment crosses each anti-diagonal exactly once (occupancies sum to 1 per diagonal)
        gb = -g[b, :al[b], :ll[b] + 1].sum(axis=2)
        for d in range(al[b] + ll[b]):
            s = sum(gb[t, d - t] for t in range(al[b]) if 0 <= d - t <= ll[b])
            assert abs(s - 1.0) < 1e-9
    eps = 1e-3
    rng = np.random.RandomState(5)
    for _ in range(40):
        b = rng.randint(B); t = rng.randint(al[b]); u = rng.randint(ll[b] + 1); k = rng.randint(K)
        p, m = lp.copy(), lp.copy()
        p[b, t, u, k] += eps
        m[b, t, u, k] -= eps
        fd = (R.transducer_loss(p, y, al, ll, want_grad=False)[0][b] -
              R.transducer_loss(m, y, al, ll, want_grad=False)[0][b]) / (p[b, t, u, k] - m[b, t, u, k])
        assert abs(fd - g[b, t, u, k]) < 2e-4, (b, t, u, k, fd, g[b, t, u, k])


def test_decode_static_greedy_cases():
    # a lattice whose every cell puts almost all mass on one move has exactly one likely path
    T, U1, K = 4, 3, 4
    lp = np.full((T, U1, K), -20.0)
    path = [(0, 0, 1), (0, 1, 3), (1, 1, 3), (2, 1, 2), (2, 2, 3), (3, 2, 3)]  # emit 1, blank, blank, emit 2, blank, end
    for t, u, k in path:
        lp[t, u, k] = -0.01
    hyp, score = R.decode_static(lp, beam_size=2, blank=3)
    assert hyp == (1, 2)
    assert abs(score - (-0.06)) < 1e-6
    assert R.decode_static(lp, beam_size=1, blank=3)[0] == (1, 2)


def test_torch_restatement_matches_live_reference_fixture():
    """oracle/torch_ref.TorchRefTransducer (the model part, everything but the loss) against the lattice the reference's
    own Transducer.forward produced (tests/golden/transducer_tiny.npz, made by oracle/gen_golden.py)."""
    import os
    import torch
    from oracle import torch_ref
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "transducer_tiny.npz"))
    cfg = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 11, 2]], "rnn": {"dim": 16, "bidirectional": True, "layers": 2}},
           "decoder": {"embedding_dim": 12, "layers": 2}}
    m = torch_ref.TorchRefTransducer(40, 10, cfg)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")})
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(g["x"]), torch.from_numpy(g["y_mat"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-5, atol=1e-6)
    assert np.abs(np.exp(out.numpy()).sum(axis=3) - 1).max() < 1e-5
