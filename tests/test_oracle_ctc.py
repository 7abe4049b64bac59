"""oracle/ctc_ref.c (the CPU CTC restatement).  The reference's warp-ctc is not vendored and its tests hold no golden
loss (PARITY UNPINNED, see oracle/ctc_ref.c), so the restatement is pinned by brute-force path enumeration,
finite differences and torch's CPU ctc_loss."""
import itertools
import math

import numpy as np
import pytest
import torch

from oracle import ctc_ref


def brute_force_nll(acts, label, blank):
    """-log sum over all length-T paths that collapse (merge repeats, drop blanks) to `label`."""
    T, K = acts.shape
    z = acts.astype(np.float64)
    ly = z - np.log(np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)) - z.max(1, keepdims=True)
    tot = 0.0
    for path in itertools.product(range(K), repeat=T):
        out, prev = [], None
        for p in path:
            if p != blank and p != prev:
                out.append(p)
            prev = p
        if out == list(label):
            tot += math.exp(sum(ly[t, p] for t, p in enumerate(path)))
    return -math.log(tot) if tot > 0 else float("inf")


@pytest.mark.parametrize("T,K,label", [(4, 3, [0]), (5, 3, [0, 1]), (5, 3, [1, 1]), (6, 4, [2, 0, 2]), (3, 3, []),
                                       (4, 3, [0, 0])])
def test_brute_force(T, K, label):
    rng = np.random.RandomState(T * 10 + K)
    acts = rng.randn(1, T, K).astype(np.float32)
    blank = K - 1
    c, _ = ctc_ref.ctc_loss(acts, np.array(label, dtype=np.int32), [T], [len(label)], blank=blank)
    assert abs(c[0] - brute_force_nll(acts[0], label, blank)) < 1e-10


def test_blank_zero_convention():
    rng = np.random.RandomState(3)
    acts = rng.randn(1, 5, 3).astype(np.float32)
    c, _ = ctc_ref.ctc_loss(acts, np.array([1, 2], dtype=np.int32), [5], [2], blank=0)
    assert abs(c[0] - brute_force_nll(acts[0], [1, 2], 0)) < 1e-10


def test_infeasible_is_inf_with_zero_grad():
    acts = np.random.RandomState(0).randn(1, 3, 4).astype(np.float32)
    c, g = ctc_ref.ctc_loss(acts, np.array([0, 0, 1], dtype=np.int32), [3], [3])  # needs T >= 4
    assert np.isinf(c[0]) and c[0] > 0 and np.all(g == 0)
    c32, g32 = ctc_ref.ctc_loss(acts, np.array([0, 0, 1], dtype=np.int32), [3], [3], dtype=np.float32)
    assert np.isinf(c32[0]) and np.all(g32 == 0)


def _rand_case(seed, B, T, K, Lmax, ragged_T=False):
    rng = np.random.RandomState(seed)
    acts = rng.randn(B, T, K).astype(np.float32) * 2
    ll = rng.randint(0, Lmax + 1, B).astype(np.int32)
    labs = np.concatenate([rng.randint(0, K - 1, l) for l in ll] + [np.zeros(0, int)]).astype(np.int32)
    al = (rng.randint(T // 2, T + 1, B) if ragged_T else np.full(B, T)).astype(np.int32)
    return acts, labs, al, ll


@pytest.mark.parametrize("ragged", [False, True])
def test_matches_torch_cpu(ragged):
    acts, labs, al, ll = _rand_case(1, 6, 40, 9, 12, ragged)
    c, g = ctc_ref.ctc_loss(acts, labs, al, ll)
    x = torch.tensor(acts, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(x, 2).transpose(0, 1)
    lt = torch.nn.functional.ctc_loss(lp, torch.tensor(labs, dtype=torch.long), torch.tensor(al, dtype=torch.long),
                                      torch.tensor(ll, dtype=torch.long), blank=8, reduction="none")
    lt.sum().backward()
    np.testing.assert_allclose(c, lt.detach().numpy(), rtol=1e-12)
    gt = x.grad.numpy().copy()
    for b in range(len(al)):
        gt[b, al[b]:] = 0
    np.testing.assert_allclose(g, gt, atol=1e-12)


def test_finite_differences():
    acts, labs, al, ll = _rand_case(2, 2, 12, 5, 4)
    c, g = ctc_ref.ctc_loss(acts, labs, al, ll)
    eps = 1e-3
    rng = np.random.RandomState(0)
    for _ in range(20):
        b, t, k = rng.randint(2), rng.randint(12), rng.randint(5)
        ap, am = acts.copy(), acts.copy()
        ap[b, t, k] += eps
        am[b, t, k] -= eps
        cp, _ = ctc_ref.ctc_loss(ap, labs, al, ll, want_grad=False)
        cm, _ = ctc_ref.ctc_loss(am, labs, al, ll, want_grad=False)
        fd = (cp[b] - cm[b]) / (float(ap[b, t, k]) - float(am[b, t, k]))
        assert abs(fd - g[b, t, k]) < 1e-5


def test_f32_port_tracks_f64_and_layouts_agree():
    acts, labs, al, ll = _rand_case(4, 5, 120, 29, 30, ragged_T=True)
    c, g = ctc_ref.ctc_loss(acts, labs, al, ll)
    c32, g32 = ctc_ref.ctc_loss(acts, labs, al, ll, dtype=np.float32, num_threads=2)
    np.testing.assert_allclose(c32, c, rtol=1e-5)
    np.testing.assert_allclose(g32, g, atol=5e-4)
    ct, gt = ctc_ref.ctc_loss(np.ascontiguousarray(acts.transpose(1, 0, 2)), labs, al, ll, batch_first=False)
    assert np.array_equal(ct, c) and np.array_equal(gt.transpose(1, 0, 2), g)


def test_rejects_bad_labels():
    acts = np.zeros((1, 4, 3), dtype=np.float32)
    with pytest.raises(ValueError):
        ctc_ref.ctc_loss(acts, np.array([2], dtype=np.int32), [4], [1])  # label == blank
    with pytest.raises(ValueError):
        ctc_ref.ctc_loss(acts, np.array([5], dtype=np.int32), [4], [1])
