"""The one-operator Transducer joint (sa_joint_fused_*: relu(xa + ya) -> fc2 -> log_softmax without the (B,T,U1,H)
tensor; reference: speech/models/transducer_model.py:72-76) against (a) a float64 torch restatement of those three lines
on the CPU and (b) the unfused HIP operators, forward and every gradient.  Tolerances: the products are fp32 MFMA with
fp32 accumulation over H <= 512 terms (forward) and over up to B*T*U1 lattice rows (weight gradient), so gradients are
compared relative to the tensor's magnitude (2e-5 of max|ref|, the bound used for the other fp32 GEMM tests)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # B, T, U1, H, K
    (2, 37, 21, 128, 29),
    (3, 70, 16, 256, 29),
    (2, 131, 33, 512, 29),
    (1, 5, 3, 192, 32),
    (2, 9, 17, 64, 7),
    (1, 1, 1, 512, 29),
]


def _inputs(B, T, U1, H, K, seed):
    g = torch.Generator().manual_seed(seed)
    xa = torch.randn(B, T, H, generator=g)
    ya = torch.randn(B, U1, H, generator=g)
    w = torch.randn(K, H, generator=g) / H ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    glp = torch.randn(B, T, U1, K, generator=g)
    return xa, ya, w, b, glp


def _reference(xa, ya, w, b, glp):
    xa, ya, w, b = [t.double().requires_grad_(True) for t in (xa, ya, w, b)]
    z = torch.relu(xa[:, :, None, :] + ya[:, None, :, :])
    out = torch.log_softmax(z @ w.t() + b, dim=3)
    out.backward(glp.double())
    return out.detach(), xa.grad, ya.grad, w.grad, b.grad


def _close(got, ref, what, tol=2e-5):
    got, ref = got.double().cpu(), ref.double()
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1.0)
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_joint_matches_float64(shape):
    from speech_amd import transducer as tr
    B, T, U1, H, K = shape
    assert tr.joint_fused_supported(B, T, U1, H, K)
    xa, ya, w, b, glp = _inputs(*shape, seed=11)
    ref = _reference(xa, ya, w, b, glp)
    dev = [t.cuda().requires_grad_(True) for t in (xa, ya, w, b)]
    out = tr.FusedJointFunction.apply(*dev)
    out.backward(glp.cuda())
    _close(out.detach(), ref[0], "logp", tol=1e-5)
    for t, r, name in zip(dev, ref[1:], ("dxa", "dya", "dW2", "db2")):
        _close(t.grad, r, name)
    # a lattice row is a distribution
    assert torch.allclose(out.detach().exp().sum(-1), torch.ones(B, T, U1, device="cuda"), atol=1e-5)


def test_fused_joint_matches_unfused_operators():
    from speech_amd import ops, transducer as tr
    shape = (2, 50, 19, 256, 29)
    xa, ya, w, b, glp = _inputs(*shape, seed=5)
    dev = [t.cuda().requires_grad_(True) for t in (xa, ya, w, b)]
    out = tr.FusedJointFunction.apply(*dev)
    out.backward(glp.cuda())
    xa2, ya2 = xa.cuda().requires_grad_(True), ya.cuda().requires_grad_(True)
    z = tr.JointFunction.apply(xa2, ya2)
    logits = ops.gemm(z.detach().view(-1, shape[3]), w.cuda(), trans_b=True, bias=b.cuda()).view(*shape[:3], shape[4])
    logits.requires_grad_(True)
    out2 = tr.LogSoftmaxFunction.apply(logits)
    out2.backward(glp.cuda())
    _close(out.detach(), out2.detach().cpu(), "logp vs unfused", tol=1e-5)
    dz = ops.gemm(logits.grad.view(-1, shape[4]), w.cuda()).view_as(z)
    z.backward(dz)
    _close(dev[0].grad, xa2.grad.cpu(), "dxa vs unfused")
    _close(dev[1].grad, ya2.grad.cpu(), "dya vs unfused")


def test_fused_joint_is_deterministic_and_rejects_other_shapes():
    from speech_amd import transducer as tr
    shape = (2, 40, 18, 128, 29)
    xa, ya, w, b, glp = _inputs(*shape, seed=3)
    runs = []
    for _ in range(2):
        dev = [t.cuda().requires_grad_(True) for t in (xa, ya, w, b)]
        out = tr.FusedJointFunction.apply(*dev)
        out.backward(glp.cuda())
        runs.append([out.detach()] + [t.grad for t in dev])
    for p, q in zip(*runs):
        assert torch.equal(p, q)
    assert not tr.joint_fused_supported(2, 40, 18, 100, 29)   # H % 64
    assert not tr.joint_fused_supported(2, 40, 18, 1024, 29)  # H > 512
    assert not tr.joint_fused_supported(2, 40, 18, 128, 40)   # K > 32
