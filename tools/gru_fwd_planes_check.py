"""gru_fwd_planes_kernel (option gru.fwd_planes = 1, the default) against the f32-input-MFMA one-launch kernel (= 0) and an
fp64 restatement of the stack (torch.float64 on the device): error of each against fp64, their difference, and the time of
a stack-forward call (HIP events over 10 calls).   python tools/gru_fwd_planes_check.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_amd import ops, _lib


def case(L, B, T, I0, H, seed=0):
    torch.manual_seed(seed)
    x = torch.randn(T, B, I0, device="cuda")
    k = 1.0 / H ** 0.5
    mk = lambda *s: torch.empty(*s, device="cuda").uniform_(-k, k)
    return x, [mk(3 * H, I0 if l == 0 else H) for l in range(L)], [mk(3 * H) for l in range(L)], [mk(3 * H, H) for l in range(L)], [mk(3 * H) for l in range(L)]


def ref64(x, w_ih, b_ih, w_hh, b_hh, L, H):
    inp = x.double()
    outs = []
    for l in range(L):
        Wi, Wh, bi, bh = w_ih[l].double(), w_hh[l].double(), b_ih[l].double(), b_hh[l].double()
        ai = inp @ Wi.t() + bi
        h = torch.zeros(inp.shape[1], H, dtype=torch.float64, device="cuda")
        hs = []
        for t in range(inp.shape[0]):
            ah = h @ Wh.t() + bh
            r = torch.sigmoid(ai[t, :, :H] + ah[:, :H])
            z = torch.sigmoid(ai[t, :, H:2 * H] + ah[:, H:2 * H])
            n = torch.tanh(ai[t, :, 2 * H:] + r * ah[:, 2 * H:])
            h = (1 - z) * n + z * h
            hs.append(h)
        inp = torch.stack(hs)
        outs.append(inp)
    return outs


def run(planes, args, L, H, stash=True, drop=None):
    _lib.set_option("gru.fwd_planes", planes)
    out = ops.gru_stack_fwd(*args, L, 1, H, want_stash=stash, drop=drop)
    torch.cuda.synchronize()
    assert ops.persist_status() == 0, "persist status"
    return out


def timed(planes, args, L, H):
    _lib.set_option("gru.fwd_planes", planes)
    for _ in range(3):
        ops.gru_stack_fwd(*args, L, 1, H, want_stash=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gru_stack_fwd(*args, L, 1, H, want_stash=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


shapes = [(4, 32, 120, 64, 512), (3, 20, 33, 24, 256), (2, 48, 19, 24, 128), (2, 32, 21, 40, 384), (4, 64, 30, 48, 512), (1, 7, 50, 16, 512)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:2]
for (L, B, T, I0, H) in shapes:
    args = case(L, B, T, I0, H)
    r64 = ref64(*args, L, H)
    hp, sp = run(1, args, L, H)
    hf, sf = run(0, args, L, H)
    for l in range(L):
        ep = float((hp[l].double() - r64[l]).abs().max()); ef = float((hf[l].double() - r64[l]).abs().max())
        print("L=%d B=%d T=%d H=%d layer %d: |planes - fp64| %.3e  |f32 kernel - fp64| %.3e  |planes - f32 kernel| %.3e  stash diff %.3e"
              % (L, B, T, H, l, ep, ef, float((hp[l] - hf[l]).abs().max()), float((sp[l] - sf[l]).abs().max())))
    hp2, _ = run(1, args, L, H, stash=False)
    print("   no-stash instance equal:", all(torch.equal(a, b) for a, b in zip(hp, hp2)), " finite:", all(bool(torch.isfinite(t).all()) for t in hp + sp))
if "quick" not in sys.argv:
    args = case(4, 32, 498, 800, 512)
    for _ in range(2):
        print("S-LIBRI stack forward: planes %.3f ms   f32 kernel %.3f ms" % (timed(1, args, 4, 512), timed(0, args, 4, 512)))
    args = case(1, 32, 498, 800, 512)
    print("L=1: planes %.3f ms   f32 kernel %.3f ms" % (timed(1, args, 1, 512), timed(0, args, 1, 512)))
    args = case(2, 32, 498, 800, 512)
    print("L=2: planes %.3f ms   f32 kernel %.3f ms" % (timed(1, args, 2, 512), timed(0, args, 2, 512)))
    d = (0.3, 11, 64)
    args = case(4, 32, 45, 40, 512)
    a = run(1, args, 4, 512, drop=d); b = run(0, args, 4, 512, drop=d)
    print("dropout 0.3: max |planes - f32 kernel| over h / stash / h_drop: %.3e / %.3e / %.3e" % (
        max(float((x - y).abs().max()) for x, y in zip(a[0], b[0])), max(float((x - y).abs().max()) for x, y in zip(a[1], b[1])),
        max(float((x - y).abs().max()) for x, y in zip(a[2], b[2]))))
_lib.set_option("gru.fwd_planes", 1)
