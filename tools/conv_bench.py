"""Conv kernel times at the S-LIBRI first conv and the TIMIT stacked conv: python tools/conv_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd import ops

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for name, (B, C, T, F, O, kh, kw, s, need_dx) in {"slibri1": (32, 1, 1000, 80, 32, 5, 32, 2, False),
                                                   "timit1": (8, 1, 300, 161, 32, 5, 32, 2, False),
                                                   "timit2": (8, 32, 148, 65, 32, 5, 32, 1, True),
                                                   "wsj2": (16, 32, 398, 25, 32, 5, 8, 2, True)}.items():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, C, T, F, device="cuda", generator=g)
    w = torch.randn(O, C, kh, kw, device="cuda", generator=g) / (C * kh * kw) ** 0.5
    b = torch.randn(O, device="cuda", generator=g) * 0.1
    y, ys = ops.conv2d_relu_fwd(x, w, b, s, "nchw")[:2]
    dy = torch.randn_like(y)
    tf = bench(lambda: ops.conv2d_relu_fwd(x, w, b, s, "nchw"))
    tb = bench(lambda: ops.conv2d_relu_bwd(x, w, y, dy, ys, s, need_dx=need_dx))
    fl = 2.0 * y.numel() * C * kh * kw
    print("%-8s fwd %.3f ms (%.1f TF)  bwd %.3f ms (%.1f TF incl. dx=%s)  SA_CONV_DW2=%s" % (
        name, tf, fl / tf / 1e9, tb, fl * (2 if need_dx else 1) / tb / 1e9, need_dx, os.environ.get("SA_CONV_DW2")))
