"""GEMM throughput on the shapes of the S-LIBRI train step (and a square calibration point), with the error of each
result against an fp64 product: run once as is (split-bf16 kernel, the default) and once with SA_GEMM_EXACT=1 (f32-input
MFMA kernel).  rel = |C - C64|_F / |C64|_F;  max = max |C - C64| / (|A| |B|)_max (error relative to the absolute-value
product, the scale rounding errors live on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_amd import ops
ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]


def bench(name, M, N, K, ta, tb, iters=10):
    if ONLY and not any(o in name for o in ONLY):
        return
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(2): ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    err = ""
    if M * N <= 4096 * 4096 and M * K + N * K <= 64e6:
        A = (a.t() if ta else a).double()
        Bm = (b.t() if tb else b).double()
        ref = A @ Bm
        d = out.double() - ref
        scale = (A.abs() @ Bm.abs()).max()
        err = "  rel %.2e  max/abs-product %.2e" % (float(d.norm() / ref.norm()), float(d.abs().max() / scale))
    print("%-28s M=%6d N=%5d K=%6d %s%s: %8.1f us  %6.1f TF%s" % (name, M, N, K, "T" if ta else "N", "T" if tb else "N", ms * 1e3, 2.0 * M * N * K / ms / 1e9, err))
print("SA_GEMM_EXACT =", os.environ.get("SA_GEMM_EXACT", "(unset: split-bf16 kernel)"))
bench("square", 4096, 4096, 4096, False, True)
bench("square", 4096, 4096, 4096, False, False)
bench("square", 4096, 4096, 4096, True, False)
bench("i2h layer0", 15936, 1536, 800, False, True)
bench("i2h chunk (x3 grouped)", 1024, 1536, 512, False, True)
bench("dx layer0", 15936, 800, 1536, False, False)
bench("dmid chunk", 1024, 512, 1536, False, False)
bench("dW_ih l0", 1536, 800, 15936, True, False)
bench("dW_ih/hh", 1536, 512, 15936, True, False)
bench("fc fwd", 15936, 29, 512, False, True)
bench("fc dW", 29, 512, 15936, True, False)
bench("fc dx", 15936, 512, 29, False, False)
bench("conv fwd", 398400, 32, 160, False, True)
bench("conv dW", 32, 160, 398400, True, False)
