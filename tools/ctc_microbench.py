"""Event-timed micro-benchmark of the CTC kernels (M-CTC and the saturating-batch variant).  GPU only."""
import sys
import os
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_amd.ctc import ctc_loss_raw


def bench(B, T, K, L, iters=20):
    rng = np.random.RandomState(2017)
    acts = torch.from_numpy(rng.randn(B, T, K).astype(np.float32)).cuda()
    labs = torch.from_numpy(rng.randint(0, K - 1, B * L).astype(np.int32))
    al = torch.full((B,), T, dtype=torch.int32)
    ll = torch.full((B,), L, dtype=torch.int32)
    for _ in range(3):
        ctc_loss_raw(acts, labs, al, ll)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ctc_loss_raw(acts, labs, al, ll)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    alg = B * T * K * 4 * 2 + B * L * 4 + B * 4
    print("B=%d T=%d K=%d L=%d: %.3f ms/call  %.0f utt/s  alg %.2f GB/s" % (B, T, K, L, ms, B / ms * 1e3, alg / ms / 1e6))


if __name__ == "__main__":
    bench(32, 1000, 29, 100)
    bench(256, 1000, 29, 100)
    bench(2048, 1000, 29, 100, iters=5)


def bench_score_only(B, T, K, L, iters=20):
    rng = np.random.RandomState(2017)
    acts = torch.from_numpy(rng.randn(B, T, K).astype(np.float32)).cuda()
    labs = torch.from_numpy(rng.randint(0, K - 1, B * L).astype(np.int32))
    al = torch.full((B,), T, dtype=torch.int32)
    ll = torch.full((B,), L, dtype=torch.int32)
    for _ in range(3):
        ctc_loss_raw(acts, labs, al, ll, want_grad=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ctc_loss_raw(acts, labs, al, ll, want_grad=False)
    e1.record()
    torch.cuda.synchronize()
    print("score-only B=%d T=%d L=%d: %.3f ms/call" % (B, T, L, e0.elapsed_time(e1) / iters))


if __name__ == "__main__":
    bench_score_only(32, 1000, 29, 100)
    bench_score_only(32, 1000, 29, 60)
