"""What the M-CTC call time (B=32, T=1000, V+1=29, L=100; bench.py's ctc_loss_step_ms) is made of:
host time per call (enqueue only), device time per call in a back-to-back loop, the same with the chip kept busy
(a large GEMM enqueued in front of every call -- its time subtracted -- so that the clocks are where a train step has them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd.ctc import CTCLabels, ctc_loss_raw
B, T, V, L = 32, 1000, 28, 100
rng = np.random.RandomState(2017)
acts = torch.from_numpy(rng.randn(B, T, V + 1).astype(np.float32)).cuda()
lab = CTCLabels(rng.randint(0, V, B * L).astype(np.int32), np.full(B, T, np.int32), np.full(B, L, np.int32), acts.device)
n = 50
for _ in range(5): ctc_loss_raw(acts, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): ctc_loss_raw(acts, lab)
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
def ev(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
loop = ev(lambda: ctc_loss_raw(acts, lab), n)
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
from speech_amd import ops
c = torch.empty(4096, 4096, device="cuda")
g = lambda: ops.gemm(a, b, out=c)
for _ in range(3): g()
gemm = ev(g, 20)
both = ev(lambda: (g(), ctc_loss_raw(acts, lab)), 20)
print("host enqueue %.1f us/call | back-to-back loop %.1f us/call | behind a %.0f us GEMM: %.1f us/call"
      % (host * 1e6, loop, gemm, both - gemm))
