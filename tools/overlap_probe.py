"""Do long MFMA GEMMs on a second stream overlap with the latency-bound GRU step chain?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_amd import ops
L, D, B, T, H, I0 = 4, 1, 32, 498, 512, 800
torch.manual_seed(0)
x = torch.randn(T, B, I0, device="cuda")
k = 1.0 / H ** 0.5
w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
b = [torch.zeros(3 * H, device="cuda") for l in range(L)]
A = torch.randn(1024, 1536, device="cuda"); Bm = torch.randn(1024, 512, device="cuda"); C = torch.zeros(1536, 512, device="cuda")
side = torch.cuda.Stream()
def chain(): ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
def gemms(n):
    for _ in range(n): ops.gemm(A, Bm, trans_a=True, out=C, beta=1.0)
def timeit(f):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)
chain(); gemms(4)
N = 160
t_chain = timeit(chain); t_gemm = timeit(lambda: gemms(N))
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): gemms(N)
    chain()
    torch.cuda.current_stream().wait_stream(side)
t_both = timeit(both)
print("chain %.2f ms  %d gemms %.2f ms (%.1f us each)  concurrent %.2f ms  (sum %.2f)" % (t_chain, N, t_gemm, t_gemm / N * 1e3, t_both, t_chain + t_gemm))
