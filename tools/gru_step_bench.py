"""Times sa_gru_stack_fwd at the S-LIBRI shape (L=4, B=32, T'=498, H=512) -- ablations via SA_GRU_DBG."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_amd import ops
L, D, B, T, H, I0 = 4, 1, 32, 498, 512, 800
torch.manual_seed(0)
x = torch.randn(T, B, I0, device="cuda")
k = 1.0 / H ** 0.5
w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
b_ih = [torch.zeros(3 * H, device="cuda") for l in range(L)]
b_hh = [torch.zeros(3 * H, device="cuda") for l in range(L)]
for chunk in [int(c) for c in os.environ.get("CHUNKS", "32").split(",")]:
    for stash in (True,):
        for _ in range(2):
            ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=stash, chunk=chunk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=stash, chunk=chunk)
        e1.record(); torch.cuda.synchronize()
        print("dbg=%s chunk=%d stash=%s: %.3f ms per stack fwd" % (os.environ.get("SA_GRU_DBG", "0"), chunk, stash, e0.elapsed_time(e1) / 3))
