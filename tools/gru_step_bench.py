"""Times sa_gru_stack_fwd/bwd at the S-LIBRI shape (L=4, B=32, T'=498, H=512) for several chain counts."""
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
b_ih = [torch.zeros(3 * H, device="cuda") for l in range(L)]
b_hh = [torch.zeros(3 * H, device="cuda") for l in range(L)]
dtop = torch.randn(T, B, H, device="cuda")
ref = None
for chains in [int(c) for c in os.environ.get("CHAINS", "1,2,4,8").split(",")]:
    ops.N_CHAINS = chains
    for _ in range(2):
        h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=True)
        out = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, D, H, I0)
    torch.cuda.synchronize()
    if ref is None:
        ref = (h[-1].clone(), out[2].clone())
    same = torch.equal(ref[0], h[-1]) and torch.equal(ref[1], out[2])
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(3):
        h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=True)
    e[1].record()
    for _ in range(3):
        ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, D, H, I0)
    e[2].record(); torch.cuda.synchronize()
    print("chains=%d: fwd %.3f ms  bwd %.3f ms  identical=%s" % (chains, e[0].elapsed_time(e[1]) / 3, e[1].elapsed_time(e[2]) / 3, same))
