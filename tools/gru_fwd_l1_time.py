"""Un-instrumented time of the fused forward recurrence for an L-layer stack (default 1: a layer-0-type group only, no
lower-layer rows), per time step: python tools/gru_fwd_l1_time.py [L]   (SA_GRU_ABLATE experiments: timing only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_amd import ops
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
D, B, T, H, I0 = 1, 32, 498, 512, 800
torch.manual_seed(0)
x = torch.randn(T, B, I0, device="cuda")
k = 1.0 / H ** 0.5
w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
b = [torch.zeros(3 * H, device="cuda") for l in range(L)]
for _ in range(3):
    ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("L=%d stack forward %.3f ms  (%.2f us per time step incl. the layer-0 projection and fills), ablate=%s status=%d"
      % (L, ms, ms * 1e3 / (T + L - 1), os.environ.get("SA_GRU_ABLATE", "-"), ops.persist_status()))
