import os, sys
os.environ["SA_GRU_TIMING"] = "1"; os.environ["SA_GRU_FUSED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from speech_amd import ops, _lib
L, D, B, T, H, I0 = 4, 1, 32, 498, 512, 800
torch.manual_seed(0)
x = torch.randn(T, B, I0, device="cuda")
k = 1.0 / H ** 0.5
w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
b = [torch.zeros(3 * H, device="cuda") for l in range(L)]
for _ in range(2):
    ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
torch.cuda.synchronize()
ws = _lib.WORKSPACE._bufs[(str(x.device), "gru_stack")]
sync = ws[ws.numel() - 16384: ws.numel()].cpu().numpy().view(np.uint64)
tim = sync[128:128 + 4 * 256].reshape(4, 2, 32, 4).astype(np.float64) * 0.01 / T   # [layer][btile][unit tile][phase] us/step
for l in range(L):
    m = tim[l].reshape(-1, 4).mean(0)
    print("layer %d: input %.2f  poll %.2f  recurrent mfma %.2f  reduce+gates %.2f  | total %.2f us/step" % (l, m[0], m[1], m[2], m[3], m.sum()))
