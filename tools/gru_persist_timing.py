import os, sys
os.environ["SA_GRU_TIMING"] = "1"
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
nbytes = _lib.lib().sa_gru_stack_fwd_workspace_bytes(L, D, B, T, H, I0)
sync = ws[ws.numel() - 32768: ws.numel()].cpu().numpy().view(np.uint64)
tim = sync[128:128 + 4 * 256].reshape(-1, 4).astype(np.float64) * 0.01   # us, 10 ns ticks (sync + 256 uints = 128 u64)
steps = T  # each block saw ~T steps of its layer in the last call (accumulated over the call's chunks)
print("per-step us (mean over blocks): wait %.2f  load+mfma %.2f  epilogue %.2f  drain+publish %.2f  | total %.2f" % tuple(list(tim.mean(0) / steps) + [tim.sum(1).mean() / steps]))
print("block 0:", tim[0] / steps, " block 255:", tim[255] / steps)
