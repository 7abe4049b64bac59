"""Phase clocks of the persistent backward recurrence (SA_GRU_TIMING=1): per time step, mean over the 256 blocks.
With the clocks on, the library runs the recurrence chunk by chunk (the one-launch mode of gru_bwd_fused_kernel is off:
the accumulators are sized for one launch per chunk); the per-step phases are the same."""
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
h, st = ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
dtop = torch.randn(T, B, H, device="cuda")
for _ in range(2):
    ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, D, H, I0)
torch.cuda.synchronize()
ws = _lib.WORKSPACE._bufs[(str(x.device), "gru_stack_bwd")]
nbytes = _lib.lib().sa_gru_stack_bwd_workspace_bytes(L, D, B, T, H, I0)
sync = ws[ws.numel() - 32768: ws.numel()].cpu().numpy().view(np.uint64)
tim = sync[256:256 + 5 * 256].reshape(-1, 5).astype(np.float64)   # sync + 512 uints = 256 u64
steps = T
us = tim[:, :4] * 0.01 / steps
if os.environ.get("SA_GRU_FUSE_DX", "1") != "0":
    print("(fused kernel: the four phases are gather | recurrent mfma | barrier .. publish | tail = copies, next operands, second product)")
print("per-step us (mean over blocks): poll+load %.2f  mfma %.2f  reduce+barrier %.2f  gates+publish %.2f | total %.2f | polling trips per step %.2f"
      % (us[:, 0].mean(), us[:, 1].mean(), us[:, 2].mean(), us[:, 3].mean(), us.sum(1).mean(), tim[:, 4].mean() / steps))
if os.environ.get("SA_GRU_DBG_HOT") == "3":  # hot3: slots = flush | reduce + gates | exchange stores | row-major stores | fetch
    print("hot3 per-step us: flush %.2f  reduce+gates %.2f  exchange stores %.2f  row-major stores %.2f  fetch %.2f"
          % (us[:, 0].mean(), us[:, 1].mean(), us[:, 2].mean(), us[:, 3].mean(), tim[:, 4].mean() * 0.01 / steps))
print("min/max over blocks of poll+load: %.2f / %.2f ; trips %.2f / %.2f" % (us[:, 0].min(), us[:, 0].max(), tim[:, 4].min() / steps, tim[:, 4].max() / steps))
