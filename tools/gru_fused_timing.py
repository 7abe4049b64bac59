"""In-kernel phase clocks of gru_fwd_fused_kernel<.., TIMED> (SA_GRU_TIMING=1): per layer the four phases of a time step,
the clock rate the XCD actually ran at (shader-clock cycles / wall time), polling trips beyond the first, and steps that had
to wait for the layer below.   python tools/gru_fused_timing.py [L] [steps per progress report] [nostash]"""
import os, sys
os.environ["SA_GRU_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from speech_amd import ops, _lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2:
    os.environ["SA_GRU_FWD_REPORT"] = sys.argv[2]
STASH = not (len(sys.argv) > 3 and sys.argv[3] == "nostash")
D, B, T, H, I0 = 1, 32, 498, 512, 800
SYNC = 32768
torch.manual_seed(0)
x = torch.randn(T, B, I0, device="cuda")
k = 1.0 / H ** 0.5
w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
b = [torch.zeros(3 * H, device="cuda") for l in range(L)]
for _ in range(3):
    ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=STASH)
torch.cuda.synchronize()
ws = _lib.WORKSPACE._bufs[(str(x.device), "gru_stack")]
sync = ws[ws.numel() - SYNC: ws.numel()].cpu().numpy().view(np.uint64)
tim = sync[128:128 + 8 * 64 * L].reshape(L, 2, 32, 8).astype(np.float64)   # [layer][btile][unit tile][slot]
print("L=%d report every %s steps stash=%s status=%d" % (L, os.environ.get("SA_GRU_FWD_REPORT", "4"), STASH, ops.persist_status()))
for l in range(L):
    m = tim[l].reshape(-1, 8).mean(0)
    ph = m[:4] * 0.01 / T
    mhz = m[4] / (m[5] * 0.01)
    per_bt = [tim[l, y, :, 4].mean() / (tim[l, y, :, 5].mean() * 0.01) for y in range(2)]
    print("layer %d: input %.2f  poll %.2f  recurrent mfma %.2f  reduce+gates %.2f | total %.2f us/step | clock %.0f MHz (tiles %.0f / %.0f) | "
          "poll issue %.2f  input part 2 %.2f us/step" % (l, ph[0], ph[1], ph[2], ph[3], ph.sum() + (m[6] + m[7]) * 0.01 / T, mhz, per_bt[0], per_bt[1], m[6] * 0.01 / T, m[7] * 0.01 / T))
