"""Phase clocks of the backward recurrence (option gru.timing = 1), per time step and per layer.
Round 6: the clocks run inside the ONE-LAUNCH kernel, in the form the train step uses (sa_gru_stack_bwd_wgrad: the kernel
packs the gate gradients itself, gru_bwd_fused_kernel<8, FUSE, false, PACKG>); `plain` as first argument clocks
sa_gru_stack_bwd (row-major copies instead of the packed operand).  Thread 0 of every block accumulates wall-clock ticks
(10 ns) between four points of a step:
  gather            issue of the polling trips for dai[t+1] .. the trip that holds no sentinel
  recurrent mfma    the 96 v_mfma_f32_16x16x4_f32 of dai[t+1] W_hh + the store of the partial sums to LDS
  reduce..publish   barrier, (wait for d h_out of the layer above), 4-way sum, gate gradients, the three exchange stores, re-arm
  tail              flush of the previous row's d h_out, packed / row-major copies, next operands, the SECOND product (96 MFMAs)
"""
import os, sys
os.environ["SA_GRU_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from speech_amd import ops, _lib
plain = len(sys.argv) > 1 and sys.argv[1] == "plain"
L, D, B, T, H, I0 = 4, 1, 32, 498, 512, 800
torch.manual_seed(0)
x = torch.randn(T, B, I0, device="cuda")
k = 1.0 / H ** 0.5
w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
b = [torch.zeros(3 * H, device="cuda") for l in range(L)]
h, st = ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
h_out = h
dtop = torch.randn(T, B, H, device="cuda")
wg = None
if not plain:
    wg = (x, [t.contiguous() for t in h_out], [torch.empty_like(w) for w in w_ih], [torch.empty_like(w) for w in w_hh],
          [torch.empty(3 * H, device="cuda") for _ in range(L)], [torch.empty(3 * H, device="cuda") for _ in range(L)])
for _ in range(2):
    ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, D, H, I0, wgrad=wg)
torch.cuda.synchronize()
ws = _lib.WORKSPACE._bufs[(str(x.device), "gru_stack_bwd")]
sync = ws[ws.numel() - 32768: ws.numel()].cpu().numpy().view(np.uint64)
tim = sync[256:256 + 5 * 256].reshape(-1, 5).astype(np.float64)   # sync + 512 uints = 256 u64
steps = T
us = tim[:, :4] * 0.01 / steps
print("form: %s, status %d" % ("sa_gru_stack_bwd (row-major copies)" if plain else "sa_gru_stack_bwd_wgrad (PACKG)", ops.persist_status()))
print("all blocks, us per step: gather %.2f | recurrent mfma %.2f | reduce..publish %.2f | tail %.2f | total %.2f | polling trips per step %.2f"
      % (us[:, 0].mean(), us[:, 1].mean(), us[:, 2].mean(), us[:, 3].mean(), us.sum(1).mean(), tim[:, 4].mean() / steps))
per = tim.reshape(L, -1, 5)  # block index = (job * batch tiles + batch tile) * unit tiles + unit tile; job 0 = the TOP layer
print("per layer, us per step: gather | recurrent mfma | reduce..publish | tail | total | trips")
for j in range(L):
    u = per[j][:, :4] * 0.01 / steps
    print("  layer %d: %.2f | %.2f | %.2f | %.2f | %.2f | %.2f   (gather min/max over its 64 blocks %.2f / %.2f)"
          % (L - 1 - j, u[:, 0].mean(), u[:, 1].mean(), u[:, 2].mean(), u[:, 3].mean(), u.sum(1).mean(), per[j][:, 4].mean() / steps,
             u[:, 0].min(), u[:, 0].max()))
