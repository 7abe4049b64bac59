"""Profiling target: NNAttention forward + backward kernels at the shipped WSJ Seq2Seq shapes (B=16, T'=197, H=256,
KS=15), 200 calls each, for `rocprofv3 --kernel-trace --stats`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_amd import _lib  # noqa: E402

B, T, H, KS = 16, 197, 256, 15
dev = "cuda"
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
eh, ox, conv_w, conv_b, nn_w, nn_b = r(B, T, H), r(B, H), r(H, KS) * 0.1, r(H) * 0.1, r(H) * 0.1, r(1)
ax_prev = torch.softmax(r(B, T), dim=1)
ax, sx = torch.empty(B, T, device=dev), torch.empty(B, H, device=dev)
L = _lib.lib()
ws = torch.empty(L.sa_attention_workspace_bytes(B, T, H, KS), dtype=torch.uint8, device=dev)
d_sx, d_axn = r(B, H), r(B, T) * 0.01
d_eh, d_ox, d_axp = torch.zeros(B, T, H, device=dev), torch.empty(B, H, device=dev), torch.empty(B, T, device=dev)
g_cw, g_cb, g_nw, g_nb = (torch.zeros(B, H, KS, device=dev), torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev),
                          torch.zeros(B, device=dev))
p = _lib.ptr
st = _lib.cur_stream()
for _ in range(200):
    _lib.check(L.sa_attention_fwd(p(eh), p(ox), p(ax_prev), p(conv_w), p(conv_b), p(nn_w), p(nn_b), 1.0, p(ax), p(sx),
                                  B, T, H, KS, p(ws), ws.numel(), st), "fwd")
    _lib.check(L.sa_attention_bwd(p(eh), p(ox), p(ax_prev), p(conv_w), p(conv_b), p(nn_w), p(nn_b), 1.0, p(ax), p(d_sx),
                                  p(d_axn), p(d_eh), p(d_ox), p(d_axp), p(g_cw), p(g_cb), p(g_nw), p(g_nb), B, T, H, KS,
                                  p(ws), ws.numel(), st), "bwd")
torch.cuda.synchronize()
print("ok")
