"""A/B of forward-recurrence variants in ONE process: stack-forward time (HIP events over 10 calls, S-LIBRI shapes) per
(option settings); and optionally the phase clocks.   python tools/gru_fwd_variants.py "L" "name=opt:val,opt:val" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from speech_amd import ops, _lib
_lib.lib()
Ls = [int(v) for v in sys.argv[1].split(",")]
variants = []
for a in sys.argv[2:]:
    name, spec = a.split("=")
    variants.append((name, [(kv.split(":")[0], int(kv.split(":")[1])) for kv in spec.split(",") if kv]))
D, B, T, H, I0 = 1, 32, 498, 512, 800
for L in Ls:
    torch.manual_seed(0)
    x = torch.randn(T, B, I0, device="cuda")
    k = 1.0 / H ** 0.5
    w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
    w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
    b = [torch.zeros(3 * H, device="cuda") for l in range(L)]
    for rnd in range(2):
        for name, opts in variants:
            defaults = _lib.option_defaults()
            for o, v in defaults.items():
                _lib.set_option(o, v)
            for o, v in opts:
                _lib.set_option(o, v)
            for _ in range(3):
                ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            line = "L=%d %-22s %.3f ms (%.2f us per step incl. projection + fills) status %d" % (L, name, ms, ms * 1e3 / T, ops.persist_status())
            if dict(opts).get("gru.timing"):
                ws = _lib.WORKSPACE._bufs[(str(x.device), "gru_stack")]
                sync = ws[ws.numel() - 32768: ws.numel()].cpu().numpy().view(np.uint64)
                tim = sync[128:128 + 8 * 64 * L].reshape(L, 2, 32, 8).astype(np.float64)
                print(line)
                for l in range(L):
                    m = tim[l].reshape(-1, 8).mean(0)
                    ph = m[:4] * 0.01 / T
                    print("    layer %d: input %.2f (part 2 %.2f, poll issue %.2f)  poll %.2f  recurrent mfma %.2f  reduce+gates %.2f | total %.2f us/step | clock %.0f MHz"
                          % (l, ph[0], m[7] * 0.01 / T, m[6] * 0.01 / T, ph[1], ph[2], ph[3], ph.sum() + (m[6] + m[7]) * 0.01 / T, m[4] / (m[5] * 0.01)))
            else:
                print(line)
