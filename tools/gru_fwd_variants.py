"""A/B of the fused forward recurrence inside ONE process / one box: how often a layer reports its progress to the layer
above (library option gru.fwd_report = 1 / 2 / 4 / 8 / 16 steps; 4 is the default), round robin, HIP events over 10 stack-forward calls
each; outputs compared with the default's bit for bit.   python tools/gru_fwd_variants.py [L] [rounds]
(The round-5 sweep that also covered the kernel of rounds 1-4 and the position of the first polling trip is recorded in
profiles/r05_forward_recurrence_experiments.txt.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_amd import ops, _lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
D, B, T, H, I0 = 1, 32, 498, 512, 800
torch.manual_seed(0)
x = torch.randn(T, B, I0, device="cuda")
k = 1.0 / H ** 0.5
w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
b = [torch.empty(3 * H, device="cuda").uniform_(-k, k) for l in range(L)]
VARIANTS = [("report%d" % r, {"gru.fwd_report": r}) for r in (4, 1, 2, 8, 16)]

def run(env, n):
    for name, value in env.items():
        _lib.set_option(name, value)
    out = None
    for _ in range(2):
        out = ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gru_stack_fwd(x, w_ih, b, w_hh, b, L, D, H, want_stash=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


ref = None
times = {name: [] for name, _ in VARIANTS}
for r in range(ROUNDS):
    for name, env in VARIANTS:
        ms, (h, st) = run(env, 10)
        times[name].append(ms)
        if ref is None:
            ref = ([t.clone() for t in h], [t.clone() for t in st])
        elif r == 0:
            dh = max(float((a - c).abs().max()) for a, c in zip(h, ref[0]))
            ds = max(float((a - c).abs().max()) for a, c in zip(st, ref[1]))
            print("%-16s max|h - h_ref| = %.3g  max|stash - stash_ref| = %.3g  status %d" % (name, dh, ds, ops.persist_status()), flush=True)
for name, _ in VARIANTS:
    v = times[name]
    print("L=%d %-16s %s  min %.3f ms  (%.2f us per time step incl. the layer-0 projection and fills)"
          % (L, name, " ".join("%.3f" % t for t in v), min(v), min(v) * 1e3 / (T + L - 1)), flush=True)
