"""A/B of the persistent forward chunk kernel against the one-launch-per-step path (must be bit-identical)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 1:
    from speech_amd import ops
    L, D, B, T, H, I0 = [int(v) for v in sys.argv[2:8]]
    torch.manual_seed(0)
    x = torch.randn(T, B, I0, device="cuda")
    k = 1.0 / H ** 0.5
    w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
    w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
    b_ih = [torch.empty(3 * H, device="cuda").uniform_(-k, k) for l in range(L)]
    b_hh = [torch.empty(3 * H, device="cuda").uniform_(-k, k) for l in range(L)]
    for _ in range(2):
        h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=True, chunk=28)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=True, chunk=28)
    e1.record(); torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / 3
    dtop = torch.randn(T, B, H, device="cuda")
    for _ in range(2):
        dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, D, H, I0, want_dx=True, chunk=28)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, D, H, I0, want_dx=True, chunk=28)
    e1.record(); torch.cuda.synchronize()
    torch.save({"h": [t.cpu() for t in h], "st": [t.cpu() for t in st] + [t.cpu() for t in dai] + [t.cpu() for t in dah] +
                [dx.cpu()], "ms": fwd_ms, "bwd_ms": e0.elapsed_time(e1) / 3}, sys.argv[1])
else:
    for shape in [(4, 1, 32, 498, 512, 800)]:
        outs = []
        for mode in ("0", "2", "3"):   # step kernels, XCD-local persistent groups, + flag-less hand-off
            f = "/tmp/persist_%s.pt" % mode
            env = dict(os.environ, SA_GRU_PERSIST=mode)
            r = subprocess.run([sys.executable, __file__, f] + [str(v) for v in shape], env=env, timeout=120)
            outs.append(torch.load(f) if r.returncode == 0 else None)
        a = outs[0]
        for name, b in (("xcd-local", outs[1]), ("xcd-flagless", outs[2])):
            if a is None or b is None:
                print(shape, name, "FAILED to run"); continue
            same = all(torch.equal(x, y) for x, y in zip(a["h"], b["h"])) and all(torch.equal(x, y) for x, y in zip(a["st"], b["st"]))
            if not same:
                names = ["h%d" % i for i in range(len(a["h"]))] + ["st%d" % i for i in range(len(a["h"]))] + \
                        ["dai%d" % i for i in range(len(a["h"]))] + ["dah%d" % i for i in range(len(a["h"]))] + ["dx"]
                for nm, x, y in zip(names, list(a["h"]) + list(a["st"]), list(b["h"]) + list(b["st"])):
                    if not torch.equal(x, y):
                        d = (x - y).abs()
                        bad = (d > 0).reshape(d.shape[0], -1).any(dim=1).nonzero().flatten()
                        print("   ", nm, "max|diff| %.3e  max|ref| %.3e  differing t: first %d last %d count %d" %
                              (float(d.max()), float(x.abs().max()), int(bad[0]), int(bad[-1]), bad.numel()))
            print(shape, "fwd: steps %.3f ms  %s %.3f ms | bwd: steps %.3f ms  %s %.3f ms | bit-identical (h, stash, dai, "
                  "dah, dx)=%s" % (a["ms"], name, b["ms"], a["bwd_ms"], name, b["bwd_ms"], same))
