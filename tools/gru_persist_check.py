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
        h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=True)
    e1.record(); torch.cuda.synchronize()
    torch.save({"h": [t.cpu() for t in h], "st": [t.cpu() for t in st], "ms": e0.elapsed_time(e1) / 3}, sys.argv[1])
else:
    for shape in [(2, 1, 5, 37, 64, 24), (4, 1, 32, 498, 512, 800), (3, 1, 20, 70, 128, 40)]:
        outs = []
        for mode in ("0", "1"):
            f = "/tmp/persist_%s.pt" % mode
            env = dict(os.environ, SA_GRU_PERSIST=mode)
            r = subprocess.run([sys.executable, __file__, f] + [str(v) for v in shape], env=env, timeout=120)
            outs.append(torch.load(f) if r.returncode == 0 else None)
        a, b = outs
        if a is None or b is None:
            print(shape, "FAILED to run"); continue
        same = all(torch.equal(x, y) for x, y in zip(a["h"], b["h"])) and all(torch.equal(x, y) for x, y in zip(a["st"], b["st"]))
        print(shape, "steps %.3f ms  persistent %.3f ms  bit-identical=%s" % (a["ms"], b["ms"], same))
