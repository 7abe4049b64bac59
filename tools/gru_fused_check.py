"""A/B of the fused forward layer wavefront (the default: one launch, in-kernel input projections) against the chunked
path (SA_GRU_FUSED=0: persistent chunk kernels + per-chunk projection GEMMs); the backward columns time the same chunked
backward on both sides (a fused backward was measured slower and dropped).  Not bit-identical by construction (the projection is
summed in a different order): compared to a tolerance, and timed."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 2:
    from speech_amd import ops, _lib
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
    assert _lib.lib().sa_gru_persist_status() == 0, "persistent kernels reported an error"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=True)
    e1.record(); torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / 3
    dtop = torch.randn(T, B, H, device="cuda")
    for _ in range(2):
        dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, D, H, I0, want_dx=True)
    torch.cuda.synchronize()
    assert _lib.lib().sa_gru_persist_status() == 0, "persistent kernels reported an error (backward)"
    e0.record()
    for _ in range(3):
        dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, D, H, I0, want_dx=True)
    e1.record(); torch.cuda.synchronize()
    torch.save({"h": [t.cpu() for t in h], "st": [t.cpu() for t in st], "ms": fwd_ms, "bwd_ms": e0.elapsed_time(e1) / 3,
                "g": [t.cpu() for t in dai] + [t.cpu() for t in dah] + [dx.cpu()]}, sys.argv[1])
else:
    shapes = [(2, 1, 20, 9, 512, 24), (4, 1, 32, 60, 512, 48), (4, 1, 32, 498, 512, 800)]
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        shapes = shapes[:2]
    for shape in shapes:
        outs = []
        for fused in ("0", "1"):
            f = "/tmp/fused_%s.pt" % fused
            env = dict(os.environ, SA_GRU_FUSED=fused)
            r = subprocess.run([sys.executable, __file__, f] + [str(v) for v in shape], env=env, timeout=100)
            outs.append(torch.load(f) if r.returncode == 0 else None)
        a, b = outs
        if a is None or b is None:
            print(shape, "FAILED to run"); continue
        worst = 0.0
        for x, y in zip(a["h"] + a["st"], b["h"] + b["st"]):
            worst = max(worst, float(((x - y).abs() / (1e-3 + x.abs())).max()))
        gworst = 0.0
        for x, y in zip(a["g"], b["g"]):
            gworst = max(gworst, float((x - y).abs().max() / (1e-6 + x.abs().max())))
        print(shape, "fwd: default %.3f ms  fused %.3f ms  max rel diff %.2e | bwd: default %.3f ms  fused %.3f ms  "
              "max diff / max %.2e  finite=%s" % (a["ms"], b["ms"], worst, a["bwd_ms"], b["bwd_ms"], gworst,
                                                  all(torch.isfinite(t).all() for t in b["g"])))
