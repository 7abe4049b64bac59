"""M-DEC: prefix beam search on softmax(4 randn) (32, 498, 29), blank 28 -- kernel-only time (HIP events
around the library call) and the time of the whole decoder.beam_decode call (incl. the copy back and the Python lists).
    python tools/decode_bench.py [beam ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speech_amd import _lib, decoder  # noqa: E402

rng = np.random.RandomState(2017)
z = torch.from_numpy((4.0 * rng.randn(32, 498, 29)).astype(np.float32)).cuda()
probs = torch.softmax(z, dim=2)
for beam in [int(a) for a in sys.argv[1:]] or [1, 8]:
    x, lens, B, T, S = decoder._prep(probs, None)
    L = _lib.lib()
    out_labels = torch.empty(B, T, dtype=torch.int32, device=x.device)
    out_lens = torch.empty(B, dtype=torch.int32, device=x.device)
    nll = torch.empty(B, dtype=torch.float32, device=x.device)
    ws = _lib.WORKSPACE.get(L.sa_ctc_beam_workspace_bytes(T, S, B, beam), x.device, "beam")

    def call():
        _lib.check(L.sa_ctc_beam_decode(_lib.ptr(x), x.stride(1), x.stride(0), _lib.ptr(lens), S, B, T, beam, 28, 0,
                                        _lib.ptr(out_labels), _lib.ptr(out_lens), _lib.ptr(nll), _lib.ptr(ws),
                                        ws.numel(), _lib.cur_stream()), "sa_ctc_beam_decode")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        decoder.beam_decode(probs, beam_size=beam, blank=28)
    torch.cuda.synchronize()
    all_ms = (time.perf_counter() - t0) / 10 * 1e3
    print("beam %d: device %.3f ms (%.2f us per frame), whole call %.3f ms" % (beam, dev_ms, dev_ms * 1e3 / T, all_ms))
