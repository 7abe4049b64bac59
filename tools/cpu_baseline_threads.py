import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle.torch_ref import TorchRefCTC, train_step
x_h, lab_h = bench.synthetic(0)
x = torch.from_numpy(x_h); ll = np.full(bench.B, bench.L, np.int32)
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads())
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    torch.manual_seed(2017)
    model = TorchRefCTC(bench.F, bench.V, bench.S_LIBRI)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    train_step(model, opt, x, lab_h, ll, threads=th)
    t0 = time.perf_counter(); train_step(model, opt, x, lab_h, ll, threads=th); dt = time.perf_counter() - t0
    print("threads %3d: %.2f s/step  %.2f utt/s" % (th, dt, bench.B / dt), flush=True)
