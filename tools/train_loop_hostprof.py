"""Host-side profile (cProfile, main thread) of train.py's loop on HOST batches at S-LIBRI: where the loop's time goes when the
device step is shorter than the host's work per step.   python tools/train_loop_hostprof.py [steps]"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from speech_amd import dist, ops
from speech_amd.models import CTC
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(2017)
model = CTC(bench.F, bench.V, bench.S_LIBRI).cuda()
flat_p, flat_g = model.flatten_parameters_()
model.set_train()
rng = np.random.RandomState(4242)
host = []
for _ in range(4):
    host.append((tuple(rng.randn(bench.T, bench.F).astype(np.float32) for _ in range(bench.B)),
                 tuple(rng.randint(0, bench.V, bench.L).tolist() for _ in range(bench.B))))
norm = torch.zeros(1, device="cuda")
pipe = ops.ScalarPipe()


def loop(n, prof=None):
    def batches():
        for k in range(n):
            yield host[k % 4]
    t0 = time.perf_counter()
    for k, (batch, shape) in enumerate(dist.with_global_shapes(batches(), model)):
        model.set_global_batch(*shape)
        model.zero_grad(set_to_none=True)
        loss = model.loss(batch)
        ops.backward(loss)
        ops.stamp_health(flat_g)
        dist.allreduce_gradients(flat_g)
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
        pipe.push(loss, k, lag=3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


loop(5)
print("loop: %.3f ms per step, ahead hits %s" % (loop(steps), getattr(model, "ahead_hits", None)))
print("loop again: %.3f ms per step" % loop(steps))
pr = cProfile.Profile()
pr.enable()
ms = loop(steps)
pr.disable()
print("profiled loop: %.3f ms per step" % ms)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:7000])
