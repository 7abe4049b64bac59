"""Host-side profile of CTC.infer (cProfile over 200 calls): where the time between two calls' device work goes."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from speech_amd.models import CTC
name = sys.argv[1] if len(sys.argv) > 1 else "timit"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
timit = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2], [32, 5, 32, 1]], "rnn": {"dim": 256, "layers": 4, "bidirectional": True}}}
cfg, freq, vocab, frames = (bench.S_LIBRI, 80, 28, 1000) if name == "slibri" else (timit, 161, 48, 300)
torch.manual_seed(2017)
model = CTC(freq, vocab, cfg).cuda()
model.set_eval()
rng = np.random.RandomState(11)
batch = (tuple(rng.randn(frames, freq).astype(np.float32) for _ in range(bs)), tuple([0, 1] for _ in range(bs)))
for _ in range(5):
    model.infer(batch)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    model.infer(batch)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
