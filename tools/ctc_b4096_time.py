"""M-CTC at B = 4096 (bench.py's ctc_b4096_leg) on its own: ms per call, for one-box A/B of the throughput-regime kernels
(SPEECH_AMD_LIB=... selects another build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for B in (int(a) for a in (sys.argv[1:] or ["4096"])):
    print("B", B, "ms", [round(bench.ctc_b4096_leg(dev, B), 4) for _ in range(3)], flush=True)
