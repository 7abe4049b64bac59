"""Profiling target: train steps of the bidirectional S-LIBRI variant (SURVEY 8d M-STEP: 4 x biGRU-512, 18.2 M params)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc  # noqa: E402

bi = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": True}}}
print(bc.m_step("S-LIBRI bidirectional", 80, 28, bi, 32, 1000, 100, 4))
