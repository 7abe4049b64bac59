"""Profiling target: train steps on the shipped examples/timit/ctc_config.json shapes (F=161, V+1=49, B=8, T=300,
conv [[32,5,32,2],[32,5,32,1]], 4 x bidirectional GRU-256)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc  # noqa: E402

timit = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2], [32, 5, 32, 1]],
                                     "rnn": {"dim": 256, "layers": 4, "bidirectional": True}}}
print(bc.m_step("timit", 161, 48, timit, int(os.environ.get("B", 8)), 300, 40, 5))
