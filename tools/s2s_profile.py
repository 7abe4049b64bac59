"""Profiling target: Seq2Seq train steps / decode at BASELINE config 4 shapes (see tools/bench_configs.m_seq2seq)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc  # noqa: E402

print(json.dumps(bc.m_seq2seq(10)))
