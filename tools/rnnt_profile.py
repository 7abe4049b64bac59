"""Profiling target: RNN-Transducer train steps at BASELINE config 5 shapes (see tools/bench_configs.m_transducer)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc  # noqa: E402

print(json.dumps(bc.m_transducer(10)))
