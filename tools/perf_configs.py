#!/usr/bin/env python3
"""The `other_configs` block of bench.py on its own (BASELINE configs[3] = cfg4 and one GPU's share of configs[4] = cfg5):
Msamples/s, per-kernel-class ms, visit counters.  `python tools/perf_configs.py [cfg4|cfg5]`"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

only = sys.argv[1] if len(sys.argv) > 1 else ""
res = bench.other_configs(bench.load_pkg(), 0, only=only)
for k, v in res.items():
    print(k, json.dumps(v))
