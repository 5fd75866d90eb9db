#!/usr/bin/env python3
"""Dispatch timeline of the last render in a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous
dispatch.  usage: python3 tools/timeline.py <..._kernel_trace.csv> [n_last=20]"""
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
df = df.sort_values("Start_Timestamp").tail(n)
t0 = df["Start_Timestamp"].iloc[0]
prev_end = None
for _, r in df.iterrows():
    gap = 0.0 if prev_end is None else (r["Start_Timestamp"] - prev_end) / 1e3
    print("%9.1f us  dur %8.1f us  gap %7.1f us  %s" % ((r["Start_Timestamp"] - t0) / 1e3, (r["End_Timestamp"] - r["Start_Timestamp"]) / 1e3, gap, r["Kernel_Name"][:80]))
    prev_end = max(prev_end or 0, r["End_Timestamp"])
