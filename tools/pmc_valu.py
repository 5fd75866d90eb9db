#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ pass of bench.py into per-kernel VALU figures per launch (read by bench.py's `valu` block).

usage: pmc_valu.py <sq_counter_collection.csv> <out.json>
Counters used: SQ_INSTS_VALU (wave-instructions), SQ_ACTIVE_INST_VALU (quad-cycles a VALU instruction was executing, summed over
waves), SQ_THREAD_CYCLES_VALU (lane-cycles), SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE (summed over the 8 XCDs).
"""
import json
import sys

import pandas as pd


def main():
    src, out = sys.argv[1:3]
    df = pd.read_csv(src)
    df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")
    first = df["Kernel_Name"].str.contains(r"k_shade<\d+, true", regex=True)
    df.loc[first, "k"] = "k_shade_first"
    df = df.dropna(subset=["k"])
    res = {}
    for k, g in df.groupby("k"):
        n = g["Dispatch_Id"].nunique()
        c = g.groupby("Counter_Name")["Counter_Value"].sum()
        e = {"launches": int(n)}
        for name in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "SQ_WAVES"):
            if name in c:
                e[name + "_per_launch"] = float(c[name]) / n
        if "SQ_INSTS_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_INSTS_VALU"] > 0:
            e["cycles_per_valu_instr"] = 4.0 * float(c["SQ_ACTIVE_INST_VALU"]) / float(c["SQ_INSTS_VALU"])     # quad-cycles -> cycles
            if "SQ_THREAD_CYCLES_VALU" in c:
                e["lane_utilisation"] = float(c["SQ_THREAD_CYCLES_VALU"]) / (64.0 * float(c["SQ_ACTIVE_INST_VALU"]))
        if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c and c["GRBM_GUI_ACTIVE"] > 0:
            # SIMD-cycles of the launch = 1024 SIMDs x (GRBM_GUI_ACTIVE / 8 XCDs); VALU-busy cycles = 4 x SQ_ACTIVE_INST_VALU
            e["valu_busy_fraction"] = 4.0 * float(c["SQ_ACTIVE_INST_VALU"]) / (1024.0 * float(c["GRBM_GUI_ACTIVE"]) / 8.0)
        res[k] = e
    json.dump({"source": src, "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
