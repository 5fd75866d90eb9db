#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection CSVs (one or more passes) + the ratios we look at.

  python tools/pmc_summary.py out.json pass1_counter_collection.csv [pass2.csv ...]
Kernel names are shortened to the template name + arguments; instances of one name are summed over all dispatches.
"""
import json
import re
import sys

import pandas as pd


def short(name):
    m = re.match(r"(?:void )?(k_\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def main():
    out, files = sys.argv[1], sys.argv[2:]
    table = {}
    for f in files:
        df = pd.read_csv(f)
        df["k"] = df["Kernel_Name"].map(short)
        disp = df.groupby("k")["Dispatch_Id"].nunique()
        g = df.groupby(["k", "Counter_Name"])["Counter_Value"].sum()
        for (k, c), v in g.items():
            table.setdefault(k, {})[c] = float(v)
        for k, n in disp.items():
            table.setdefault(k, {})["dispatches"] = int(n)
    for k, t in table.items():
        d = {}
        wc = t.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
                if c in t:
                    d[c + "/WAVE_CYCLES"] = round(t[c] / wc, 4)
        if t.get("SQ_INSTS_VALU") and t.get("SQ_THREAD_CYCLES_VALU") and t.get("SQ_ACTIVE_INST_VALU"):
            d["valu_lane_utilisation"] = round(t["SQ_THREAD_CYCLES_VALU"] / (64.0 * t["SQ_ACTIVE_INST_VALU"]), 4)
        if t.get("TCC_HIT_sum") is not None and t.get("TCC_MISS_sum") is not None and t["TCC_HIT_sum"] + t["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 4)
        if "FETCH_SIZE" in t:
            d["hbm_read_MB (FETCH_SIZE x2, gfx950)"] = round(2.0 * t["FETCH_SIZE"] * 1024 / 1e6, 1)
        if "WRITE_SIZE" in t:
            d["hbm_write_MB"] = round(t["WRITE_SIZE"] * 1024 / 1e6, 1)
        t["derived"] = d
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", table[k].get("dispatches", 0))):
        print(k, json.dumps(table[k]["derived"]), {c: ("%.3g" % v) for c, v in table[k].items() if c not in ("derived",) and isinstance(v, float)})


if __name__ == "__main__":
    main()
