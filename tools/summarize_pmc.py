#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel and counter, the mean value per launch.

    python3 tools/summarize_pmc.py <dir with */**/*counter_collection.csv>  > summary.json

A launch's value is the sum of the rows rocprofv3 wrote for that dispatch (one row per
counter instance/dimension).  Derived entries: l2_hit_rate = TCC_HIT/(TCC_HIT+TCC_MISS);
hbm_read_bytes_raw = FETCH_SIZE KiB * 1024; hbm_write_bytes = WRITE_SIZE KiB * 1024.
"""
import csv
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def main(root):
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> value
    for d, _, files in os.walk(root):
        for f in files:
            if not f.endswith("counter_collection.csv"):
                continue
            with open(os.path.join(d, f), newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    per[k][row["Counter_Name"]][(d, row["Dispatch_Id"])] += float(row["Counter_Value"])
    out = {}
    for k, counters in per.items():
        e = {}
        for c, disp in counters.items():
            vals = list(disp.values())
            e[c] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
        hit, miss = e.get("TCC_HIT_sum"), e.get("TCC_MISS_sum")
        if hit and miss and hit["mean"] + miss["mean"] > 0:
            e["l2_hit_rate"] = hit["mean"] / (hit["mean"] + miss["mean"])
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_raw"] = e["FETCH_SIZE"]["mean"] * 1024
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = e["WRITE_SIZE"]["mean"] * 1024
        out[k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
