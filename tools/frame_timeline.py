#!/usr/bin/env python3
"""Launch by launch timeline of the last frame in a rocprofv3 --kernel-trace directory (the launches
between the last two k_present launches): start, end, duration, queue, grid, kernel, and how long the
machine was idle, ran one kernel, ran two or more.

    python tools/frame_timeline.py gpurun_out/tl_f16p        (see tools/frame_timeline.sh)
"""
import csv, glob, sys, os
d = sys.argv[1]
f = max(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
def short(n):
    n = n.replace("void rtx::", "").replace("rtx::", "")
    return n.split("(")[0][:28]
# last frame only: find the last k_present and go back to the previous one
pres = [i for i, r in enumerate(rows) if "k_present" in r["Kernel_Name"]]
lo = pres[-2] + 1 if len(pres) >= 2 else 0
sel = rows[lo:pres[-1] + 1]
t0 = int(sel[0]["Start_Timestamp"])
print("frame: %d launches, %.3f ms" % (len(sel), (int(sel[-1]["End_Timestamp"]) - t0) / 1e6))
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.3f %9.3f %8.3f q%-3s grid %9s  %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), short(r["Kernel_Name"])))
# machine busy: union of intervals, and time with >=2 kernels
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
n = 0; last = ev[0][0]; t = {0: 0, 1: 0, 2: 0}
for ts, dlt in ev:
    t[min(n, 2)] += ts - last; last = ts; n += dlt
print("idle %.3f ms, one kernel %.3f ms, two or more %.3f ms" % (t[0] / 1e6, t[1] / 1e6, t[2] / 1e6))
