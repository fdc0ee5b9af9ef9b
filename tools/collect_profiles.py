#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of one round from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/collect_profiles.py r01

Expects what tools/profile_all.sh leaves behind: gpurun_out/kt_<walk>/ (kernel trace + stats of
bench.py per walk, and kt_bvh-serial for the default walk with chunks one after the other), gpurun_out/kt_<walk>.json (the bench line printed under the profiler) and
gpurun_out/pmc/summary.json (tools/profile_pmc.sh).  Also rewrites profiles/traffic.json, the HBM
bytes per traversal launch that bench.py reports as roofline.traffic.
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    out = os.path.join(ROOT, "profiles")
    g = os.path.join(ROOT, "gpurun_out")
    for walk in ("bvh", "kd-culled", "reference", "bvh-serial"):
        stats = glob.glob(os.path.join(g, f"kt_{walk}", "*", "*kernel_stats.csv"))
        if stats:       # gpurun merges into what earlier calls left: take the newest
            shutil.copy(max(stats, key=os.path.getmtime), os.path.join(out, f"{tag}_kernel_stats_{walk}.csv"))
        line = os.path.join(g, f"kt_{walk}.json")
        if os.path.exists(line):
            shutil.copy(line, os.path.join(out, f"{tag}_bench_under_rocprof_{walk}.json"))
    summary = os.path.join(g, "pmc", "summary.json")
    if os.path.exists(summary):
        shutil.copy(summary, os.path.join(out, f"{tag}_pmc_summary_bvh.json"))
        s = json.load(open(summary))
        total, launches = 0.0, 0
        for name, e in s.items():
            if "k_trace_pool<false" not in name or "FETCH_SIZE" not in e or "WRITE_SIZE" not in e:
                continue
            n = e["FETCH_SIZE"]["launches"]
            total += n * (2.0 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
            launches += n
        if launches:
            json.dump({"hbm_bytes_per_launch": total / launches,
                       "note": "mean over the primary (MODE 3) and shadow (MODE 4) k_trace_pool launches of "
                               "(2*FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes "
                               "(tools/profile_pmc.sh); FETCH_SIZE is in KiB and is doubled as gfx950 tallies the "
                               "128-byte requests of 16-byte-per-lane loads at 64 bytes; the loads here are gathers, "
                               "for which the doubling is uncalibrated, so this is an upper estimate",
                       "source": f"profiles/{tag}_pmc_summary_bvh.json"},
                      open(os.path.join(out, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
