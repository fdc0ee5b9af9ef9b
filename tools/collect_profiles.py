#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of one round from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/collect_profiles.py r02

Expects what tools/profile_all.sh leaves behind: gpurun_out/kt_<name>/ (kernel trace + stats of
bench.py for name in main, main-serial, f16w, f16p), gpurun_out/kt_<name>.json (the bench line printed
under the profiler) and gpurun_out/pmc_<workload>/summary.json + bench_args.txt (tools/profile_pmc.sh).
Also rewrites profiles/counters.json, from which bench.py takes the counter-based entries of its
roofline blocks: per workload and kernel the mean counter values per launch and the launches per frame.
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES",
        "SQ_WAIT_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE", "TCC_REQ_sum",
        "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum")


def frames_of(bench_args):
    a = bench_args.split()
    def val(flag, default):
        return int(a[a.index(flag) + 1]) if flag in a else default
    only = a[a.index("--only") + 1]
    if only == "main":
        return val("--steps", 60)
    if only == "c5":
        return val("--c5-frames", 1)
    return val("--frames16", 10)


def main(tag):
    out = os.path.join(ROOT, "profiles")
    g = os.path.join(ROOT, "gpurun_out")
    for name in ("main", "main-serial", "f16w", "f16p", "f16w-serial", "f16p-serial"):
        stats = glob.glob(os.path.join(g, f"kt_{name}", "*", "*kernel_stats.csv"))
        if stats:       # gpurun merges into what earlier calls left: take the newest
            shutil.copy(max(stats, key=os.path.getmtime), os.path.join(out, f"{tag}_kernel_stats_{name}.csv"))
        line = os.path.join(g, f"kt_{name}.json")
        if os.path.exists(line):
            shutil.copy(line, os.path.join(out, f"{tag}_bench_under_rocprof_{name}.json"))
    workloads = {}
    for wl in ("main", "f16w", "f16p", "c5"):
        summary = os.path.join(g, f"pmc_{wl}", "summary.json")
        args_file = os.path.join(g, f"pmc_{wl}", "bench_args.txt")
        if not (os.path.exists(summary) and os.path.exists(args_file)):
            continue
        shutil.copy(summary, os.path.join(out, f"{tag}_pmc_summary_{wl}.json"))
        bench_args = open(args_file).read().strip()
        frames = frames_of(bench_args)
        s = json.load(open(summary))
        kernels = {}
        for name, e in s.items():
            if not name.startswith("rtx::"):
                continue
            launches = max((v["launches"] for v in e.values() if isinstance(v, dict)), default=0)
            k = {"launches_per_frame": launches / frames}
            for c in KEEP:
                if c in e:
                    k[c] = e[c]["mean"]
            kernels[name.replace("rtx::", "")] = k
        workloads[wl] = {"bench_args": bench_args, "frames": frames, "kernels": kernels}
    if workloads:
        json.dump({"source": f"profiles/{tag}_pmc_summary_<workload>.json: rocprofv3 --pmc passes of `bench.py <bench_args>` "
                             "(tools/profile_pmc.sh, one pass per counter group, kernels serialised by the profiler); "
                             "FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE to be doubled on gfx950",
                   "workloads": workloads}, open(os.path.join(out, "counters.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
