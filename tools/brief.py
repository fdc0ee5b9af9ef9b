#!/usr/bin/env python3
"""Reduce bench.py's JSON line (stdin) to the figures an A/B comparison looks at."""
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    o = {}
    if d.get("value") is not None: o["value"] = d["value"]; o["ms_per_step"] = d.get("ms_per_step"); o["spread"] = d.get("spread") or d.get("value_min_median")
    f = d.get("frame_16spp") or {}
    for k, v in f.items():
        if isinstance(v, dict) and "ms_per_frame" in v:
            o[k] = [v["ms_per_frame"], v.get("ms_per_frame_min_median"), (v.get("pipelined") or {}).get("ms_per_frame")]
    print(json.dumps(o))
