#!/usr/bin/env python3
"""Per-level traversal work of the bench's 16-spp pathTrace frame (run on the MI355X box): renders it with
max_depth = 0 .. 8 under RTX_FLAG_COUNTERS and differences the counters, one JSON line per level:
rays, steps per ray (node visits + shape tests), the fraction of lanes busy while a wave walks, and
wave-steps (sum over waves of the longest lane's steps).

    python tools/level_work.py
"""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, device, scenes
W, H = 1920, 1080
device.init(0)
scene = scenes.frame16_scene("path")
desc = scene.frame(0.0)
ds = device.DeviceScene(desc, stream=torch.cuda.current_stream().cuda_stream)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
prev = None
for md in range(0, 9):
    rd = scenes.frame16_render_desc("path", W, H, 1, flags=abi.FLAG_COUNTERS | abi.FLAG_OVERWRITE)
    rd.max_depth = md
    st = ds.render_device(rd, fb.data_ptr())
    cur = {k: getattr(st, k) for k in ("rays", "lane_steps", "wave_steps", "content_tests", "triangle_tests", "aabb_tests")}
    d = cur if prev is None else {k: cur[k] - prev[k] for k in cur}
    if d["rays"]:
        print(json.dumps({"level": md, "rays": d["rays"], "steps_per_ray": round(d["lane_steps"] / d["rays"], 1),
                          "lanes_busy": round(d["lane_steps"] / max(1, d["wave_steps"]), 3),
                          "wave_steps_M": round(d["wave_steps"] / 64 / 1e6, 2),
                          "node_steps_per_ray": round(d["content_tests"] / 2 / d["rays"], 1),
                          "tri_tests_per_ray": round(d["triangle_tests"] / d["rays"], 1)}), flush=True)
    prev = cur
