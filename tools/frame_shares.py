#!/usr/bin/env python3
"""One rank's share of ONE 1080p 16-spp frame in an N-rank world, every rank's share rendered in turn
on one GPU (no gather): the slowest rank sets the frame time, so slowest / (N = 1) is how far a
single frame splits over N GPUs.  Shares are the interleaved groups of 8 rows bench.py --gpus N and
dist.render_frame_rows deal out, one tile per rank, one call.

    python tools/frame_shares.py [whitted] [path]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raytracer_amd import abi, device, scenes
from raytracer_amd import dist as rdist

W, H = 1920, 1080


def main(kinds):
    device.init(0)
    for kind in kinds:
        scene, _ = scenes.mesh_scene(ggx=True, square_light=True, mesh="backpack")
        if kind == "path":
            scene.sphere((0.0, 10.0, 0.0), 2.0, scene.material("PlainWhiteTees", emission=(8.0, 8.0, 8.0)))
        ds = device.DeviceScene(scene.frame(0.0), stream=torch.cuda.current_stream().cuda_stream)
        base = None
        for world in (1, 2, 4, 8):
            worst = 0.0
            for rank in range(world):
                y0, rows, stride = rdist.rank_rows(H, world, rank, 8)
                buf = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")

                def frame(f):
                    cam = scenes.orbit_camera(f / 30.0)
                    common = dict(camera=cam, n=4, frame=f, tile=(0, y0, W, rows), row_group=8, row_stride=stride, flags=abi.FLAG_OVERWRITE)
                    if kind == "whitted":
                        rd = scenes.render_desc(W, H, integrator=abi.INTEGRATOR_WHITTED, light_n=2, max_depth=3, bias=1e-2, **common)
                    else:
                        rd = scenes.render_desc(W, H, integrator=abi.INTEGRATOR_PATH, max_depth=8, bias=1e-3, **common)
                    ds.render_device(rd, buf.data_ptr())
                frame(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for f in range(1, 6):
                    frame(f)
                torch.cuda.synchronize()
                worst = max(worst, (time.perf_counter() - t0) / 5 * 1e3)
            base = base or worst
            print(json.dumps({"integrator": kind, "ranks": world, "slowest_share_ms": round(worst, 3), "speedup": round(base / worst, 2)}), flush=True)
        ds.close()


if __name__ == "__main__":
    main(sys.argv[1:] or ["whitted", "path"])
