#!/usr/bin/env python3
"""How the two 1080p 16-spp frames of bench.py split over N ranks, rehearsed on ONE GPU (no RCCL).

  shares    one rank's share of ONE frame in an N-rank world -- the interleaved groups of 8 rows
            bench.py --gpus N and dist.render_frame_rows deal out, one tile, one call -- every rank's
            share rendered in turn: the slowest rank sets the frame time, so (N = 1) / slowest is how
            far a single frame splits over N GPUs before any gather.
  sequence  the frames of the sequence dealt out whole, frame f to rank f % N (bench.py's `sequence`
            leg): a rank renders whole frames with F of them in flight, so N ranks deliver N times a
            rank's rate, less what rank 0 spends taking in the other ranks' presented frames.  Printed:
            ms per frame of one rank for F = 1 .. 8 (one stream per frame in flight, chunks one after
            the other), and the same with seven device-to-device copies of a presented frame per own
            frame queued beside the renders -- the stand-in for what rank 0 of eight receives and places.

    python tools/frame_shares.py [whitted] [path] [--no-shares] [--no-sequence]
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


def shares(kind, ds):
    base = None
    for world in (1, 2, 4, 8):
        worst = 0.0
        for rank in range(world):
            y0, rows, stride = rdist.rank_rows(H, world, rank, 8)
            buf = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")

            def frame(f):
                rd = scenes.frame16_render_desc(kind, W, H, f, flags=abi.FLAG_OVERWRITE, tile=(0, y0, W, rows), row_group=8, row_stride=stride)
                ds.render_device(rd, buf.data_ptr())
            frame(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(1, 6):
                frame(f)
            torch.cuda.synchronize()
            worst = max(worst, (time.perf_counter() - t0) / 5 * 1e3)
        base = base or worst
        print(json.dumps({"integrator": kind, "partition": "rows of one frame", "ranks": world, "slowest_share_ms": round(worst, 3),
                          "speedup": round(base / worst, 2)}), flush=True)


def sequence(kind, ds, frames=12):
    tonemap = abi.TONEMAP_CBRT if kind == "path" else abi.TONEMAP_NONE
    other = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")        # a presented frame of another rank
    inbox = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(7)]
    for F in (1, 2, 3, 4, 6, 8):
        for receive in (False, True):
            streams = [torch.cuda.Stream() for _ in range(F)]
            copy_stream = torch.cuda.Stream()
            handles = [ds.clone(stream=st.cuda_stream) for st in streams]
            for h in handles:
                h.set_async(True)
            fbs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(F)]
            shown = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(F)]

            def step(f):
                k = f % F
                with torch.cuda.stream(streams[k]):
                    rd = scenes.frame16_render_desc(kind, W, H, f, flags=abi.FLAG_OVERWRITE | abi.FLAG_SERIAL_CHUNKS)
                    handles[k].render_device(rd, fbs[k].data_ptr())
                    handles[k].present_device(fbs[k].data_ptr(), W, H, shown[k].data_ptr(), passes=1, tonemap=tonemap, flip=True)
                if receive:
                    with torch.cuda.stream(copy_stream):
                        for b in inbox:
                            b.copy_(other, non_blocking=True)
            for f in range(F):
                step(f)
            for h in handles:
                h.wait()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(F, F + frames):
                step(f)
            for h in handles:
                h.wait()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / frames * 1e3
            for h in handles:
                h.close()
            print(json.dumps({"integrator": kind, "partition": "whole frames", "frames_in_flight": F,
                              "with_seven_frames_received_per_own_frame": receive, "ms_per_frame_of_one_rank": round(ms, 3),
                              "hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (HIP default)")}), flush=True)


def main(argv):
    kinds = [a for a in argv if not a.startswith("--")] or ["whitted", "path"]
    device.init(0)
    for kind in kinds:
        scene = scenes.frame16_scene(kind)
        ds = device.DeviceScene(scene.frame(0.0), stream=torch.cuda.current_stream().cuda_stream, static=True)
        if "--no-shares" not in argv:
            shares(kind, ds)
        if "--no-sequence" not in argv:
            sequence(kind, ds)
        ds.close()


if __name__ == "__main__":
    main(sys.argv[1:])
