#!/usr/bin/env python3
"""Throughput benchmark of the trace loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): a 60 026-triangle mesh scene (seeded
procedural stand-in for the reference's FBX meshes) through the reference's KD
tree, Whitted integrator with one DirectionalLight: primary + hard shadow rays,
1920x1080, 1 sample per pixel.  A step is one frame of an orbiting camera
(t = frame/30): rtx_render_device into a float framebuffer resident in HBM,
then rtx_present_device (what the reference does between drawBuffer and the PNG:
8-bit sRGB, flipped).  The frames of the sequence are independent, so a rank keeps
--in-flight of them (default 4) queued on as many streams, each through its own handle
on the one device copy of the scene (rtx_scene_clone) with its own framebuffers: the
tail of one frame's launches is filled by the next frames' kernels.  With N > 1 ranks every rank renders and presents its own
frame of the sequence (frames are independent given the read-only scene) and
the N presented frames are gathered to rank 0 with RCCL inside the timed
region, as a frame driver that writes the PNGs on rank 0 needs them; per-GPU
work is fixed, so scaling is weak.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline      algorithmic bytes of the traversal kernel / its measured duration
  cpu_baseline  the CPU oracle timed on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WIDTH, HEIGHT = 1920, 1080
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6


def frame_time(step, rank, world):
    return (step * world + rank) / 30.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--in-flight", type=int, default=4,
                    help="frames of the sequence queued at once, each on a stream of its own (1: one frame at a time, "
                         "its two half-frame chunks on the library's two streams)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame16", action="store_true", help="skip the 16-spp soft-shadow frame timing")
    ap.add_argument("--serial-chunks", action="store_true",
                    help="render a frame's two half-frame chunks one after the other instead of on two streams: "
                         "launches then run alone, which is how the per-launch figures of the roofline are best read")
    ap.add_argument("--store-camera-rays", action="store_true",
                    help="write the camera rays with a generate kernel and read them back (RTX_FLAG_STORE_CAMERA_RAYS) "
                         "instead of computing them in the kernels of level 0; same image, for comparison")
    ap.add_argument("--accumulate-samples", action="store_true",
                    help="clear the sample slots, add every term and resolve (RTX_FLAG_ACCUMULATE_SAMPLES) instead of "
                         "storing finished samples; same image, for comparison")
    ap.add_argument("--store-light-rays", action="store_true",
                    help="store every light ray's direction and light distance (RTX_FLAG_STORE_LIGHT_RAYS) instead of "
                         "recomputing them from the origin; same image, for comparison")
    ap.add_argument("--no-serial-leg", action="store_true",
                    help="skip the untimed repeat of the frames with chunks one after the other (roofline.serial)")
    ap.add_argument("--walk", choices=["bvh", "kd-culled", "reference"], default="bvh",
                    help="bvh: candidates from the library's BVH, verified against the reference's KD walk "
                         "(default render path); kd-culled: the reference's KD tree walked with culling; "
                         "reference: node for node as the reference's traverseKDTree.  All three give "
                         "identical images.")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, device, scenes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    # BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 path on a
    # one-GPU box (the gather then goes through host memory); the default is RCCL.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    device.init(dev_index)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")

    W, H = args.width, args.height
    scene, _ = scenes.mesh_scene()
    desc = scene.frame(0.0)
    kd = scene.kd_stats()
    kflags = {"bvh": 0, "kd-culled": abi.FLAG_KD_CULLED, "reference": abi.FLAG_REFERENCE_ORDER}[args.walk]
    F = max(1, args.in_flight)
    if args.serial_chunks or F > 1:     # frames in flight overlap each other; their chunks need not
        kflags |= abi.FLAG_SERIAL_CHUNKS
    if args.store_camera_rays:
        kflags |= abi.FLAG_STORE_CAMERA_RAYS
    if args.accumulate_samples:
        kflags |= abi.FLAG_ACCUMULATE_SAMPLES
    if args.store_light_rays:
        kflags |= abi.FLAG_STORE_LIGHT_RAYS
    stream = torch.cuda.current_stream().cuda_stream
    ds = device.DeviceScene(desc, stream=stream)
    # handle k renders the frames with step % F == k on stream k
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(F - 1)]
    handles = [ds] + [ds.clone(stream=st.cuda_stream) for st in streams[1:]]

    def rd_for(step, flags):
        cam = scenes.orbit_camera(frame_time(step, rank, world))
        return scenes.render_desc(W, H, integrator=abi.INTEGRATOR_WHITTED, camera=cam, n=1, max_depth=3, bias=1e-2,
                                  clear=(0.21, 0.78, 0.95), flags=flags | kflags)

    # per handle: a float framebuffer and two presented frames (the gather of one, on RCCL's own
    # stream, overlaps the handle's next render)
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    fbs = [fb] + [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(F - 1)]
    shown = [torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2 * F)]
    gathered = [[torch.zeros((H, W, 3), dtype=torch.uint8, device=comm_dev) for _ in range(world)]
                if (world > 1 and rank == 0) else None for _ in range(2 * F)]
    pending = [None] * (2 * F)

    def step_fn(step, flags):
        h = step % F
        k = 2 * h + (step // F) % 2
        with torch.cuda.stream(streams[h]):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
            # the frame replaces what the buffer held (the reference clears it per frame)
            st = handles[h].render_device(rd_for(step, flags | abi.FLAG_OVERWRITE), fbs[h].data_ptr())
            handles[h].present_device(fbs[h].data_ptr(), W, H, shown[k].data_ptr(), passes=1, tonemap=abi.TONEMAP_NONE, flip=True)
            if world > 1:
                src = shown[k] if backend == "nccl" else shown[k].cpu()
                pending[k] = dist.gather(src, gathered[k], dst=0, async_op=True)
        return st

    def drain():
        for k in range(2 * F):
            if pending[k] is not None:
                with torch.cuda.stream(streams[k // 2]):
                    pending[k].wait()
                pending[k] = None

    # ---- algorithmic bytes of exactly the timed frames (untimed, counting kernels) ----
    alg_bytes = 0
    work = {k: 0 for k in ("branch_visits", "leaf_visits", "aabb_tests", "triangle_tests", "sphere_tests", "hits_accepted",
                           "content_tests")}
    for s in range(args.steps):
        st = ds.render_device(rd_for(args.warmup + s, abi.FLAG_COUNTERS), fb.data_ptr())
        alg_bytes += abi.algorithmic_bytes(st, args.walk)
        for k in work:
            work[k] += getattr(st, k)

    # The frames of the sequence are queued without the host waiting in between (the buffers stay
    # on the device and the stream orders render -> present -> gather); one wait ends the region.
    for h in handles:
        h.set_async(True)
    for s in range(args.warmup):
        step_fn(s, abi.FLAG_TRACE_TIMES)
    drain()
    for h in handles:
        h.wait()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step_fn(args.warmup + s, abi.FLAG_TRACE_TIMES)      # HIP events around the traversal launches only
    queued_s = time.perf_counter() - t0                     # host time to queue the steps
    drain()
    totals = [h.wait() for h in handles]        # statistics summed over each handle's frames
    sync()
    elapsed = time.perf_counter() - t0
    rays = sum(st.rays for st in totals)
    launches = sum(st.traverse_launches for st in totals)
    ms_trav = sum(st.ms_traverse + st.ms_shadow for st in totals)
    for h in handles:
        h.set_async(False)

    # The same frames once more, untimed, with the two chunks one after the other: every traversal
    # launch then has the machine to itself, which is what a per-launch bandwidth should be read on.
    serial = None
    if not (args.serial_chunks and F == 1) and not args.no_serial_leg:
        s_ms, s_launches = 0.0, 0
        for s in range(args.steps):
            st = ds.render_device(rd_for(args.warmup + s, abi.FLAG_TRACE_TIMES | abi.FLAG_SERIAL_CHUNKS), fb.data_ptr())
            s_ms += st.ms_traverse + st.ms_shadow
            s_launches += st.traverse_launches
        if s_ms > 0 and s_launches:
            s_achieved = alg_bytes / (s_ms / 1e3) / 1e9
            serial = {"achieved": round(s_achieved, 2), "frac": round(s_achieved / HBM_PEAK_GBS, 4),
                      "avg_launch_ms": round(s_ms / s_launches, 4), "launches": s_launches}

    tt = torch.tensor([elapsed, float(rays), ms_trav, float(launches), float(alg_bytes)], dtype=torch.float64, device=comm_dev)
    if world > 1:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed = mx[0].item()
        total_rays = tt[1].item()
    else:
        total_rays = float(rays)

    if rank == 0:
        frames = args.steps * world
        value = total_rays / elapsed / 1e6
        avg_launch_s = (ms_trav / 1e3) / max(launches, 1)
        bytes_per_launch = alg_bytes / max(launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        flops = (45 * work["branch_visits"] + 33 * (work["aabb_tests"] + work["content_tests"]) + 17 * work["content_tests"]
                 + 45 * work["triangle_tests"] + 50 * work["hits_accepted"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.walk == "bvh" and (kflags & abi.FLAG_SERIAL_CHUNKS):    # measured for that walk, per whole-frame launch
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (primary+secondary+shadow closest-hit queries)",
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: triangle mesh through KDTree, primary + hard shadow rays, 1080p 1spp",
                "mesh": "seeded bumpy UV sphere, 60024 triangles + 2 floor triangles",
                "kd_tree": kd, "resolution": [W, H], "spp": 1, "light": "DirectionalLight(-1,-1,-1)",
                "integrator": "whitted", "frames_per_step": world, "walk": args.walk,
                "parallelism": (f"frames x{world}, {'RCCL' if backend == 'nccl' else backend} gather of the presented "
                                "RGB8 frames to rank 0" if world > 1 else "single GPU"),
                "step": "render (float framebuffer in HBM) + present (8-bit sRGB, flipped); frames queued asynchronously, one wait at the end",
                "frames_in_flight": F,
            },
            "rays_per_frame": total_rays / frames,
            "ms_per_frame": round(elapsed / frames * 1e3, 4),       # sequence throughput: all ranks' frames
            "host_ms_per_step": round(queued_s / args.steps * 1e3, 4),   # rank 0's host time to queue one step
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": {"bvh": "k_trace_pool<false,3> (primary) and <false,4> (shadow): BVH candidates + KD membership verification",
                           "kd-culled": "k_trace_pool<false,1> (primary) and <false,2> (shadow): culled walk of the reference's KD tree",
                           "reference": "k_trace_pool<false,0>: the reference's KD walk, node for node"}[args.walk],
                "algorithmic_bytes_per_ray": round(alg_bytes / max(rays, 1), 1),
                "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": launches,
                # two half-frame chunks run on two streams: a launch shares the machine with the other
                # chunk's kernels, so its duration (and `achieved`) is that of a kernel on part of the GPU
                "launches_overlap": F > 1 or not args.serial_chunks,
                "serial": serial,
                "fp64_valu_tflops": round(flops / (ms_trav / 1e3) / 1e12, 3) if ms_trav > 0 else None,
                "fp64_valu_frac": round(flops / (ms_trav / 1e3) / 1e12 / FP64_PEAK_TFLOPS, 4) if ms_trav > 0 else None,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(desc, lambda f: rd_for(f, 0), args.warmup)
    for h in handles[1:]:
        h.close()
    ds.close()
    if not args.no_frame16:
        # second half of the headline metric: ms per 1080p 16-spp frame, one frame split into
        # interleaved row bands over all ranks (strong scaling), bands gathered on rank 0
        f16 = frame_16spp(world, rank, dev, comm_dev, stream, W, H)
        if rank == 0:
            out["frame_16spp"] = f16
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def frame_16spp(world, rank, dev, comm_dev, stream, W, H, frames=3):
    """BASELINE.json configs[3]: the mesh with GGX material parameters, albedo texture, checker
    floor and a 1x1 SquareLight (2x2 light samples per hit), Whitted integrator, 4x4 sub-samples."""
    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, device, scenes
    from raytracer_amd import dist as rdist
    scene, _ = scenes.mesh_scene(ggx=True, square_light=True)
    desc = scene.frame(0.0)
    ds = device.DeviceScene(desc, stream=stream)
    # Every rank renders its share of the frame -- the groups of 8 rows g with g % world == rank,
    # one tile of interleaved rows (tile_row_group / tile_row_stride) -- in ONE call, whose two
    # chunks run on the library's two streams.  The mesh fills the middle rows and the slowest
    # rank's share sets the frame time: groups of 8 rows balance the ranks to within a row group
    # without cutting the work into many short launches.  (One rank's share of an 8-rank frame,
    # timed on one GPU: 3.35 ms as 4 bands of rows, 2.78 ms as 12 bands, 2.62 ms this way;
    # 19.1 / 8 = 2.39 ms.)
    rays = [0]

    def one_frame(f):
        cam = scenes.orbit_camera(f / 30.0)

        def render_rows(y0, rows, group, stride, out):
            rd = scenes.render_desc(W, H, integrator=abi.INTEGRATOR_WHITTED, camera=cam, n=4, light_n=2, max_depth=3,
                                    bias=1e-2, frame=f, tile=(0, y0, W, rows), row_group=group, row_stride=stride)
            rays[0] += ds.render_device(rd, out.data_ptr()).rays
        return rdist.render_frame_rows(render_rows, W, H, dev, group=8,
                                       comm_device=None if comm_dev == dev else comm_dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    one_frame(0)
    rays[0] = 0
    sync()
    t0 = time.perf_counter()
    for f in range(1, frames + 1):
        one_frame(f)
    sync()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(rays[0])], dtype=torch.float64, device=comm_dev)
    if world > 1:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt = mx[0].item()
    ds.close()
    return {"workload": "configs[3]: GGX-parameter mesh, texture + checker floor, SquareLight 2x2 samples, 1080p 16 spp, "
                        f"whitted; every rank renders its interleaved groups of 8 rows (group g to rank g % {world}) as one "
                        "tile in one call, float rows gathered on rank 0",
            "ms_per_frame": round(dt / frames * 1e3, 3), "Mrays_per_s": round(tt[1].item() / dt / 1e6, 1),
            "frames": frames, "n_gpus": world, "scaling": "strong"}


def cpu_baseline(desc, rd_of_frame, first_frame, budget_s=12.0, max_frames=40):
    """The CPU oracle (a port of the reference's loop, OpenMP over pixels in place of
    concurrency::parallel_for) on a bounded sample of the same workload: whole frames of the
    timed sequence, starting with the first timed one, until about `budget_s` seconds of CPU
    work have been spent."""
    from oracle import binding as orc
    rays = 0
    frames = 0
    threads = 1
    t0 = time.perf_counter()
    while frames < max_frames:
        _, _, st, threads = orc.render(desc, rd_of_frame(first_frame + frames))
        rays += st.rays
        frames += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{frames} whole 1080p frames of the timed sequence from frame {first_frame} on "
                      f"({rays} rays, {dt:.1f} s; OpenMP schedule(dynamic,64), gcc -O2 -ffp-contract=off)"}


if __name__ == "__main__":
    main()
