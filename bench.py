#!/usr/bin/env python3
"""Throughput benchmark of the trace loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1]): the reference's backpack model (67 908 triangles,
Raytracer/Models/backpack/backpack.fbx read by the library's FBX reader and stored as a fixture) plus a
two-triangle floor through the reference's KD tree, Whitted integrator with one DirectionalLight:
primary + hard shadow rays, 1920x1080, 1 sample per pixel (the scene of Raytracer-noah-laptop.cpp:597-611,
orbit camera of :672).  A step is one frame of the orbiting camera (t = frame/30): rtx_render_device
into a float framebuffer resident in HBM, then rtx_present_device (what the reference does between
drawBuffer and the PNG: 8-bit sRGB, flipped).  The frames of the sequence are independent, so a rank
keeps --in-flight of them (default 4) queued on as many streams, each through its own handle on the one
device copy of the scene (rtx_scene_clone).  With N > 1 ranks every rank renders and presents its own
frames and the presented frames are gathered to rank 0 with RCCL inside the timed region; per-GPU work
is fixed, so `value` scales weakly.

The second half of the metric -- ms per 1080p 16-spp frame -- is measured on two frames of
BASELINE.json configs[3], each a first-class object of the line with its own roofline and CPU
baseline: `frame_16spp.whitted_area_light` (GGX-parameter mesh, texture + checker sampling, SquareLight
soft shadows, rayTrace) and `frame_16spp.path_ggx` (the same mesh under pathTrace: CookTorance / GGX Gp
bounces lit by an emissive sphere).  With N > 1 one frame is split over the ranks in interleaved groups of
rows (strong scaling) and the presented rows are gathered on rank 0.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline        all kernels of a frame against the roof they are under: the SIMDs' vector-ALU time
                  (the walk's nodes and shapes come out of L1 / L2 / Infinity Cache; HBM traffic is a few
                  percent of the algorithmic bytes).  Counter-based entries come from profiles/counters.json,
                  which names the rocprofv3 runs they were collected in (same frames as the timed ones
                  when this script is run with the driver's --steps / --warmup)
  cpu_baseline    the CPU oracle timed on a bounded sample of the same workload
  latency         the same frames one at a time (one frame in flight)
  reference_walk  the same frames walked node for node as the reference's traverseKDTree does
                  (KDTree.cpp:158-220), with that walk's own algorithmic bytes per ray
  frame_16spp     the two 1080p 16-spp frames of configs[3]
  configs         configs[0], [2] and one frame of [4], one GPU
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
L2_PEAK_GBS = 34500.0           # aggregate L2 bandwidth of the eight XCDs
FP64_PEAK_TFLOPS = 78.6
CLOCK_GHZ = 2.4
SIMDS = 256 * 4
# The roof of these kernels: every SIMD's vector ALU busy every cycle.  SQ_ACTIVE_INST_VALU counts, per
# wave, the quad-cycles its vector instructions occupy the SIMD's ALU (measured here: 4.1 cycles per
# wave-instruction for this mix of FP64 and FP32 work; a SIMD runs one wave's vector instruction at a
# time), so 4 x the counter summed over a frame's launches is SIMD-cycles of vector-ALU time, to be held
# against SIMDS x clock x frame time.
VALU_PEAK_GCYC = SIMDS * CLOCK_GHZ
# the same roof in wave-instructions per second if every one took four cycles (FP64 and packed ones do,
# plain FP32 ones take two once two waves share a SIMD): a reading aid, not the roof
VALU_GINST_AT_4 = SIMDS * CLOCK_GHZ / 4.0

TRACE_KERNELS = {
    "bvh": "k_trace_pool<false,3> (closest hit) and <false,4> (any hit): BVH candidates + KD membership verification",
    "kd-culled": "k_trace_pool<false,1> (closest hit) and <false,2> (any hit): culled walk of the reference's KD tree",
    "reference": "k_trace_pool<false,0>: the reference's KD walk, node for node",
}


def frame_time(step, rank, world):
    return (step * world + rank) / 30.0


def load_counters():
    """profiles/counters.json: per workload and kernel, the mean counter values per launch and the
    launches per frame, from rocprofv3 --pmc passes of this script (tools/profile_pmc.sh).  None when
    the file is absent."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        return json.load(open(path))
    except Exception:
        return None


def counter_figures(counters, workload, walk, ms_per_frame, dominant):
    """Counter-based entries of a roofline block: vector-ALU time, VALU wave-instructions and HBM / L2
    bytes per frame (all kernels of a frame) against the frame time measured in THIS run, and the HBM
    traffic per launch of the dominant kernel.  The counters were collected with kernels serialised
    by the profiler; the instruction and byte counts of a frame do not depend on that, the time does
    not come from there."""
    if not counters or walk != "bvh":
        return None
    w = counters.get("workloads", {}).get(workload)
    if not w:
        return None
    valu = busy = hbm = l2 = lanes = l2_hit = l2_all = tcp = tcp_miss = 0.0
    dom_traffic, dom_launches = 0.0, 0.0
    per_kernel = {}
    for name, k in w["kernels"].items():
        per_frame = k["launches_per_frame"]
        valu += per_frame * k.get("SQ_INSTS_VALU", 0.0)
        busy += per_frame * k.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0
        lanes += per_frame * k.get("SQ_THREAD_CYCLES_VALU", 0.0)
        bytes_ = (2.0 * k.get("FETCH_SIZE", 0.0) + k.get("WRITE_SIZE", 0.0)) * 1024.0
        hbm += per_frame * bytes_
        l2 += per_frame * k.get("TCC_REQ_sum", 0.0) * 128.0
        l2_hit += per_frame * k.get("TCC_HIT_sum", 0.0)
        l2_all += per_frame * (k.get("TCC_HIT_sum", 0.0) + k.get("TCC_MISS_sum", 0.0))
        tcp += per_frame * k.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
        tcp_miss += per_frame * k.get("TCP_TCC_READ_REQ_sum", 0.0)
        if any(d in name for d in dominant):
            dom_traffic += per_frame * bytes_
            dom_launches += per_frame
        if k.get("SQ_INSTS_VALU", 0.0) > 1e5:
            # the kernel's own figures from the profiled (serialised) launches: vector-ALU time over the
            # launch's own cycles (GRBM_GUI_ACTIVE is summed over the eight XCDs), lanes, cache hit rates
            cyc = k.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            hm = k.get("TCC_HIT_sum", 0.0) + k.get("TCC_MISS_sum", 0.0)
            ta = k.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
            per_kernel[name] = {
                "launches_per_frame": round(per_frame, 2),
                "valu_busy_alone": round(k.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (SIMDS * cyc), 3) if cyc else None,
                "active_lane_frac": round(k.get("SQ_THREAD_CYCLES_VALU", 0.0) / (k["SQ_INSTS_VALU"] * 64.0), 3),
                "l2_hit": round(k.get("TCC_HIT_sum", 0.0) / hm, 3) if hm else None,
                "l1_hit": round(1.0 - k.get("TCP_TCC_READ_REQ_sum", 0.0) / ta, 3) if ta else None,
                "hbm_bytes_per_launch": round(bytes_),
            }
    s = ms_per_frame / 1e3
    out = {
        "valu_busy_gcyc_per_s": round(busy / s / 1e9, 1),
        "valu_busy": round(busy / s / 1e9 / VALU_PEAK_GCYC, 4),
        "valu_winst_per_frame": round(valu),
        "valu_ginst_per_s": round(valu / s / 1e9, 2),
        "active_lane_frac": round(lanes / (valu * 64.0), 3) if valu else None,
        "hbm_bytes_per_frame": round(hbm),
        "hbm_gbs": round(hbm / s / 1e9, 1),
        "hbm_frac": round(hbm / s / 1e9 / HBM_PEAK_GBS, 4),
        "l2_request_bytes_per_frame": round(l2),
        "l2_gbs": round(l2 / s / 1e9, 1),
        "l2_frac": round(l2 / s / 1e9 / L2_PEAK_GBS, 4),
        "l2_hit": round(l2_hit / l2_all, 3) if l2_all else None,
        "l1_hit": round(1.0 - tcp_miss / tcp, 3) if tcp else None,
        "traffic_per_launch": round(dom_traffic / dom_launches) if dom_launches else None,
        "kernels": per_kernel,
        "collected_on": w.get("bench_args"),
        "source": counters.get("source"),
    }
    return out


def roofline_block(walk, alg_bytes, rays, ms_trav, launches, ms_per_frame, cf, overlap=False, flops=None):
    """The roofline object of one workload.  `bound` is the vector ALU: the walk's nodes and shapes come
    out of L1 / L2 / Infinity Cache (HBM traffic is a few percent of the algorithmic bytes), and what a
    frame's time buys is SIMD-cycles of vector-ALU work.  `achieved` = 4 x SQ_ACTIVE_INST_VALU of ALL
    kernels of a frame (profiles/counters.json) over the frame time of this run, `peak` = every SIMD
    busy at the nominal clock, so frac <= 1 holds at step level.  `traffic` = HBM bytes per launch of
    the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes.  The algorithmic bytes of the walk
    (abi.algorithmic_bytes) are given per ray and per launch with the launch duration measured live
    with HIP events on the kernel's stream; they are served from cache, so no fraction of the HBM
    peak is made of them (hbm_frac under `counters` is what reaches HBM)."""
    avg_launch_s = (ms_trav / 1e3) / max(launches, 1)
    r = {
        "bound": "valu",
        "achieved": cf["valu_busy_gcyc_per_s"] if cf else None,
        "peak": round(VALU_PEAK_GCYC, 1),
        "unit": "G SIMD-cycles/s of vector-ALU time (1024 SIMDs x 2.4 GHz)",
        "frac": cf["valu_busy"] if cf else None,
        "traffic": cf["traffic_per_launch"] if cf else None,
        "kernel": TRACE_KERNELS[walk],
        "avg_launch_ms": round(avg_launch_s * 1e3, 4),
        "launches": launches,
        "launches_overlap": overlap,
        "algorithmic_bytes_per_ray": round(alg_bytes / max(rays, 1), 1),
        "algorithmic_bytes_per_launch": round(alg_bytes / max(launches, 1)),
        "counters": cf,
    }
    if cf:
        r["valu_winst_frac_if_4_cycles_each"] = round(cf["valu_ginst_per_s"] / VALU_GINST_AT_4, 4)
    if flops is not None and ms_trav > 0:
        r["fp64_valu_tflops"] = round(flops / (ms_trav / 1e3) / 1e12, 3)
        r["fp64_valu_frac"] = round(flops / (ms_trav / 1e3) / 1e12 / FP64_PEAK_TFLOPS, 4)
    return r


def work_flops(st):
    return (45 * st.branch_visits + 33 * (st.aabb_tests + st.content_tests) + 17 * st.content_tests
            + 45 * st.triangle_tests + 50 * st.hits_accepted)


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
    ap.add_argument("--mesh", default="backpack", help="reference model fixture (backpack, teapot) or 'procedural'")
    ap.add_argument("--repeats", type=int, default=5, help="further timed regions of --steps frames for the spread (the "
                                                           "first region is the reported one)")
    ap.add_argument("--frames16", type=int, default=10, help="frames timed per 16-spp workload")
    ap.add_argument("--texture-size", type=int, default=4096,
                    help="side of the mesh's albedo map in the 16-spp workloads (4096: the size of the map the reference binds "
                         "to the backpack model, Models/backpack/1001_albedo.jpg; 256: the small checker of earlier rounds)")
    ap.add_argument("--frames16-in-flight", type=int, default=3,
                    help="frames of the 16-spp sequence kept in flight for the `pipelined` figure (0: leave it out)")
    ap.add_argument("--checksums", action="store_true",
                    help="tests: the byte sum of every presented frame the pipelined legs deliver on rank 0 (slows them down)")
    ap.add_argument("--sequence-leg", action="store_true",
                    help="run the whole-frames partition of the 16-spp sequence on one rank too (with N > 1 it always runs)")
    ap.add_argument("--only", choices=["all", "main", "f16w", "f16p", "c5"], default="all",
                    help="run one workload only (profiling runs)")
    ap.add_argument("--pmc-run", action="store_true",
                    help="profiling run: only the timed frames are rendered, once (no warm-up frames, counting pass, "
                         "one-at-a-time or reference-walk legs, repeats, other configs or CPU baseline), so every launch "
                         "the profiler sees belongs to one of the --steps (or --frames16) frames a plain run times")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame16", action="store_true", help="skip the 16-spp frames")
    ap.add_argument("--serial-chunks", action="store_true",
                    help="render a frame's two half-frame chunks one after the other instead of on two streams: "
                         "launches then run alone, which is how the per-launch figures of the roofline are best read")
    ap.add_argument("--store-camera-rays", action="store_true", help="RTX_FLAG_STORE_CAMERA_RAYS; same image, for comparison")
    ap.add_argument("--accumulate-samples", action="store_true", help="RTX_FLAG_ACCUMULATE_SAMPLES; same image, for comparison")
    ap.add_argument("--store-light-rays", action="store_true", help="RTX_FLAG_STORE_LIGHT_RAYS; same image, for comparison")
    ap.add_argument("--no-serial-leg", action="store_true",
                    help="skip the legs that repeat the timed frames one at a time and with the reference's walk order "
                         "(`latency`, `reference_walk`)")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[0] / [2] / [4] legs")
    ap.add_argument("--c5-frames", type=int, default=1, help="frames of configs[4] (about 0.6 s of GPU time and 0.6 s of host "
                                                             "preparation each, after 15 s to assemble the scene)")
    ap.add_argument("--walk", choices=["bvh", "kd-culled", "reference"], default="bvh",
                    help="bvh: candidates from the library's BVH, verified against the reference's KD walk "
                         "(default render path); kd-culled: the reference's KD tree walked with culling; "
                         "reference: node for node as the reference's traverseKDTree.  All three give "
                         "identical images.")
    args = ap.parse_args()
    if args.pmc_run:
        args.no_cpu_baseline = True
        args.no_serial_leg = True
        args.no_configs = True
        args.repeats = 0

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
    # BENCH_DIST=1 (under torch.distributed.run with one process) takes the N > 1 code path -- process
    # group, gathers, reductions -- with a single rank: RCCL's calls rehearsed on a one-GPU box
    collective = world > 1 or os.environ.get("BENCH_DIST") == "1"
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    device.init(dev_index)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    counters = load_counters()
    ctx = dict(args=args, rank=rank, world=world, dev=dev, comm_dev=comm_dev, backend=backend, counters=counters,
               collective=collective)

    out = None
    if args.only in ("all", "main"):
        out = main_workload(ctx)
    elif rank == 0:
        out = {"metric": "Mrays/s (primary+secondary+shadow closest-hit queries)", "only": args.only}
    if not args.no_frame16 and args.only not in ("main", "c5"):
        f16 = {}
        if args.only in ("all", "f16w"):
            f16["whitted_area_light"] = frame_16spp(ctx, "whitted")
        if args.only in ("all", "f16p"):
            f16["path_ggx"] = frame_16spp(ctx, "path")
        if rank == 0:
            out["frame_16spp"] = f16
            if world > 1:
                # the scaling figure of the 16-spp metric: one frame over all ranks (strong scaling);
                # `value` above is frames of a sequence dealt out to the ranks (weak scaling)
                # one frame split over the ranks by rows (latency of ONE frame), the same with several frames in
                # flight per rank, and the frames of the sequence dealt out to the ranks whole (frame f to
                # rank f % N, Raytracer.cpp:595-752 is a frame loop): all three hold total work fixed
                out["strong_scaling"] = {k: {"ms_per_frame": v["ms_per_frame"], "Mrays_per_s": v["Mrays_per_s"], "n_gpus": world,
                                             "pipelined_ms_per_frame": (v.get("pipelined") or {}).get("ms_per_frame"),
                                             "sequence_ms_per_frame": (v.get("sequence") or {}).get("ms_per_frame")}
                                         for k, v in f16.items()}
    if args.only == "c5" and world == 1:
        out["configs"] = other_configs(ctx, ("c5",))
    elif args.only == "all" and not args.no_configs and world == 1:
        out["configs"] = other_configs(ctx, ("c1", "c3", "c5"))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


def main_workload(ctx):
    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, device, scenes
    args, rank, world, dev, comm_dev, backend = (ctx[k] for k in ("args", "rank", "world", "dev", "comm_dev", "backend"))
    collective = ctx["collective"]
    W, H = args.width, args.height
    mesh = None if args.mesh == "procedural" else args.mesh
    scene, _ = scenes.mesh_scene(mesh=mesh)
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
    ds = device.DeviceScene(desc, stream=stream, static=True)      # one scene, many frames: rtx_scene_create_static
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
                if (collective and rank == 0) else None for _ in range(2 * F)]
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
            if collective:
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
    alg_bytes, flops = 0, 0
    if not args.pmc_run:
        for s in range(args.steps):
            st = ds.render_device(rd_for(args.warmup + s, abi.FLAG_COUNTERS), fb.data_ptr())
            alg_bytes += abi.algorithmic_bytes(st, args.walk)
            flops += work_flops(st)

    # The frames of the sequence are queued without the host waiting in between (the buffers stay
    # on the device and the stream orders render -> present -> gather); one wait ends the region.
    for h in handles:
        h.set_async(True)
    for s in range(0 if args.pmc_run else args.warmup):
        step_fn(s, abi.FLAG_TRACE_TIMES)
    drain()
    for h in handles:
        h.wait()

    def sync():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(first_step):
        sync()
        t0 = time.perf_counter()
        for s in range(args.steps):
            step_fn(first_step + s, abi.FLAG_TRACE_TIMES)      # HIP events around the traversal launches only
        queued = time.perf_counter() - t0                       # host time to queue the steps
        drain()
        totals = [h.wait() for h in handles]                    # statistics summed over each handle's frames
        sync()
        return time.perf_counter() - t0, queued, totals

    elapsed, queued_s, totals = timed_region(args.warmup)
    rays = sum(st.rays for st in totals)
    launches = sum(st.traverse_launches for st in totals)
    ms_trav = sum(st.ms_traverse + st.ms_shadow for st in totals)
    # the same K frames again, several times: the spread of the timed region
    spread = []
    for _ in range(max(0, args.repeats)):
        e, _, _ = timed_region(args.warmup)
        spread.append(e)
    for h in handles:
        h.set_async(False)

    # The same frames once more, one at a time: render + present, the host waits for each (two
    # half-frame chunks on the library's two streams, which is what a single call does by default).
    latency = reference_walk = None
    if not args.no_serial_leg and world == 1:
        def rd_alone(step):
            rd1 = rd_for(step, abi.FLAG_OVERWRITE)
            rd1.flags = (rd1.flags & ~abi.FLAG_SERIAL_CHUNKS) | (abi.FLAG_SERIAL_CHUNKS if args.serial_chunks else 0)
            return rd1
        ds.render_device(rd_alone(args.warmup), fb.data_ptr())          # the library's two lanes, should this be their first use
        torch.cuda.synchronize()
        l_rays = 0
        t0 = time.perf_counter()
        for s in range(args.steps):
            l_rays += ds.render_device(rd_alone(args.warmup + s), fb.data_ptr()).rays
            ds.present_device(fb.data_ptr(), W, H, shown[0].data_ptr(), passes=1, tonemap=abi.TONEMAP_NONE, flip=True)
        torch.cuda.synchronize()
        l_dt = time.perf_counter() - t0
        latency = {"ms_per_frame": round(l_dt / args.steps * 1e3, 4), "Mrays_per_s": round(l_rays / l_dt / 1e6, 1),
                   "frames": args.steps, "frames_in_flight": 1,
                   "step": "the timed frames again, one at a time: rtx_render_device + rtx_present_device, the host waits for each"}
        # The reference's own traversal work (traverseKDTree, KDTree.cpp:158-220: lesser then greater, no
        # early exit, every leaf's shapes box-tested) on a few of the timed frames: what the GPU does
        # with the walk the reference prescribes, with that walk's algorithmic bytes per ray.
        if args.walk == "bvh":
            n_ref = min(3, args.steps)
            ref_flags = abi.FLAG_REFERENCE_ORDER | abi.FLAG_SERIAL_CHUNKS
            r_bytes = r_rays = 0
            for s in range(n_ref):
                rdr = rd_for(args.warmup + s, abi.FLAG_COUNTERS)
                rdr.flags = (rdr.flags & ~kflags) | ref_flags
                st = ds.render_device(rdr, fb.data_ptr())
                r_bytes += abi.algorithmic_bytes(st, "reference")
                r_rays += st.rays
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r_ms = 0.0
            for s in range(n_ref):
                rdr = rd_for(args.warmup + s, abi.FLAG_TRACE_TIMES | abi.FLAG_OVERWRITE)
                rdr.flags = (rdr.flags & ~kflags) | ref_flags
                st = ds.render_device(rdr, fb.data_ptr())
                r_ms += st.ms_traverse + st.ms_shadow
            torch.cuda.synchronize()
            r_dt = time.perf_counter() - t0
            reference_walk = {
                "Mrays_per_s": round(r_rays / r_dt / 1e6, 1), "ms_per_frame": round(r_dt / n_ref * 1e3, 3), "frames": n_ref,
                "kernel": TRACE_KERNELS["reference"],
                "traversal_ms_per_frame": round(r_ms / n_ref, 3),
                "algorithmic_bytes_per_ray": round(r_bytes / max(r_rays, 1), 1),
                "algorithmic_gbs": round(r_bytes / (r_ms / 1e3) / 1e9, 1) if r_ms > 0 else None,
                "note": "algorithmic bytes = 72 B per branch visit + 8 per leaf + 52 per shape-box test + 72 per triangle / 32 per "
                        "sphere test + 124 per accepted hit + 88 per ray, counted by the same kernel's counting build on these "
                        "frames; the 20 MB tree is served from L2 / Infinity Cache, so the rate may exceed the HBM peak and is "
                        "not held against it",
            }

    tt = torch.tensor([elapsed, float(rays)] + spread, dtype=torch.float64, device=comm_dev)
    if collective:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed = mx[0].item()
        spread = [mx[2 + k].item() for k in range(len(spread))]
        total_rays = tt[1].item()
    else:
        total_rays = float(rays)

    out = None
    if rank == 0:
        frames = args.steps * world
        value = total_rays / elapsed / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        regions = sorted([elapsed] + spread)
        cf = counter_figures(ctx["counters"], "main", args.walk, ms_per_step / world, ("k_trace_pool<false",))
        mesh_name = ("reference model 'backpack' (Raytracer/Models/backpack/backpack.fbx through the library's FBX reader; "
                     "fixture tests/golden/mesh_backpack.npz), 67908 triangles + 2 floor triangles") if mesh == "backpack" else \
            (f"reference model '{mesh}' + 2 floor triangles" if mesh else "seeded bumpy UV sphere, 60024 triangles + 2 floor triangles")
        out = {
            "metric": "Mrays/s (primary+secondary+shadow closest-hit queries)",
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: triangle mesh from Raytracer/Models through KDTree, primary + hard shadow rays, 1080p 1spp",
                "mesh": mesh_name,
                "kd_tree": kd, "resolution": [W, H], "spp": 1, "light": "DirectionalLight(-1,-1,-1)",
                "integrator": "whitted", "frames_per_step": world, "walk": args.walk,
                "parallelism": (f"frames x{world}, {'RCCL' if backend == 'nccl' else backend} gather of the presented "
                                "RGB8 frames to rank 0" if world > 1 else "single GPU"),
                "step": "render (float framebuffer in HBM) + present (8-bit sRGB, flipped); frames queued asynchronously, one wait at the end",
                "scene": "created once with rtx_scene_create_static, the camera moves",
                "frames_in_flight": F,
            },
            "rays_per_frame": total_rays / frames,
            "ms_per_frame": round(elapsed / frames * 1e3, 4),       # sequence throughput: all ranks' frames
            "host_ms_per_step": round(queued_s / args.steps * 1e3, 4),   # rank 0's host time to queue one step
            # the timed region repeated on the same frames: `value` is the first region's
            "timed_region_ms": {"reported": round(elapsed * 1e3, 3), "repeats": len(regions),
                                "min": round(regions[0] * 1e3, 3), "median": round(regions[len(regions) // 2] * 1e3, 3),
                                "max": round(regions[-1] * 1e3, 3)},
            "roofline": roofline_block(args.walk, alg_bytes, rays, ms_trav, launches, ms_per_step / world, cf,
                                       overlap=F > 1 or not args.serial_chunks, flops=flops),
        }
        if latency is not None:
            out["latency"] = latency
            out["one_frame_at_a_time_ms"] = latency["ms_per_frame"]
        if reference_walk is not None:
            out["reference_walk"] = reference_walk
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_frames(desc, lambda f: rd_for(f, 0), args.warmup)
    for h in handles[1:]:
        h.close()
    ds.close()
    return out


def frame_16spp(ctx, kind):
    """BASELINE.json configs[3] at 1920x1080, 4x4 sub-samples per pixel.
    kind 'whitted': rayTrace; the mesh with GGX material parameters and an albedo texture, checker
    floor, one 1x1 SquareLight sampled 2x2 per hit (soft shadows).
    kind 'path': pathTrace, depth 8; the same mesh and floor, CookTorance / GGX Gp on every bounce, lit
    by an emissive sphere where the square light was (pathTrace knows no Light objects)."""
    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, device, scenes
    from raytracer_amd import dist as rdist
    args, rank, world, dev, comm_dev = (ctx[k] for k in ("args", "rank", "world", "dev", "comm_dev"))
    collective = ctx["collective"]
    W, H = args.width, args.height
    frames = max(1, args.frames16)
    mesh = None if args.mesh == "procedural" else args.mesh
    scene = scenes.frame16_scene("path" if kind == "path" else "whitted", mesh=mesh, texture_size=args.texture_size)
    desc = scene.frame(0.0)
    stream = torch.cuda.current_stream().cuda_stream
    ds = device.DeviceScene(desc, stream=stream, static=True)      # one scene, many frames: rtx_scene_create_static
    kflags = {"bvh": 0, "kd-culled": abi.FLAG_KD_CULLED, "reference": abi.FLAG_REFERENCE_ORDER}[args.walk]
    if args.serial_chunks:
        kflags |= abi.FLAG_SERIAL_CHUNKS

    def rd_for(f, flags=0, tile=None, group=0, stride=0):
        return scenes.frame16_render_desc(kind, W, H, f, flags=flags | kflags, tile=tile, row_group=group, row_stride=stride)

    # Every rank renders its share of the frame -- the groups of 8 rows g with g % world == rank, one
    # tile of interleaved rows -- in ONE call (overwriting: no fill), presents its rows (8-bit sRGB)
    # and rank 0 gathers those and places them with one strided copy.
    acc = {"rays": 0, "ms_trav": 0.0, "launches": 0}
    shown_rows = {}

    def one_frame(f, flags):
        def render_rows(y0, rows, group, stride, out):
            st = ds.render_device(rd_for(f, flags | abi.FLAG_OVERWRITE, (0, y0, W, rows), group, stride), out.data_ptr())
            acc["rays"] += st.rays
            acc["ms_trav"] += st.ms_traverse + st.ms_shadow
            acc["launches"] += st.traverse_launches

        def finish(rows_f32):
            n = rows_f32.shape[0]
            if n not in shown_rows:
                shown_rows[n] = torch.zeros((n, W, 3), dtype=torch.uint8, device=dev)
            ds.present_device(rows_f32.data_ptr(), W, n, shown_rows[n].data_ptr(), passes=1,
                              tonemap=abi.TONEMAP_CBRT if kind == "path" else abi.TONEMAP_NONE, flip=False)
            return shown_rows[n]
        return rdist.render_frame_rows(render_rows, W, H, dev, group=8, finish=finish, zero=False,
                                       comm_device=None if comm_dev == dev else comm_dev)

    def sync():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    # algorithmic bytes: one counted frame (the first timed one), untimed
    alg_bytes_frame, flops_frame, rays_frame = 0, 0, 0
    if not args.pmc_run:
        y0, rows, stride = rdist.rank_rows(H, world, rank, 8)
        scratch = torch.zeros((max(rows, 1), W, 3), dtype=torch.float32, device=dev)
        st = ds.render_device(rd_for(1, abi.FLAG_COUNTERS, (0, y0, W, rows), 8, stride), scratch.data_ptr())
        alg_bytes_frame, flops_frame, rays_frame = abi.algorithmic_bytes(st, args.walk), work_flops(st), st.rays
        del scratch
    if not args.pmc_run:
        one_frame(0, abi.FLAG_TRACE_TIMES)
    acc.update(rays=0, ms_trav=0.0, launches=0)
    sync()
    per_frame = []
    t0 = time.perf_counter()
    for f in range(1, frames + 1):
        t1 = time.perf_counter()
        one_frame(f, abi.FLAG_TRACE_TIMES)
        per_frame.append(time.perf_counter() - t1)
    sync()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(acc["rays"])], dtype=torch.float64, device=comm_dev)
    if collective:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt = mx[0].item()
    total_rays = tt[1].item()
    pipelined = sequence = None
    if args.frames16_in_flight > 0 and not args.pmc_run:
        pipelined = pipelined_frames(ctx, kind, ds, rd_for, frames, args.frames16_in_flight, "rows")
        if world > 1 or args.sequence_leg:
            sequence = pipelined_frames(ctx, kind, ds, rd_for, frames, args.frames16_in_flight, "frames")
    out = None
    if rank == 0:
        ms_per_frame = dt / frames * 1e3
        per_frame.sort()
        wl = "f16w" if kind == "whitted" else "f16p"
        dominant = ("k_trace_pool<false, 4>",) if kind == "whitted" else ("k_trace_pool<false, 3>",)
        cf = counter_figures(ctx["counters"], wl, args.walk, ms_per_frame, dominant) if world == 1 else None
        # the counted frame stands for the timed ones (the camera moves 1/30 s per frame)
        scale = (acc["rays"] / rays_frame) if rays_frame else 0.0
        out = {
            "workload": ("configs[3] (rayTrace): GGX-parameter mesh, albedo texture + checker floor, SquareLight 1x1 sampled "
                         "2x2 per hit, 1080p 16 spp" if kind == "whitted" else
                         "configs[3] (pathTrace): the same mesh and floor, CookTorance/GGX Gp bounces, depth 8, lit by an "
                         "emissive sphere, 1080p 16 spp"),
            "mesh": args.mesh, "integrator": kind,
            "albedo_texture": f"{args.texture_size} x {args.texture_size} RGB8 ({args.texture_size * args.texture_size * 3 >> 20} MiB), seeded "
                              "procedural, addressed by the model's own uvs, nearest texel (Texture.cpp:39-55); the reference binds a "
                              "4096 x 4096 map to this model (Models/backpack/1001_albedo.jpg, Model.cpp:91-94)",
            "ms_per_frame": round(ms_per_frame, 3),
            "ms_per_frame_min_median": [round(per_frame[0] * 1e3, 3), round(per_frame[len(per_frame) // 2] * 1e3, 3)],
            "Mrays_per_s": round(total_rays / dt / 1e6, 1),
            "rays_per_frame": round(total_rays / frames),
            "frames": frames, "n_gpus": world, "scaling": "strong",
            "step": ("every rank renders its interleaved groups of 8 rows (group g to rank g % N) as one tile in one call, "
                     "presents them (8-bit sRGB) and rank 0 gathers the presented rows" if world > 1 else
                     "render (float framebuffer in HBM) + present (8-bit sRGB); one frame at a time, its two half-frame chunks "
                     "on the library's two streams"),
            "roofline": roofline_block(args.walk, alg_bytes_frame * scale, acc["rays"], acc["ms_trav"], acc["launches"],
                                       ms_per_frame, cf, overlap=not args.serial_chunks, flops=flops_frame * scale),
        }
        if cf:
            out["active_lane_frac"] = cf["active_lane_frac"]
        if pipelined is not None:
            out["pipelined"] = pipelined
        if sequence is not None:
            out["sequence"] = sequence
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_rows(desc, lambda tile: rd_for(1, 0, tile), W, H)
    ds.close()
    return out


def pipelined_frames(ctx, kind, ds, rd_for, frames, in_flight, partition):
    """The same frames as a sequence with `in_flight` of them queued at a time on every rank (frame
    through a handle cloned from the scene, each handle on a stream of its own, its chunks one
    after the other: the other frames in flight fill its tails).  Two partitions over the ranks:
      "rows"    every frame is split: a rank renders its interleaved groups of 8 rows, presents them and
                rank 0 gathers and places the presented rows;
      "frames"  the frames are dealt out whole, frame f to rank f % world (`frames` per rank, so
                frames x world in all); a rank renders and presents its frame and rank 0 gathers the
                presented frames.
    A level of a pathTrace frame lasts at least as long as its longest ray's chain of dependent steps
    however few rays it holds, so a rank's share of ONE frame does not shrink with the number of ranks
    the way its ray count does; whole frames per rank do not have that floor.  Nothing waits on the
    host until the end.  Returns, on rank 0, the time per frame over all ranks (max over ranks) and
    the rate: the figures to compare across N."""
    import torch
    import torch.distributed as dist
    from raytracer_amd import abi
    from raytracer_amd import dist as rdist
    args, rank, world, dev, comm_dev, backend = (ctx[k] for k in ("args", "rank", "world", "dev", "comm_dev", "backend"))
    collective = ctx["collective"]
    W, H = args.width, args.height
    F = max(1, in_flight)
    by_rows = partition == "rows"
    if by_rows:
        y0, rows, stride = rdist.rank_rows(H, world, rank, 8)
        n_groups = (H + 7) // 8
        per_rank = (n_groups + world - 1) // world
        most = per_rank * 8                              # rows of the busiest rank: what every rank sends
    else:
        y0, rows, stride, per_rank, most = 0, H, 0, 0, H
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    handles = [ds.clone(stream=st.cuda_stream) for st in streams]
    for h in handles:
        h.set_async(True)
    packed = [torch.zeros((most, W, 3), dtype=torch.float32, device=dev) for _ in range(F)]
    shown = [torch.zeros((most, W, 3), dtype=torch.uint8, device=dev) for _ in range(F)]
    gathered = [torch.zeros((world, most, W, 3), dtype=torch.uint8, device=comm_dev) if (collective and rank == 0) else None
                for _ in range(F)]
    frame_buf = [torch.zeros((per_rank * world * 8, W, 3), dtype=torch.uint8, device=comm_dev)
                 if (by_rows and collective and rank == 0) else None for _ in range(F)]
    pending = [None] * F
    tonemap = abi.TONEMAP_CBRT if kind == "path" else abi.TONEMAP_NONE

    checksums = {}                                       # frame -> sum of its presented bytes, as rank 0 holds them
    frame_of = [None] * F

    def place(k):
        # frame row (j * world + r) * 8 + i  <-  rank r, packed row j * 8 + i
        if by_rows:
            frame_buf[k].view(per_rank, world, 8, W, 3).copy_(gathered[k].view(world, per_rank, 8, W, 3).transpose(0, 1))
            if args.checksums:
                checksums[frame_of[k]] = int(frame_buf[k][:H].sum(dtype=torch.int64).item())
        elif args.checksums:
            for r in range(world):
                checksums[frame_of[k] * world + r] = int(gathered[k][r].sum(dtype=torch.int64).item())

    def step(j):
        # the j-th step of this rank: frame j split by rows, or this rank's own frame j * world + rank
        k = j % F
        f = j if by_rows else j * world + rank
        with torch.cuda.stream(streams[k]):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
                if rank == 0:
                    place(k)
            if rows:
                tile = (0, y0, W, rows)
                rd = rd_for(f, abi.FLAG_OVERWRITE | abi.FLAG_SERIAL_CHUNKS, tile, 8 if by_rows else 0, stride)
                handles[k].render_device(rd, packed[k].data_ptr())
                handles[k].present_device(packed[k].data_ptr(), W, rows, shown[k].data_ptr(), passes=1, tonemap=tonemap,
                                          flip=not by_rows)
            if collective:
                src = shown[k] if backend == "nccl" else shown[k].cpu()
                frame_of[k] = j
                pending[k] = dist.gather(src, list(gathered[k].unbind(0)) if rank == 0 else None, dst=0, async_op=True)
            elif args.checksums:
                # (one rank: the presented frame is the frame; summed at once, which waits for it)
                checksums[f] = int(shown[k][:rows].sum(dtype=torch.int64).item())

    def drain():
        for k in range(F):
            if pending[k] is not None:
                with torch.cuda.stream(streams[k]):
                    pending[k].wait()
                    if rank == 0:
                        place(k)
                pending[k] = None

    def sync():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for j in range(F):                                   # warm-up: every handle once
        step(j)
    drain()
    for h in handles:
        h.wait()
    sync()
    t0 = time.perf_counter()
    for j in range(F, F + frames):
        step(j)
    drain()
    rays = sum(h.wait().rays for h in handles)
    sync()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(rays)], dtype=torch.float64, device=comm_dev)
    if collective:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt = mx[0].item()
    for h in handles:
        h.close()
    if rank != 0:
        return None
    total_frames = frames if by_rows else frames * world
    return {"partition": partition, **({"checksums": {str(k): v for k, v in sorted(checksums.items())}} if args.checksums else {}), "frames_in_flight": F, "frames": total_frames,
            "ms_per_frame": round(dt / total_frames * 1e3, 3),
            "Mrays_per_s": round(tt[1].item() / dt / 1e6, 1), "n_gpus": world,
            "hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (HIP default)"),
            "step": ("frames of the sequence queued without waiting, one stream per frame in flight; per frame every rank renders "
                     "and presents its interleaved rows" + (" and rank 0 gathers and places them" if world > 1 else "")) if by_rows else
                    ("frames of the sequence dealt out whole, frame f to rank f % N, queued without waiting on one stream per frame "
                     "in flight; a rank renders and presents its frame and rank 0 gathers the presented frames")}


def other_configs(ctx, which):
    """BASELINE.json configs[0], [2] and [4] on one GPU, each a short leg of the default run:
      c1  the classic Whitted scene (glass + mirror sphere, checker floor, one PointLight), 640x480, 1 spp,
          depth 1 -- the configuration the reference's CPU path is quoted on; also timed on the CPU oracle;
      c3  the same plus a Biconvex lens, mirror and copper spheres and an emissive sphere, pathTrace to
          depth 8 with CookTorance shading, 1080p;
      c5  ONE frame of the animated 4K sequence: sixteen 60 k-triangle meshes (960 k triangles), three
          emissive spheres on circle0/1/2, pathTrace depth 8, 3840x2160 at 64 spp, with the host's
          per-frame preparation (redoAABBs + buildKDTree + flatten, rtx_scene_create) timed beside it.
    Each: frames one at a time (render + wait), and c1 / c3 also with four frames in flight."""
    import torch
    from raytracer_amd import abi, device, scenes
    args, dev = ctx["args"], ctx["dev"]
    stream = torch.cuda.current_stream().cuda_stream
    out = {}

    def timed(ds, rd_of, frames, W, H, in_flight=0):
        fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        ds.render_device(rd_of(0, abi.FLAG_OVERWRITE), fb.data_ptr())
        torch.cuda.synchronize()
        rays = 0
        t0 = time.perf_counter()
        for f in range(1, frames + 1):
            rays += ds.render_device(rd_of(f, abi.FLAG_OVERWRITE), fb.data_ptr()).rays
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r = {"ms_per_frame": round(dt / frames * 1e3, 4), "Mrays_per_s": round(rays / dt / 1e6, 1),
             "rays_per_frame": round(rays / frames), "frames": frames, "finite": bool(torch.isfinite(fb).all().item())}
        if in_flight > 1:
            streams = [torch.cuda.Stream(device=dev) for _ in range(in_flight)]
            clones = [ds.clone(stream=st.cuda_stream) for st in streams]
            fbs = [torch.zeros_like(fb) for _ in range(in_flight)]
            for c in clones:
                c.set_async(True)

            def queue(f):
                k = f % in_flight
                with torch.cuda.stream(streams[k]):
                    clones[k].render_device(rd_of(f, abi.FLAG_OVERWRITE | abi.FLAG_SERIAL_CHUNKS), fbs[k].data_ptr())
            for f in range(in_flight):
                queue(f)
            for c in clones:
                c.wait()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(1, 4 * frames + 1):
                queue(f)
            q_rays = sum(c.wait().rays for c in clones)
            torch.cuda.synchronize()
            dtq = time.perf_counter() - t0
            r["in_flight"] = {"frames_in_flight": in_flight, "frames": 4 * frames, "ms_per_frame": round(dtq / (4 * frames) * 1e3, 4),
                              "Mrays_per_s": round(q_rays / dtq / 1e6, 1)}
            for c in clones:
                c.close()
        return r

    if "c1" in which:
        out["c1"] = config1_leg(ctx, timed)
    if "c3" in which:
        out["c3"] = config3_leg(ctx, timed)
    if "c5" in which and args.c5_frames > 0:
        out["c5"] = config5_leg(ctx)
    return out


def config1_leg(ctx, timed):
    import torch
    from raytracer_amd import abi, device, scenes
    args = ctx["args"]
    stream = torch.cuda.current_stream().cuda_stream
    scene, cam = scenes.whitted_scene()
    desc = scene.frame(0.0)

    def rd_c1(f, flags=0):
        return scenes.render_desc(640, 480, integrator=abi.INTEGRATOR_WHITTED, camera=cam, n=1, max_depth=1, bias=1e-3, frame=f,
                                  flags=flags)
    with device.DeviceScene(desc, stream=stream, static=True) as ds:
        c1 = timed(ds, rd_c1, 20, 640, 480, in_flight=4)
    c1["workload"] = ("configs[0]: classic Whitted scene (glass + mirror sphere over a checker floor of two triangles, one PointLight), "
                      "640x480, 1 spp, depth 1; rayTrace as the reference has it (refraction recurses, no mirror branch, "
                      "Raytracer-Noah-Desktop.cpp:452-564)")
    c1["kd_tree"] = scene.kd_stats()
    if not args.no_cpu_baseline:
        from oracle import binding as orc
        t0 = time.perf_counter()
        n, cpu_rays, threads = 0, 0, 1
        while n < 50 and time.perf_counter() - t0 < 3.0:
            _, _, st, threads = orc.render(desc, rd_c1(1 + n))
            cpu_rays += st.rays
            n += 1
        dt = time.perf_counter() - t0
        c1["cpu_baseline"] = {"value": round(cpu_rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                              "ms_per_frame": round(dt / n * 1e3, 2), "sample": f"{n} whole frames ({cpu_rays} rays, {dt:.1f} s)"}
    scene.close()
    return c1


def config3_leg(ctx, timed):
    import torch
    from raytracer_amd import abi, device, scenes
    args = ctx["args"]
    stream = torch.cuda.current_stream().cuda_stream
    scene, cam = scenes.lens_scene()
    desc = scene.frame(0.0)

    def rd_c3(f, flags=0):
        return scenes.render_desc(1920, 1080, integrator=abi.INTEGRATOR_PATH, camera=cam, n=1, max_depth=8, bias=1e-4, frame=f,
                                  flags=flags)
    with device.DeviceScene(desc, stream=stream, static=True) as ds:
        c3 = timed(ds, rd_c3, 20, 1920, 1080, in_flight=4)
    c3["workload"] = ("configs[2]: the Whitted scene plus a Biconvex lens, mirror and copper spheres, lit by an emissive sphere; "
                      "pathTrace to depth 8 with CookTorance shading, 1080p, 1 spp")
    if not args.no_cpu_baseline:
        from oracle import binding as orc
        t0 = time.perf_counter()
        _, _, st, threads = orc.render(desc, rd_c3(1, 0))
        dt = time.perf_counter() - t0
        c3["cpu_baseline"] = {"value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                              "ms_per_frame": round(dt * 1e3, 1), "sample": f"one whole frame ({st.rays} rays, {dt:.1f} s)"}
    scene.close()
    return c3


def config5_leg(ctx):
    """configs[4], one frame (or a few) of the sequence."""
    import torch
    from raytracer_amd import abi, device, scenes
    args, dev = ctx["args"], ctx["dev"]
    stream = torch.cuda.current_stream().cuda_stream
    if True:
        t0 = time.perf_counter()
        scene = scenes.animated_scene(copies=16)
        assemble_s = time.perf_counter() - t0
        W5, H5 = 3840, 2160
        fb = torch.zeros((H5, W5, 3), dtype=torch.float32, device=dev)
        shown = torch.zeros((H5, W5, 3), dtype=torch.uint8, device=dev)
        host_s = create_s = gpu_s = 0.0
        rays = 0
        tris = 0
        # frame 0 warms the host's allocations and the device's workspaces (left out of a profiling run)
        for f in range(1 if args.pmc_run else 0, args.c5_frames + 1):
            t = (1 + f) / 30.0
            t0 = time.perf_counter()
            desc = scene.frame(t)                        # redoAABBs + buildKDTree + flatten (Raytracer.cpp:634-636)
            t1 = time.perf_counter()
            ds = device.DeviceScene(desc, stream=stream)    # rtx_scene_create: derived tables, candidate BVH, upload
            t2 = time.perf_counter()
            rd = scenes.render_desc(W5, H5, integrator=abi.INTEGRATOR_PATH, camera=scenes.orbit_camera(t, radius=30.0, height=6.0),
                                    n=8, max_depth=8, bias=1e-3, frame=1 + f, flags=abi.FLAG_OVERWRITE)
            st = ds.render_device(rd, fb.data_ptr())
            ds.present_device(fb.data_ptr(), W5, H5, shown.data_ptr(), passes=1, tonemap=abi.TONEMAP_CBRT, flip=True)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            tris = desc.n_triangles
            ds.close()
            if f > 0:
                host_s += t1 - t0; create_s += t2 - t1; gpu_s += t3 - t2; rays += st.rays
        n = args.c5_frames
        out = {
            "workload": "configs[4]: one frame of the animated 4K sequence -- 16 seeded 60 k-triangle meshes + floor, three emissive spheres "
                        "on circle0/1/2 (Animatable.cpp:22-35), pathTrace depth 8 with CookTorance, 3840x2160, 64 spp (n = 8), t = frame/30",
            "triangles": tris, "kd_tree": scene.kd_stats(), "frames": n,
            "ms_per_frame": round(gpu_s / n * 1e3, 1), "Mrays_per_s": round(rays / gpu_s / 1e6, 1), "rays_per_frame": round(rays / n),
            "host_prepare_ms_per_frame": {"redoAABBs_buildKDTree_flatten": round(host_s / n * 1e3, 1), "rtx_scene_create": round(create_s / n * 1e3, 1),
                                          "note": "per frame of an animated scene; raytracer_amd/sequence.py runs it on a worker thread while the "
                                                  "previous frame renders"},
            "scene_assembly_s_once": round(assemble_s, 1),
            "finite": bool(torch.isfinite(fb).all().item()),
            "cpu_baseline": None,
            "cpu_baseline_note": "not timed: 530 M camera paths per frame are hours of CPU oracle time",
            # the one configuration whose scene (about 1.2 GB) does not fit the 256 MB Infinity Cache
            "counters": counter_figures(ctx["counters"], "c5", args.walk, gpu_s / n * 1e3, ("k_trace_pool<false, 3>",)),
        }
        scene.close()
    return out


def cpu_baseline_frames(desc, rd_of_frame, first_frame, budget_s=10.0, max_frames=40):
    """The CPU oracle (a port of the reference's loop, OpenMP over pixels in place of
    concurrency::parallel_for) on a bounded sample of the same workload: whole frames of the
    timed sequence, starting with the first timed one, until about `budget_s` seconds of CPU
    work have been spent.  The CPU walks the reference's KD tree in the reference's order (the GPU
    finds the same hits through its own BVH): same rays, same answers, not the same work.
    `reference_costs` repeats a part of the sample with the overheads the reference itself has
    (leaf vectors copied per visit, sphere matrices per test / hit)."""
    from oracle import binding as orc
    rays, frames, threads = 0, 0, 1
    t0 = time.perf_counter()
    while frames < max_frames:
        _, _, st, threads = orc.render(desc, rd_of_frame(first_frame + frames))
        rays += st.rays
        frames += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    orc.set_reference_costs(True)
    try:
        t1 = time.perf_counter()
        rc_rays, rc_frames = 0, 0
        while rc_frames < max(1, frames // 3):
            _, _, st, _ = orc.render(desc, rd_of_frame(first_frame + rc_frames))
            rc_rays += st.rays
            rc_frames += 1
        rc_dt = time.perf_counter() - t1
    finally:
        orc.set_reference_costs(False)
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{frames} whole 1080p frames of the timed sequence from frame {first_frame} on "
                      f"({rays} rays, {dt:.1f} s; OpenMP schedule(dynamic,64), gcc -O2 -ffp-contract=off)",
            "walk": "the reference's KD tree in the reference's visit order (KDTree.cpp:158-220)",
            "reference_costs": {"value": round(rc_rays / rc_dt / 1e6, 4), "unit": "Mrays/s",
                                "sample": f"{rc_frames} of those frames with the leaf vector copied per visit (KDTree.cpp:206) "
                                          f"and the sphere matrix work per test and hit (Sphere.cpp:47-48,74) ({rc_dt:.1f} s)"}}


def cpu_baseline_rows(desc, rd_of_tile, W, H, budget_s=8.0, band=4):
    """The CPU oracle on bands of rows of one 16-spp frame (the first timed one), alternating
    outwards from the middle of the picture (where the mesh is) until about `budget_s` seconds
    have been spent; then one band again with the reference's own overheads."""
    from oracle import binding as orc
    rays, rows, threads = 0, 0, 1
    mid = H // 2
    order = [mid]
    for j in range(1, H // band):
        order += [y for y in (mid + j * band, mid - j * band) if 0 <= y and y + band <= H]
    t0 = time.perf_counter()
    used = []
    for y in order:
        _, _, st, threads = orc.render(desc, rd_of_tile((0, y, W, band)))
        rays += st.rays
        rows += band
        used.append(y)
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    orc.set_reference_costs(True)
    try:
        t1 = time.perf_counter()
        _, _, st, _ = orc.render(desc, rd_of_tile((0, used[0], W, band)))
        rc = st.rays / (time.perf_counter() - t1) / 1e6
    finally:
        orc.set_reference_costs(False)
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "ms_per_frame_extrapolated": round(dt / rows * H * 1e3, 1),
            "sample": f"{rows} of {H} rows of the first timed frame, bands of {band} rows outwards from the middle row "
                      f"({min(used)}..{max(used) + band - 1}; {rays} rays, {dt:.1f} s; OpenMP schedule(dynamic,64), gcc -O2 "
                      "-ffp-contract=off)",
            "walk": "the reference's KD tree in the reference's visit order (KDTree.cpp:158-220)",
            "reference_costs": {"value": round(rc, 4), "unit": "Mrays/s", "sample": f"the middle band of {band} rows again"}}


if __name__ == "__main__":
    main()
