#!/usr/bin/env python3
"""Single-GPU timings of the five benchmark configurations (bench.py times configs[1] only).

    python tools/bench_configs.py [c1 c2 c3 c4w c4p c5 ...] [--frames K]

Prints one JSON line per configuration: rays by kind, Mrays/s, ms/frame, per-stage GPU times,
host scene-build times.  Framebuffers stay in HBM; scene build / flatten / upload are timed
separately and are not part of ms_per_frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def configs():
    from raytracer_amd import abi, scenes

    def c1():
        s, cam = scenes.whitted_scene()
        return s, lambda f: scenes.render_desc(640, 480, integrator=abi.INTEGRATOR_WHITTED, camera=cam, n=1, max_depth=1,
                                               bias=1e-3, flags=abi.FLAG_WHITTED_REFLECT), 0.0

    def c2():
        s, _ = scenes.mesh_scene()
        return s, lambda f: scenes.render_desc(1920, 1080, integrator=abi.INTEGRATOR_WHITTED, camera=scenes.orbit_camera(f / 30.0),
                                               n=1, max_depth=3, bias=1e-2, clear=(0.21, 0.78, 0.95)), 0.0

    def c3():
        s, cam = scenes.lens_scene()
        return s, lambda f: scenes.render_desc(1920, 1080, integrator=abi.INTEGRATOR_PATH, camera=cam, n=1, max_depth=8,
                                               bias=1e-4, frame=f), 0.0

    def c4w():
        s, _ = scenes.mesh_scene(ggx=True, square_light=True)
        return s, lambda f: scenes.render_desc(1920, 1080, integrator=abi.INTEGRATOR_WHITTED, camera=scenes.orbit_camera(f / 30.0),
                                               n=4, light_n=2, max_depth=3, bias=1e-2, frame=f), 0.0

    def c4p():
        s, _ = scenes.mesh_scene(ggx=True, square_light=True)
        # path-tracing semantics need an emitter: the square light's place is taken by an emissive sphere
        s.sphere((0.0, 10.0, 0.0), 2.0, s.material("PlainWhiteTees", emission=(8.0, 8.0, 8.0)))
        return s, lambda f: scenes.render_desc(1920, 1080, integrator=abi.INTEGRATOR_PATH, camera=scenes.orbit_camera(f / 30.0),
                                               n=4, max_depth=8, bias=1e-3, frame=f), 0.0

    def c5():
        s = scenes.animated_scene(copies=16)
        return s, lambda f: scenes.render_desc(3840, 2160, integrator=abi.INTEGRATOR_PATH,
                                               camera=scenes.orbit_camera(f / 30.0, radius=30.0, height=6.0),
                                               n=8, max_depth=8, bias=1e-3, frame=f), None

    return {"c1": c1, "c2": c2, "c3": c3, "c4w": c4w, "c4p": c4p, "c5": c5}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["c1", "c2", "c3", "c4w", "c4p"])
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--c5-frames", type=int, default=2, help="frames of c5 (each about a second of GPU time and of host preparation)")
    ap.add_argument("--sort-material", action="store_true", help="shade pathTrace levels in material order")
    ap.add_argument("--refill-lanes", action="store_true", help="BVH walk with finished lanes refilled from the queue")
    ap.add_argument("--store-camera-rays", action="store_true", help="camera rays written by a generate kernel and read back")
    ap.add_argument("--accumulate-samples", action="store_true", help="sample slots cleared, added to and resolved")
    ap.add_argument("--serial", action="store_true", help="one chunk at a time on the caller's stream (no two-lane overlap)")
    ap.add_argument("--no-stage-times", action="store_true", help="no per-stage events (their gaps cost time on short frames)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="also time the frames queued asynchronously on this many streams (handles cloned from the scene)")
    ap.add_argument("--brief", action="store_true", help="print only config, rays/s, ms/frame and stage times")
    args = ap.parse_args()
    import torch
    from raytracer_amd import abi, device
    device.init(0)
    table = configs()
    stream = torch.cuda.current_stream().cuda_stream
    for name in args.names:
        scene, rd_of, fixed_t = table[name]()
        animated = fixed_t is None
        frames = min(args.frames, args.c5_frames) if name == "c5" else args.frames
        t0 = time.perf_counter()
        desc = scene.frame(0.0)
        t_host = time.perf_counter() - t0
        t0 = time.perf_counter()
        ds = device.DeviceScene(desc, stream=stream)
        t_up = time.perf_counter() - t0
        rd0 = rd_of(0)
        fb = torch.zeros((rd0.height, rd0.width, 3), dtype=torch.float32, device="cuda")
        ds.render_device(rd_of(0), fb.data_ptr())           # warm-up
        torch.cuda.synchronize()
        acc = {}
        rays = {"rays_primary": 0, "rays_secondary": 0, "rays_shadow": 0}
        host_s = 0.0
        t0 = time.perf_counter()
        for f in range(frames):
            if animated:
                th = time.perf_counter()
                ds.close()
                desc = scene.frame((1 + f) / 30.0)
                ds = device.DeviceScene(desc, stream=stream)
                host_s += time.perf_counter() - th
            fb.zero_()
            rd = rd_of(1 + f)
            rd.flags |= (0 if args.no_stage_times else abi.FLAG_STAGE_TIMES) | (abi.FLAG_SORT_MATERIAL if args.sort_material else 0)
            rd.flags |= abi.FLAG_SERIAL_CHUNKS if args.serial else 0
            rd.flags |= abi.FLAG_REFILL_LANES if args.refill_lanes else 0
            rd.flags |= abi.FLAG_STORE_CAMERA_RAYS if args.store_camera_rays else 0
            rd.flags |= abi.FLAG_ACCUMULATE_SAMPLES if args.accumulate_samples else 0
            st = ds.render_device(rd, fb.data_ptr())
            for k in rays:
                rays[k] += getattr(st, k)
            for k in ("ms_generate", "ms_traverse", "ms_shade", "ms_shadow", "ms_resolve"):
                acc[k] = acc.get(k, 0.0) + getattr(st, k) / frames
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0 - host_s
        total = sum(rays.values())
        out = {"config": name, "sort_material": args.sort_material, "resolution": [rd0.width, rd0.height], "spp": rd0.pixel_multisample_n ** 2,
               "shapes": desc.n_shapes, "kd": scene.kd_stats(), "frames": frames,
               **{k: v // frames for k, v in rays.items()},
               "Mrays_per_s": round(total / dt / 1e6, 1), "ms_per_frame": round(dt / frames * 1e3, 3),
               "stage_ms": {k: round(v, 3) for k, v in acc.items()},
               "host_first_frame_s": round(t_host, 3), "upload_s": round(t_up, 3),
               "host_rebuild_s_per_frame": round(host_s / frames, 3) if animated else None,
               "finite": bool(torch.isfinite(fb).all().item())}
        out["refill_lanes"] = args.refill_lanes
        if args.in_flight > 1 and not animated:
            # the same frames queued without waiting, frame f through handle f mod F on stream f mod F
            F = args.in_flight
            streams = [torch.cuda.Stream() for _ in range(F)]
            clones = [ds.clone(stream=st_.cuda_stream) for st_ in streams]
            fbs = [torch.zeros_like(fb) for _ in range(F)]
            for c in clones:
                c.set_async(True)

            def queue(f):
                k = f % F
                with torch.cuda.stream(streams[k]):
                    fbs[k].zero_()
                    rd = rd_of(1 + f)
                    rd.flags |= abi.FLAG_SERIAL_CHUNKS
                    clones[k].render_device(rd, fbs[k].data_ptr())
            for f in range(F):
                queue(f)
            for c in clones:
                c.wait()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(frames):
                queue(f)
            q_rays = sum(c.wait().rays for c in clones)
            torch.cuda.synchronize()
            dtq = time.perf_counter() - t0
            out["in_flight"] = {"frames_in_flight": F, "ms_per_frame": round(dtq / frames * 1e3, 3),
                                "Mrays_per_s": round(q_rays / dtq / 1e6, 1)}
            for c in clones:
                c.close()
        if args.brief:
            out = {k: out[k] for k in ("config", "refill_lanes", "Mrays_per_s", "ms_per_frame", "stage_ms", "in_flight") if k in out}
        print(json.dumps(out), flush=True)
        if animated:
            # the same frames through the sequence driver: the host work of frame f+1 (redoAABBs,
            # KD build, flatten, BVH build, upload) runs on a worker thread while frame f renders
            from raytracer_amd import sequence
            ds.close()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k, t, dsk, _ in sequence.device_scenes(scene, [(1 + f) / 30.0 for f in range(frames)], stream=stream):
                fb.zero_()
                dsk.render_device(rd_of(1 + k), fb.data_ptr())
            torch.cuda.synchronize()
            print(json.dumps({"config": name, "sequence_prefetch_s_per_frame": round((time.perf_counter() - t0) / frames, 3),
                              "sequential_s_per_frame": round(dt / frames + host_s / frames, 3)}), flush=True)
        ds.close()
        scene.close()


if __name__ == "__main__":
    main()
