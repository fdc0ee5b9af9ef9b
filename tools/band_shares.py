# one rank's share of the 16-spp frame in an N-rank world, on one GPU: how far from 1/N of the frame time?
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, device, scenes, dist as rdist
device.init(0)
W, H = 1920, 1080
scene, _ = scenes.mesh_scene(ggx=True, square_light=True)
desc = scene.frame(0.0)
ds = device.DeviceScene(desc, stream=torch.cuda.current_stream().cuda_stream)
for world in (1, 2, 4, 8):
    for bpr, B in ((4, 4), (max(4, 96 // world), 4)):
        streams = [torch.cuda.Stream() for _ in range(B)]
        handles = [ds.clone(stream=st.cuda_stream) for st in streams]
        for h in handles:
            h.set_async(True)
        worst = 0.0
        for rank in range(world):
            rects = rdist.band_rects(W, H, world, rank, bands_per_rank=bpr)
            bufs = [torch.zeros((r[3], W, 3), dtype=torch.float32, device="cuda") for r in rects]
            def frame(f):
                cam = scenes.orbit_camera(f / 30.0)
                for k, (x0, y0, w, h) in enumerate(rects):
                    rd = scenes.render_desc(W, H, integrator=abi.INTEGRATOR_WHITTED, camera=cam, n=4, light_n=2, max_depth=3,
                                            bias=1e-2, frame=f, tile=(x0, y0, w, h), flags=abi.FLAG_SERIAL_CHUNKS)
                    handles[k % B].render_device(rd, bufs[k].data_ptr())
                for h in handles:
                    h.wait()
            frame(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(1, 6):
                frame(f)
            torch.cuda.synchronize()
            worst = max(worst, (time.perf_counter() - t0) / 5 * 1e3)
        for h in handles:
            h.close()
        print(f"world {world} bands/rank {bpr} streams {B}: slowest rank {worst:.3f} ms per frame", flush=True)
    # the same share as ONE tile of interleaved groups of 8 rows, rendered by one call (two chunks on two streams)
    for flags, tag in ((0, "two lanes"), (abi.FLAG_SERIAL_CHUNKS, "one stream")):
        worst = 0.0
        for rank in range(world):
            y0, rows, stride = rdist.rank_rows(H, world, rank, 8)
            buf = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
            def frame(f):
                rd = scenes.render_desc(W, H, integrator=abi.INTEGRATOR_WHITTED, camera=scenes.orbit_camera(f / 30.0), n=4, light_n=2,
                                        max_depth=3, bias=1e-2, frame=f, tile=(0, y0, W, rows), row_group=8, row_stride=stride, flags=flags)
                ds.render_device(rd, buf.data_ptr())
            frame(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(1, 6):
                frame(f)
            torch.cuda.synchronize()
            worst = max(worst, (time.perf_counter() - t0) / 5 * 1e3)
        print(f"world {world} one tile of interleaved 8-row groups, {tag}: slowest rank {worst:.3f} ms per frame", flush=True)
