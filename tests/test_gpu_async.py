"""Asynchronous scene mode (rtx_scene_set_async / rtx_scene_wait): frames queued without the host
waiting give the images and the summed statistics of the same frames rendered one call at a time."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["whitted", "path"])
def test_queued_frames_equal_waited_frames(gpu, kind):
    import torch
    if kind == "whitted":
        scene, cam = scenes.mesh_scene(slices=60, stacks=30)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=3, bias=1e-2, clear=(0.2, 0.7, 0.9))
    else:
        scene, cam = scenes.lens_scene()
        kw = dict(integrator=abi.INTEGRATOR_PATH, n=2, max_depth=8, bias=1e-4)
    desc = scene.frame(0.0)
    W, H, frames = 1280, 900, 4             # large enough for two lanes
    stream = torch.cuda.current_stream().cuda_stream
    with gpu.DeviceScene(desc, stream=stream) as ds:
        rds = [scenes.render_desc(W, H, camera=cam, seed=7, frame=f, flags=abi.FLAG_TRACE_TIMES, **kw) for f in range(frames)]
        waited, shown_w = [], []
        totals = dict(rays_primary=0, rays_secondary=0, rays_shadow=0, traverse_launches=0)
        for rd in rds:
            fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
            out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
            st = ds.render_device(rd, fb.data_ptr())
            ds.present_device(fb.data_ptr(), W, H, out.data_ptr())
            for k in totals:
                totals[k] += getattr(st, k)
            waited.append(fb.cpu().numpy())
            shown_w.append(out.cpu().numpy())
        ds.set_async(True)
        fbs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in rds]
        outs = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in rds]
        for rd, fb, out in zip(rds, fbs, outs):
            assert ds.render_device(rd, fb.data_ptr()) is None
            ds.present_device(fb.data_ptr(), W, H, out.data_ptr())
        st = ds.wait()
        for k, v in totals.items():
            assert getattr(st, k) == v, k
        assert st.ms_traverse > 0
        for a, fb, b, out in zip(waited, fbs, shown_w, outs):
            assert np.array_equal(a, fb.cpu().numpy())
            assert np.array_equal(b, out.cpu().numpy())
        # nothing left over for the next hand-out
        again = ds.wait()
        assert again.rays == 0 and again.traverse_launches == 0
        # host-buffer calls are refused while asynchronous; back to waiting calls
        with pytest.raises(gpu.RtxError):
            ds.render(scenes.render_desc(64, 48, camera=cam, **kw))
        ds.set_async(False)
        img, st1 = ds.render(scenes.render_desc(64, 48, camera=cam, **kw))
        assert st1.rays_primary == 64 * 48 * kw["n"] ** 2 and np.isfinite(img).all()


def test_frames_in_flight_on_cloned_handles(gpu):
    """rtx_scene_clone: handles on one device copy of the scene, each with its own stream, render
    frames concurrently; images equal those of the original handle one call at a time, and the
    clones outlive the handle they were made from."""
    import torch
    scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True, square_light=True)
    desc = scene.frame(0.0)
    W, H, frames = 640, 360, 6
    kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=2, light_n=2, max_depth=3, bias=1e-2)
    rds = [scenes.render_desc(W, H, camera=cam, seed=3, frame=f, **kw) for f in range(frames)]
    ds = gpu.DeviceScene(desc)
    want, rays = [], 0
    for rd in rds:
        img, st = ds.render(rd)
        want.append(img)
        rays += st.rays
    streams = [torch.cuda.Stream() for _ in range(3)]
    clones = [ds.clone(stream=s.cuda_stream) for s in streams]
    ds.close()                                  # the device arrays stay while a clone lives
    for c in clones:
        c.set_async(True)
    fbs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in rds]
    torch.cuda.synchronize()
    for f, rd in enumerate(rds):
        clones[f % 3].render_device(rd, fbs[f].data_ptr())
    got_rays = sum(c.wait().rays for c in clones)
    assert got_rays == rays
    for a, fb in zip(want, fbs):
        assert np.array_equal(a, fb.cpu().numpy())
    for c in clones:
        c.close()
