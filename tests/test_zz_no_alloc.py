"""rtx_render_device_noalloc against rtx_render_device, through the C ABI.

The allocation-free entry point keeps its ray queues, light-ray queues, sample slots and
counters in a static device array that keeps the previous call's contents.  Its images must be
those of rtx_render_device bit for bit (integer-exact float bits), call after call on one GPU,
including a smaller call that follows a larger one and finds the larger one's data in the array."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


def _cases():
    # (name, scene factory, render-desc keywords, sizes of two calls in a row: the second is smaller)
    yield ("path_lens", scenes.lens_scene,
           dict(integrator=abi.INTEGRATOR_PATH, n=2, max_depth=6, bias=1e-4, passes=2, seed=4, clear=(0.2, 0.7, 0.9)),
           [(160, 120), (48, 30)])
    yield ("whitted_soft", lambda: scenes.mesh_scene(slices=60, stacks=30, ggx=True, square_light=True),
           dict(integrator=abi.INTEGRATOR_WHITTED, n=2, light_n=2, max_depth=2, bias=1e-2, seed=7),
           [(150, 90), (37, 21)])
    yield ("whitted_refract_point", scenes.whitted_scene,
           dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=4, bias=1e-3),
           [(200, 150), (3, 2)])
    yield ("gather", scenes.whitted_scene,
           dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=2, flags=abi.FLAG_GLOBAL_LIGHTING, global_samples=5),
           [(96, 64), (40, 24)])


@pytest.mark.parametrize("name,make,kw,sizes", list(_cases()), ids=[c[0] for c in _cases()])
def test_noalloc_render_equals_render_device(gpu, name, make, kw, sizes):
    import torch
    scene, cam = make()
    desc = scene.frame(0.2)
    kw = dict(kw)
    flags = kw.pop("flags", 0)
    with gpu.DeviceScene(desc) as ds:
        for (W, H) in sizes:
            for extra in (0, abi.FLAG_OVERWRITE):
                rd = scenes.render_desc(W, H, camera=cam, flags=flags | extra, **kw)
                start = torch.full((H, W, 3), 0.125, dtype=torch.float32, device="cuda")
                want = start.clone()
                ds.render_device(rd, want.data_ptr())
                got = start.clone()
                ds.render_device_noalloc(rd, got.data_ptr())
                torch.cuda.synchronize()
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (name, W, H, extra)
                assert torch.isfinite(got).all() and float(got.max()) > 0.0


def test_noalloc_tile_of_a_frame_and_interleaved_rows(gpu):
    """A tile inside a larger frame and a tile of interleaved row groups (one rank's share)."""
    import torch
    scene, cam = scenes.mesh_scene(slices=60, stacks=30)
    desc = scene.frame(0.0)
    with gpu.DeviceScene(desc) as ds:
        for tile, group, stride in (((40, 24, 70, 50), 0, 0), ((0, 8, 150, 40), 8, 16)):
            rd = scenes.render_desc(150, 90, integrator=abi.INTEGRATOR_WHITTED, camera=cam, max_depth=2, bias=1e-2,
                                    tile=tile, row_group=group, row_stride=stride, clear=(0.21, 0.78, 0.95))
            want = torch.zeros((tile[3], tile[2], 3), dtype=torch.float32, device="cuda")
            got = torch.zeros_like(want)
            ds.render_device(rd, want.data_ptr())
            ds.render_device_noalloc(rd, got.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_noalloc_refuses_what_does_not_fit(gpu):
    import torch
    scene, cam = scenes.whitted_scene()
    desc = scene.frame(0.0)
    with gpu.DeviceScene(desc) as ds:
        fb = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
        rd = scenes.render_desc(1920, 1080, integrator=abi.INTEGRATOR_PATH, camera=cam, max_depth=2)
        with pytest.raises(gpu.RtxError) as e:
            ds.render_device_noalloc(rd, fb.data_ptr())
        assert e.value.code == abi.RTX_ERR_LIMIT and "scratch" in str(e.value)
        rd = scenes.render_desc(64, 64, integrator=abi.INTEGRATOR_PATH, camera=cam, max_depth=2, flags=abi.FLAG_COUNTERS)
        with pytest.raises(gpu.RtxError) as e:
            ds.render_device_noalloc(rd, fb.data_ptr())
        assert e.value.code == abi.RTX_ERR_INVALID
        # the largest tile the header promises: 256 x 256 at 1 spp with four light samples per hit
        scene2, cam2 = scenes.mesh_scene(slices=40, stacks=20, ggx=True, square_light=True)
        with gpu.DeviceScene(scene2.frame(0.0)) as ds2:
            rd = scenes.render_desc(256, 256, integrator=abi.INTEGRATOR_WHITTED, camera=cam2, n=1, light_n=2, max_depth=2,
                                    bias=1e-2)
            want = torch.zeros((256, 256, 3), dtype=torch.float32, device="cuda")
            got = torch.zeros_like(want)
            ds2.render_device(rd, want.data_ptr())
            ds2.render_device_noalloc(rd, got.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_noalloc_first_call_on_a_fresh_scene_allocates_nothing(gpu):
    """What the header promises: the entry point is usable as the FIRST render call on a handle (no
    workspace, stream or event of an earlier rtx_render_device call to lean on) and neither it nor
    the kernels it queues take device memory.  Free device memory is read before and after two
    calls (hipMemGetInfo through torch); the image equals rtx_render_device's, rendered afterwards."""
    import torch
    scene, cam = scenes.lens_scene()
    desc = scene.frame(0.1)
    rd = scenes.render_desc(120, 80, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=5, bias=1e-4, seed=9,
                            clear=(0.2, 0.7, 0.9))
    got = torch.zeros((80, 120, 3), dtype=torch.float32, device="cuda")
    again = torch.zeros_like(got)
    want = torch.zeros_like(got)
    torch.cuda.synchronize()
    with gpu.DeviceScene(desc) as ds:
        free_before, _ = torch.cuda.mem_get_info()
        ds.render_device_noalloc(rd, got.data_ptr())
        ds.render_device_noalloc(rd, again.data_ptr())
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before, f"{free_before - free_after} bytes of device memory taken"
        ds.render_device(rd, want.data_ptr())
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        assert torch.equal(again.view(torch.int32), want.view(torch.int32))
        assert float(got.max()) > 0.0
