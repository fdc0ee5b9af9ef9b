"""The N>1 partition/gather path on CPU: world_size 2 over gloo, with the CPU oracle
standing in for the device renderer.  The assembled image must equal the single-process
image bit for bit (random streams are keyed by the global pixel index)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


W, H = 96, 70


def _scene_and_desc():
    from raytracer_amd import scenes
    scene, cam = scenes.lens_scene()
    return scene, scene.frame(0.0), cam


def _rd(cam, tile=None, frame_t=None):
    from raytracer_amd import abi, scenes
    if frame_t is not None:
        cam = scenes.orbit_camera(frame_t, radius=9.0, height=3.0)
    return scenes.render_desc(W, H, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=4, bias=1e-4,
                              seed=4, clear=(0.3, 0.3, 0.3), tile=tile)


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from raytracer_amd import dist as rdist
    from oracle import binding as orc
    scene, desc, cam = _scene_and_desc()
    if mode in ("bands", "even-bands"):
        def render_rect(x0, y0, w, h, out):
            img, _, _, _ = orc.render(desc, _rd(cam, tile=(x0, y0, w, h)), threads=2)
            out += torch.from_numpy(img)
        frame = rdist.render_frame_bands(render_rect, W, H, torch.device("cpu"), band_rows=16,
                                         bands_per_rank=3 if mode == "even-bands" else None)
        if rank == 0:
            np.save(out_path, frame.numpy())
    elif mode == "rows":
        def render_rows(y0, rows, group, stride, out):
            from raytracer_amd import abi, scenes
            rd = _rd(cam, tile=(0, y0, W, rows))
            rd.tile_row_group, rd.tile_row_stride = group, stride
            img, _, _, _ = orc.render(desc, rd, threads=2)
            out += torch.from_numpy(img)
        frame = rdist.render_frame_rows(render_rows, W, H, torch.device("cpu"), group=8)
        if rank == 0:
            np.save(out_path, frame.numpy())
    elif mode == "rows-rgb8":
        # every rank presents its own rows (8-bit sRGB, unflipped) and those are gathered; the
        # render call overwrites, so the packed buffer is not cleared
        def render_rows(y0, rows, group, stride, out):
            rd = _rd(cam, tile=(0, y0, W, rows))
            rd.tile_row_group, rd.tile_row_stride = group, stride
            img, _, _, _ = orc.render(desc, rd, threads=2)
            out.copy_(torch.from_numpy(img))

        def finish(rows):
            return torch.from_numpy(orc.present(rows.numpy(), passes=1, tonemap=1, flip=False))
        for _ in range(2):              # the second frame reuses every buffer
            frame = rdist.render_frame_rows(render_rows, W, H, torch.device("cpu"), group=8, finish=finish, zero=False)
        if rank == 0:
            np.save(out_path, frame.numpy())
    else:
        def render_frame(f, out):
            img, _, _, _ = orc.render(desc, _rd(cam, frame_t=f / 30.0), threads=2)
            out += torch.from_numpy(img)
        frames = rdist.render_sequence_frames(render_frame, 3, W, H, torch.device("cpu"))
        if rank == 0:
            np.save(out_path, np.stack([f.numpy() for f in frames]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["bands", "even-bands", "rows", "rows-rgb8", "frames"])
def test_two_ranks_reproduce_the_single_process_image(tmp_path, mode):
    out_path = str(tmp_path / "out.npy")
    mp.spawn(_worker, args=(2, _free_port(), mode, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    from oracle import binding as orc
    scene, desc, cam = _scene_and_desc()
    if mode in ("bands", "even-bands", "rows"):
        want, _, _, _ = orc.render(desc, _rd(cam))
        assert np.array_equal(got, want)
    elif mode == "rows-rgb8":
        want, _, _, _ = orc.render(desc, _rd(cam))
        assert got.dtype == np.uint8 and np.array_equal(got, orc.present(want, passes=1, tonemap=1, flip=False))
    else:
        for f in range(3):
            want, _, _, _ = orc.render(desc, _rd(cam, frame_t=f / 30.0))
            assert np.array_equal(got[f], want)


def test_band_partition_covers_the_frame_once():
    from raytracer_amd import dist as rdist
    for world in (1, 2, 3, 8):
        seen = np.zeros(1080, int)
        for r in range(world):
            for (x0, y0, w, h) in rdist.band_rects(1920, 1080, world, r):
                assert x0 == 0 and w == 1920
                seen[y0:y0 + h] += 1
        assert (seen == 1).all()
    assert rdist.frames_for_rank(10, 4, 1) == [1, 5, 9]
    # interleaved groups of rows: one strided tile per rank
    for world in (1, 2, 3, 8):
        for height, group in ((1080, 8), (70, 8), (5, 8), (33, 4)):
            seen = np.zeros(height, int)
            counts = []
            for r in range(world):
                y0, rows, stride = rdist.rank_rows(height, world, r, group)
                counts.append(rows)
                for k in range(rows):
                    seen[y0 + (k // group) * stride + k % group] += 1
            assert (seen == 1).all() and sum(counts) == height
    # equal-share bands: every rank owns the same number of bands and, within a row, of rows
    for world in (1, 2, 3, 4, 8):
        seen = np.zeros(1080, int)
        rows = []
        for r in range(world):
            rects = rdist.band_rects(1920, 1080, world, r, bands_per_rank=4)
            assert len(rects) == 4
            rows.append(sum(h for (_, _, _, h) in rects))
            for (x0, y0, w, h) in rects:
                seen[y0:y0 + h] += 1
        assert (seen == 1).all() and max(rows) - min(rows) <= 4
    assert len(rdist.band_rects(8, 3, 2, 0, bands_per_rank=4)) + len(rdist.band_rects(8, 3, 2, 1, bands_per_rank=4)) == 3


def test_render_frame_rows_returns_a_view_that_the_next_call_overwrites():
    """The frame render_frame_rows returns lives in a buffer kept between frames: a clone of the
    first frame keeps its contents, the tensor returned first shows the second frame afterwards."""
    from raytracer_amd import dist as rdist
    value = [1.0]

    def render_rows(y0, rows, group, stride, out):
        out.fill_(value[0])
    for height in (16, 21):                 # rows in frame order / rows placed through the gather buffers
        value[0] = 1.0
        first = rdist.render_frame_rows(render_rows, 4, height, torch.device("cpu"), group=8)
        kept = first.clone()
        value[0] = 2.0
        second = rdist.render_frame_rows(render_rows, 4, height, torch.device("cpu"), group=8)
        assert first.shape == (height, 4, 3) and torch.all(kept == 1.0) and torch.all(second == 2.0)
        assert first.data_ptr() == second.data_ptr() and torch.all(first == 2.0)
