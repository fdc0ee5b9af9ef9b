"""GPU parity of the per-pixel loop (rtx_render) against the CPU oracle.

Tolerances (from the task's correctness statement): per-pixel RGB within 1e-4
relative; for multi-bounce images at least 60 dB PSNR.  Ray counts must agree
exactly.  Differences come only from libm (pow, acos, atan2) and, for the
path integrator, from carrying the weight product forward instead of folding
it on the way back up the recursion."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes
from oracle import binding as orc
from helpers import psnr, rel_err

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4


def _both(gpu, desc, rd, passes_split=None):
    with gpu.DeviceScene(desc) as ds:
        img, stats = ds.render(rd)
    ref, _, ref_stats, _ = orc.render(desc, rd)
    return img, stats, ref, ref_stats


def _same_ray_counts(a, b):
    assert (a.rays_primary, a.rays_secondary, a.rays_shadow) == (b.rays_primary, b.rays_secondary, b.rays_shadow)


def test_whitted_point_light_depth1(gpu):
    """Config 1: glass + mirror sphere, checker floor, one PointLight, 1 spp."""
    scene, cam = scenes.whitted_scene()
    desc = scene.frame(0.0)
    rd = scenes.render_desc(160, 120, integrator=abi.INTEGRATOR_WHITTED, camera=cam, max_depth=1, bias=1e-3,
                            flags=abi.FLAG_COUNTERS | abi.FLAG_REFERENCE_ORDER)
    img, stats, ref, ref_stats = _both(gpu, desc, rd)
    _same_ray_counts(stats, ref_stats)
    assert stats.rays_shadow > 0 and stats.rays_secondary > 0
    for k in ("branch_visits", "leaf_visits", "aabb_tests", "triangle_tests", "sphere_tests", "hits_accepted"):
        assert getattr(stats, k) == getattr(ref_stats, k), k
    assert rel_err(img, ref).max() < REL_TOL
    assert img.max() > 0.05


def test_whitted_reflect_extension(gpu):
    scene, cam = scenes.whitted_scene()
    desc = scene.frame(0.0)
    rd = scenes.render_desc(128, 96, integrator=abi.INTEGRATOR_WHITTED, camera=cam, max_depth=3, bias=1e-3,
                            flags=abi.FLAG_WHITTED_REFLECT, clear=(0.21, 0.78, 0.95))
    img, stats, ref, ref_stats = _both(gpu, desc, rd)
    _same_ray_counts(stats, ref_stats)
    assert psnr(img, ref) >= 60.0
    assert np.quantile(rel_err(img, ref), 0.999) < REL_TOL


def test_mesh_directional_light_hard_shadows(gpu):
    """Config 2 at reduced size: triangle soup through the KD tree, primary + shadow rays."""
    scene, cam = scenes.mesh_scene(slices=100, stacks=50)
    desc = scene.frame(0.0)
    rd = scenes.render_desc(192, 108, integrator=abi.INTEGRATOR_WHITTED, camera=cam, max_depth=3, bias=1e-2,
                            clear=(0.21, 0.78, 0.95), flags=abi.FLAG_COUNTERS | abi.FLAG_REFERENCE_ORDER)
    img, stats, ref, ref_stats = _both(gpu, desc, rd)
    _same_ray_counts(stats, ref_stats)
    assert stats.branch_visits == ref_stats.branch_visits and stats.aabb_tests == ref_stats.aabb_tests
    assert rel_err(img, ref).max() < REL_TOL


def _scene_cases():
    yield "whitted", scenes.whitted_scene, dict(integrator=abi.INTEGRATOR_WHITTED, max_depth=3, bias=1e-3)
    yield "mesh", lambda: scenes.mesh_scene(slices=100, stacks=50), dict(integrator=abi.INTEGRATOR_WHITTED,
                                                                        max_depth=3, bias=1e-2)
    yield "mesh_soft", lambda: scenes.mesh_scene(slices=60, stacks=30, ggx=True, square_light=True), dict(
        integrator=abi.INTEGRATOR_WHITTED, n=2, light_n=2, max_depth=2, bias=1e-2)
    yield "live_path", scenes.live_scene, dict(integrator=abi.INTEGRATOR_PATH, n=2, max_depth=8, bias=1e-4)
    yield "lens_path", scenes.lens_scene, dict(integrator=abi.INTEGRATOR_PATH, n=2, max_depth=8, bias=1e-4,
                                               clear=(0.2, 0.7, 0.9))
    yield "mesh_path", lambda: scenes.mesh_scene(slices=60, stacks=30, ggx=True), dict(
        integrator=abi.INTEGRATOR_PATH, n=2, max_depth=4, bias=1e-4, clear=(2.0, 2.0, 2.0))


@pytest.mark.parametrize("name,make,kw", list(_scene_cases()), ids=[c[0] for c in _scene_cases()])
def test_culled_walk_gives_identical_images(gpu, name, make, kw):
    """The default render path skips KD nodes that cannot change the closest hit and stops
    shadow rays at the first occluder; its images must be bit-identical to the ones produced
    by walking the tree exactly as the reference does, with far fewer box tests."""
    scene, cam = make()
    desc = scene.frame(0.3)
    kw = dict(kw)
    flags = kw.pop("flags", 0)
    fast = scenes.render_desc(144, 96, camera=cam, seed=9, flags=flags | abi.FLAG_COUNTERS, **kw)
    mid = scenes.render_desc(144, 96, camera=cam, seed=9, flags=flags | abi.FLAG_COUNTERS | abi.FLAG_KD_CULLED, **kw)
    slow = scenes.render_desc(144, 96, camera=cam, seed=9,
                              flags=flags | abi.FLAG_COUNTERS | abi.FLAG_REFERENCE_ORDER, **kw)
    with gpu.DeviceScene(desc) as ds:
        a, sa = ds.render(fast)         # BVH candidates + membership verification (default)
        m, sm = ds.render(mid)          # reference KD tree walked with culling
        b, sb = ds.render(slow)         # reference order, nothing skipped
        fast.flags |= abi.FLAG_REFILL_LANES
        c, sc = ds.render(fast)         # BVH walk with finished lanes refilled from the queue
    assert np.array_equal(a, b)
    assert np.array_equal(c, b)
    _same_ray_counts(sc, sb)
    for k in ("content_tests", "triangle_tests", "sphere_tests", "aabb_tests", "hits_accepted", "fallback_rays"):
        assert getattr(sc, k) == getattr(sa, k), k       # the same walk per ray, only scheduled differently
    assert 0 < sa.lane_steps <= sa.wave_steps and 0 < sc.lane_steps <= sc.wave_steps
    assert np.array_equal(m, b)
    _same_ray_counts(sa, sb)
    _same_ray_counts(sm, sb)
    assert sb.content_tests == 0
    if desc.n_branches > 100:
        assert sm.aabb_tests + sm.content_tests + sm.branch_visits < sb.aabb_tests + sb.branch_visits
        assert sa.aabb_tests + sa.content_tests + sa.branch_visits < sb.aabb_tests + sb.branch_visits


def test_square_light_soft_shadows_multisample(gpu):
    """SquareLight sampling drives the counter-based RNG: identical samples on both sides."""
    scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True, square_light=True)
    desc = scene.frame(0.3)
    rd = scenes.render_desc(96, 72, integrator=abi.INTEGRATOR_WHITTED, camera=cam, n=2, light_n=2, max_depth=2,
                            bias=1e-2, seed=7)
    img, stats, ref, ref_stats = _both(gpu, desc, rd)
    _same_ray_counts(stats, ref_stats)
    # every lit hit sends light_n * light_n = 4 shadow rays; nothing is refracted in this scene
    assert stats.rays_secondary == 0 and stats.rays_shadow % 4 == 0 and 0 < stats.rays_shadow <= 4 * stats.rays_primary
    assert rel_err(img, ref).max() < REL_TOL


def test_path_live_scene_depth8(gpu):
    """pathTrace on the reference's own scene: mirror, glass, emissive spheres."""
    scene, cam = scenes.live_scene()
    desc = scene.frame(0.5)
    rd = scenes.render_desc(96, 96, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=8, bias=1e-4,
                            passes=2, seed=0)
    img, stats, ref, ref_stats = _both(gpu, desc, rd)
    _same_ray_counts(stats, ref_stats)
    assert psnr(img, ref) >= 60.0
    assert np.quantile(rel_err(img, ref), 0.999) < REL_TOL


def test_path_lens_scene_config3(gpu):
    """Config 3 at reduced size: Biconvex refraction + mirror + CookTorance, depth 8."""
    scene, cam = scenes.lens_scene()
    desc = scene.frame(0.0)
    rd = scenes.render_desc(128, 96, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=8, bias=1e-4,
                            clear=(0.21, 0.78, 0.95), seed=3)
    img, stats, ref, ref_stats = _both(gpu, desc, rd)
    _same_ray_counts(stats, ref_stats)
    assert psnr(img, ref) >= 60.0
    assert np.quantile(rel_err(img, ref), 0.999) < REL_TOL


def test_path_ggx_textured_mesh(gpu):
    """Config 4 materials under path semantics: texture + checker sampling, GGX geometry term."""
    scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True)
    desc = scene.frame(0.2)
    rd = scenes.render_desc(80, 60, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=4, bias=1e-4,
                            clear=(2.0, 2.0, 2.0), seed=11)
    img, stats, ref, ref_stats = _both(gpu, desc, rd)
    _same_ray_counts(stats, ref_stats)
    assert psnr(img, ref) >= 60.0


def test_basic_flat_colour_mode(gpu):
    scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True)
    desc = scene.frame(0.0)
    rd = scenes.render_desc(96, 64, integrator=abi.INTEGRATOR_PATH, camera=cam, flags=abi.FLAG_BASIC,
                            clear=(0.5, 0.5, 0.5))
    img, stats, ref, ref_stats = _both(gpu, desc, rd)
    assert np.array_equal(img, ref)


def test_tiles_passes_and_chunks_compose(gpu, monkeypatch):
    """A frame rendered as tiles, pass by pass, with tiny chunks equals the frame rendered at once."""
    scene, cam = scenes.whitted_scene()
    desc = scene.frame(0.0)
    W, H = 100, 70
    full = scenes.render_desc(W, H, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=3, passes=3, seed=5,
                              clear=(1.0, 1.0, 1.0))
    with gpu.DeviceScene(desc) as ds:
        whole, _ = ds.render(full)
        monkeypatch.setenv("RTX_MAX_PATHS", "4096")
        out = np.zeros((H, W, 3), np.float32)
        for (x0, y0, w, h) in [(0, 0, 37, 70), (37, 0, 63, 33), (37, 33, 63, 37)]:
            acc = np.zeros((h, w, 3), np.float32)
            for p in range(3):
                rd = scenes.render_desc(W, H, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=3,
                                        passes=1, first_pass=p, seed=5, clear=(1.0, 1.0, 1.0), tile=(x0, y0, w, h))
                acc, _ = ds.render(rd, accum=acc)
            out[y0:y0 + h, x0:x0 + w] = acc
    assert np.array_equal(out, whole)


@pytest.mark.parametrize("kind", ["path", "whitted_soft", "whitted_reflect", "gather"])
def test_two_chunks_at_a_time_equal_one_after_the_other(gpu, monkeypatch, kind):
    """Chunks rendered in pairs on the library's two streams (forced here at a small size, with
    many chunks per frame) give the image and ray counts of the one-stream order, bit for bit."""
    if kind == "path":
        scene, cam = scenes.lens_scene()
        kw = dict(integrator=abi.INTEGRATOR_PATH, n=2, max_depth=8, bias=1e-4, passes=2)
    elif kind == "whitted_soft":
        scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True, square_light=True)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=2, max_depth=3, bias=1e-2, light_n=2)
    elif kind == "whitted_reflect":
        scene, cam = scenes.whitted_scene()
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=3, flags=abi.FLAG_WHITTED_REFLECT)
    else:
        scene, cam = scenes.whitted_scene()
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=2, flags=abi.FLAG_GLOBAL_LIGHTING, global_samples=6)
    desc = scene.frame(0.2)
    flags = kw.pop("flags", 0)
    out = []
    with gpu.DeviceScene(desc) as ds:
        for serial in (True, False):
            if not serial:
                monkeypatch.setenv("RTX_TWO_LANE_MIN", "128")
                monkeypatch.setenv("RTX_MAX_PATHS", "8192")
            rd = scenes.render_desc(150, 90, camera=cam, flags=flags | (abi.FLAG_SERIAL_CHUNKS if serial else 0), **kw)
            img, st = ds.render(rd)
            out.append((img, st))
    _same_ray_counts(out[0][1], out[1][1])
    if kind == "whitted_reflect":       # several rays add into one sample slot, in any order
        assert rel_err(out[0][0], out[1][0]).max() < 1e-6
    else:
        assert np.array_equal(out[0][0], out[1][0])
    assert np.isfinite(out[0][0]).all() and out[0][0].max() > 0


@pytest.mark.parametrize("kind", ["path", "whitted_soft", "whitted_refract", "whitted_reflect", "gather", "sorted", "walks"])
def test_computed_camera_rays_equal_stored_ones(gpu, monkeypatch, kind):
    """Level 0 is rendered without a stored queue by default (its kernels compute the camera ray
    of their sample slot); RTX_FLAG_STORE_CAMERA_RAYS writes the rays with a generate kernel and
    reads them back.  Images, ray counts and traversal work are equal bit for bit, over several
    chunks and passes, whichever walk traces the rays."""
    flags = 0
    if kind in ("path", "sorted"):
        scene, cam = scenes.lens_scene()
        kw = dict(integrator=abi.INTEGRATOR_PATH, n=2, max_depth=8, bias=1e-4, passes=2)
        flags = abi.FLAG_SORT_MATERIAL if kind == "sorted" else 0
    elif kind == "whitted_soft":
        scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True, square_light=True)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=2, max_depth=3, bias=1e-2, light_n=2)
    elif kind == "whitted_refract":       # the refraction child continues the camera ray's random stream
        scene, cam = scenes.whitted_scene()
        scene.square_light((0.0, 9.0, 2.0), (0.0, -1.0, 0.0), 2.0, 2.0)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=2, max_depth=5, light_n=2)
    elif kind == "whitted_reflect":
        scene, cam = scenes.whitted_scene()
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=3)
        flags = abi.FLAG_WHITTED_REFLECT
    elif kind == "gather":
        scene, cam = scenes.whitted_scene()
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=2, global_samples=6)
        flags = abi.FLAG_GLOBAL_LIGHTING
    else:
        scene, cam = scenes.mesh_scene(slices=60, stacks=30)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=2)
    desc = scene.frame(0.2)
    monkeypatch.setenv("RTX_TWO_LANE_MIN", "128")
    monkeypatch.setenv("RTX_MAX_PATHS", "8192")
    walks = (0, abi.FLAG_KD_CULLED, abi.FLAG_REFERENCE_ORDER, abi.FLAG_LINEAR_SCAN) if kind == "walks" else (0,)
    with gpu.DeviceScene(desc) as ds:
        for walk in walks:
            f = flags | walk | abi.FLAG_COUNTERS
            a, sa = ds.render(scenes.render_desc(150, 90, camera=cam, flags=f, **kw))
            b, sb = ds.render(scenes.render_desc(150, 90, camera=cam, flags=f | abi.FLAG_STORE_CAMERA_RAYS, **kw))
            _same_ray_counts(sa, sb)
            for k in ("content_tests", "triangle_tests", "sphere_tests", "aabb_tests", "hits_accepted", "fallback_rays"):
                assert getattr(sa, k) == getattr(sb, k), k
            if kind == "whitted_reflect":       # several rays add into one sample slot, in any order
                assert rel_err(a, b).max() < 1e-6
            else:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert np.isfinite(a).all() and a.max() > 0


@pytest.mark.parametrize("kind", ["path", "path_1spp", "basic", "whitted_soft", "whitted_1spp", "whitted_refract", "gather"])
def test_stored_samples_equal_accumulated_ones(gpu, monkeypatch, kind):
    """A sample slot whose whole value comes from one thread is stored once (with one sub-sample
    per pixel: written to the framebuffer at once, no resolve launch); RTX_FLAG_ACCUMULATE_SAMPLES
    clears the slots, adds every term and resolves.  Same bits, over chunks, passes, tiles that
    are added to or overwritten, and with stored camera rays too."""
    flags = 0
    if kind in ("path", "path_1spp", "basic"):
        scene, cam = scenes.lens_scene()
        kw = dict(integrator=abi.INTEGRATOR_PATH, n=1 if kind == "path_1spp" else 2, max_depth=8, bias=1e-4, passes=3)
        flags = abi.FLAG_BASIC if kind == "basic" else 0
    elif kind in ("whitted_soft", "whitted_1spp"):
        scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True, square_light=True)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1 if kind == "whitted_1spp" else 2, max_depth=3, bias=1e-2,
                  light_n=2, passes=2)
    elif kind == "whitted_refract":       # children: terms of a slot come from several launches, nothing is stored
        scene, cam = scenes.whitted_scene()
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=4)
    else:
        scene, cam = scenes.mesh_scene(slices=40, stacks=20)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=2, global_samples=4)
        flags = abi.FLAG_GLOBAL_LIGHTING
    desc = scene.frame(0.2)
    monkeypatch.setenv("RTX_TWO_LANE_MIN", "128")
    monkeypatch.setenv("RTX_MAX_PATHS", "8192")
    with gpu.DeviceScene(desc) as ds:
        for extra in (0, abi.FLAG_OVERWRITE, abi.FLAG_STORE_CAMERA_RAYS):
            junk = np.full((90, 150, 3), 0.25, np.float32)
            a, sa = ds.render(scenes.render_desc(150, 90, camera=cam, flags=flags | extra, **kw), accum=junk.copy())
            b, sb = ds.render(scenes.render_desc(150, 90, camera=cam, flags=flags | extra | abi.FLAG_ACCUMULATE_SAMPLES, **kw),
                              accum=junk.copy())
            _same_ray_counts(sa, sb)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), extra
            assert np.isfinite(a).all() and a.max() > 0.25
        # a tile inside a larger frame
        rd = scenes.render_desc(150, 90, camera=cam, flags=flags, tile=(40, 24, 70, 50), **kw)
        a, _ = ds.render(rd)
        rd.flags |= abi.FLAG_ACCUMULATE_SAMPLES
        b, _ = ds.render(rd)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("kind", ["path", "whitted_soft", "whitted_1spp"])
def test_interleaved_row_tile_equals_the_rows_of_the_frame(gpu, monkeypatch, kind):
    """tile_row_group / tile_row_stride: a tile made of every stride-th group of rows (one rank's
    share of a frame, rendered by one call) holds exactly those rows of the full-frame image,
    packed; whatever the group size, the chunking and the sample storage mode."""
    if kind == "path":
        scene, cam = scenes.lens_scene()
        kw = dict(integrator=abi.INTEGRATOR_PATH, n=2, max_depth=6, bias=1e-4, passes=2)
    else:
        scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True, square_light=True)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1 if kind == "whitted_1spp" else 2, max_depth=3, bias=1e-2, light_n=2)
    desc = scene.frame(0.2)
    Wd, Ht = 150, 91
    with gpu.DeviceScene(desc) as ds:
        full, sf = ds.render(scenes.render_desc(Wd, Ht, camera=cam, **kw))
        monkeypatch.setenv("RTX_TWO_LANE_MIN", "128")
        monkeypatch.setenv("RTX_MAX_PATHS", "8192")
        rays = 0
        for group, world in ((8, 3), (5, 2), (1, 4)):
            from raytracer_amd import dist as rdist
            seen = np.zeros(Ht, int)
            for rank in range(world):
                y0, rows, stride = rdist.rank_rows(Ht, world, rank, group)
                if rows == 0:
                    continue
                idx = np.array([y0 + (k // group) * stride + k % group for k in range(rows)])
                seen[idx] += 1
                for extra in (0, abi.FLAG_ACCUMULATE_SAMPLES | abi.FLAG_STORE_CAMERA_RAYS):
                    rd = scenes.render_desc(Wd, Ht, camera=cam, tile=(0, y0, Wd, rows), row_group=group, row_stride=stride,
                                            flags=extra, **kw)
                    got, st = ds.render(rd)
                    assert np.array_equal(got.view(np.uint32), full[idx].view(np.uint32)), (group, world, rank, extra)
                if group == 8:
                    rays += st.rays
            assert (seen == 1).all()
        assert rays == sf.rays          # the ranks' shares add up to the frame's rays
        # a stride below the group, or rows past the frame, are refused
        bad = scenes.render_desc(Wd, Ht, camera=cam, tile=(0, 0, Wd, 16), row_group=8, row_stride=4, **kw)
        with pytest.raises(RuntimeError):
            ds.render(bad)
        bad = scenes.render_desc(Wd, Ht, camera=cam, tile=(0, 80, Wd, 16), row_group=8, row_stride=16, **kw)
        with pytest.raises(RuntimeError):
            ds.render(bad)


@pytest.mark.parametrize("kind", ["point_refract", "directional_mesh", "walks"])
def test_light_rays_recomputed_from_their_origin_equal_stored_ones(gpu, monkeypatch, kind):
    """One Point or Directional light: a light ray is a function of its origin, so by default the
    shadow queue holds origins only and the traversal and the second shading half recompute
    direction and light distance; RTX_FLAG_STORE_LIGHT_RAYS stores them.  Same bits, same work."""
    if kind == "point_refract":
        scene, cam = scenes.whitted_scene()
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=2, max_depth=5)
    else:
        scene, cam = scenes.mesh_scene(slices=60, stacks=30)
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=1, max_depth=2, bias=1e-2)
    desc = scene.frame(0.2)
    monkeypatch.setenv("RTX_TWO_LANE_MIN", "128")
    monkeypatch.setenv("RTX_MAX_PATHS", "8192")
    walks = (0, abi.FLAG_KD_CULLED, abi.FLAG_REFERENCE_ORDER) if kind == "walks" else (0,)
    with gpu.DeviceScene(desc) as ds:
        for walk in walks:
            f = walk | abi.FLAG_COUNTERS
            a, sa = ds.render(scenes.render_desc(150, 90, camera=cam, flags=f, **kw))
            b, sb = ds.render(scenes.render_desc(150, 90, camera=cam, flags=f | abi.FLAG_STORE_LIGHT_RAYS, **kw))
            _same_ray_counts(sa, sb)
            assert sa.rays_shadow > 0
            for k in ("content_tests", "triangle_tests", "sphere_tests", "aabb_tests", "hits_accepted", "fallback_rays",
                      "branch_visits", "leaf_visits"):
                assert getattr(sa, k) == getattr(sb, k), k
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert np.isfinite(a).all() and a.max() > 0


def test_overwrite_flag_equals_cleared_buffer(gpu):
    """RTX_FLAG_OVERWRITE: the first pass stores, later passes add; a buffer full of other
    values gives the image a cleared buffer gives."""
    scene, cam = scenes.whitted_scene()
    desc = scene.frame(0.0)
    kw = dict(integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=3, passes=3, seed=2, clear=(0.0, 0.0, 0.0))
    with gpu.DeviceScene(desc) as ds:
        want, _ = ds.render(scenes.render_desc(90, 60, **kw))
        junk = np.full((60, 90, 3), -7.5, np.float32)
        got, _ = ds.render(scenes.render_desc(90, 60, flags=abi.FLAG_OVERWRITE, **kw), accum=junk)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_linear_scan_equals_kd_image(gpu):
    scene, cam = scenes.lens_scene()
    desc = scene.frame(0.0)
    a = scenes.render_desc(80, 60, integrator=abi.INTEGRATOR_WHITTED, camera=cam, max_depth=2)
    b = scenes.render_desc(80, 60, integrator=abi.INTEGRATOR_WHITTED, camera=cam, max_depth=2,
                           flags=abi.FLAG_LINEAR_SCAN)
    with gpu.DeviceScene(desc) as ds:
        ia, _ = ds.render(a)
        ib, _ = ds.render(b)
    ref, _, _, _ = orc.render(desc, b)
    assert rel_err(ib, ref).max() < REL_TOL
    assert psnr(ia, ib) >= 60.0


@pytest.mark.parametrize("name", ["live", "lens", "mesh_ggx"])
def test_material_sorted_shading_gives_identical_images(gpu, name):
    """RTX_FLAG_SORT_MATERIAL only changes the order in which a level's rays are shaded:
    every path still adds its one terminal value to its own sample slot."""
    if name == "live":
        scene, cam = scenes.live_scene()
    elif name == "lens":
        scene, cam = scenes.lens_scene()
    else:
        scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True)
        scene.sphere((0.0, 10.0, 0.0), 2.0, scene.material("PlainWhiteTees", emission=(8.0, 8.0, 8.0)))
    desc = scene.frame(0.3)
    imgs = []
    with gpu.DeviceScene(desc) as ds:
        for fl in (0, abi.FLAG_SORT_MATERIAL):
            rd = scenes.render_desc(200, 120, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=6, bias=1e-4,
                                    flags=fl, passes=2)
            img, st = ds.render(rd)
            imgs.append((img, st.rays))
    assert imgs[0][1] == imgs[1][1]
    assert np.array_equal(imgs[0][0], imgs[1][0])
    assert np.isfinite(imgs[0][0]).all() and imgs[0][0].max() > 0


def test_present_matches_oracle_and_writes_png(gpu, tmp_path):
    """rtx_present (cbrt tonemap in FP32, sRGB framebuffer bytes, vertical flip) against the
    oracle's restatement; the device's cbrtf may differ from glibc's in the last bit, which can
    move a value across a rounding boundary: at most one 8-bit step, on very few pixels.  Without
    the tonemap the bytes are equal."""
    from PIL import Image
    from raytracer_amd import output
    scene, cam = scenes.live_scene()
    desc = scene.frame(0.2)
    rd = scenes.render_desc(160, 100, integrator=abi.INTEGRATOR_PATH, camera=cam, n=2, max_depth=4, bias=1e-4, passes=3)
    with gpu.DeviceScene(desc) as ds:
        img, _ = ds.render(rd)
        edge = np.array([[[0.0, 1.0, 8.0], [-1.0, np.nan, 3 * 0.0031308]]], dtype=np.float32)
        for accum, passes in ((img, 3), (edge, 3)):
            for tonemap in (abi.TONEMAP_CBRT, abi.TONEMAP_NONE):
                for flip in (True, False):
                    got = ds.present(accum, passes=passes, tonemap=tonemap, flip=flip)
                    want = orc.present(accum, passes=passes, tonemap=tonemap, flip=flip)
                    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
                    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())
                    if tonemap == abi.TONEMAP_NONE:     # the transfer function's steps come from the host's powf
                        assert diff.max() == 0
        fw = output.FrameWriter(root=str(tmp_path / "out"))
        path = fw.save(ds, 0, img, passes=3)
        back = np.asarray(Image.open(path).convert("RGB"))
        assert np.array_equal(back, ds.present(img, passes=3))
        assert back.shape == (100, 160, 3) and back.max() > 50


@pytest.mark.parametrize("mesh", ["backpack", None], ids=["backpack", "procedural"])
def test_full_size_bench_workload_properties(gpu, mesh):
    """The bench workload at its full size (the reference's backpack model, 67 908 triangles, or
    the procedural 60 024-triangle mesh; 1920x1080, Whitted + directional light), where the CPU
    oracle would take minutes: size-independent properties instead.
    The three walks must give the same image bit for bit and the same ray counts; four tiles
    rendered separately must compose to the whole frame; a second pass must add exactly the
    first pass's image again (no random numbers on this path)."""
    import torch
    scene, _ = scenes.mesh_scene(mesh=mesh)
    desc = scene.frame(0.0)
    cam = scenes.orbit_camera(0.4)
    W, H = 1920, 1080

    def rd(flags=0, tile=None, passes=1):
        return scenes.render_desc(W, H, integrator=abi.INTEGRATOR_WHITTED, camera=cam, n=1, max_depth=3, bias=1e-2,
                                  clear=(0.21, 0.78, 0.95), flags=flags, tile=tile, passes=passes)

    with gpu.DeviceScene(desc, static=(mesh is not None)) as ds:      # the bench creates its scene as static
        fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        st_bvh = ds.render_device(rd(), fb.data_ptr())
        img_bvh = fb.clone()
        for flags in (abi.FLAG_REFERENCE_ORDER, abi.FLAG_KD_CULLED):
            fb.zero_()
            st = ds.render_device(rd(flags), fb.data_ptr())
            assert (st.rays_primary, st.rays_shadow) == (st_bvh.rays_primary, st_bvh.rays_shadow)
            assert torch.equal(fb, img_bvh)
        assert st_bvh.rays_primary == W * H and 0 < st_bvh.rays_shadow < W * H
        # tiles
        whole = torch.zeros_like(img_bvh)
        for (x0, y0) in ((0, 0), (960, 0), (0, 536), (960, 536)):
            w, h = 960, (536 if y0 == 0 else H - 536)
            part = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
            ds.render_device(rd(tile=(x0, y0, w, h)), part.data_ptr())
            whole[y0:y0 + h, x0:x0 + w] = part
        assert torch.equal(whole, img_bvh)
        # passes accumulate in FP32, pass by pass
        fb.zero_()
        ds.render_device(rd(passes=2), fb.data_ptr())
        assert torch.equal(fb, img_bvh + img_bvh)
        assert torch.isfinite(img_bvh).all() and float(img_bvh.max()) > 0.5


def test_degenerate_scenes(gpu):
    """No shapes, no lights, lights without shapes, several lights of all three kinds (the
    running `direct /= samples` of Raytracer-Noah-Desktop.cpp:498), with and without the gather:
    ray counts equal the oracle's and the images match."""
    from raytracer_amd.host import HostScene, make_camera
    cam = make_camera((0.0, 1.0, 5.0), (0.0, 0.0, 0.0))

    def empty():
        return HostScene()

    def only_light():
        s = HostScene()
        s.point_light((0.0, 5.0, 0.0))
        return s

    def unlit_sphere():
        s = HostScene()
        s.sphere((0.0, 0.0, 0.0), 1.0, s.material("Red"))
        return s

    def three_lights():
        s = unlit_sphere()
        s.point_light((0.0, 5.0, 0.0))
        s.directional_light((-1.0, -1.0, -1.0))
        s.square_light((0.0, 5.0, 0.0), (0.0, -1.0, 0.0), 1.0, 1.0)
        return s

    for make, kw in ((empty, {}), (only_light, {}), (unlit_sphere, {}), (three_lights, {}),
                     (three_lights, dict(flags=abi.FLAG_GLOBAL_LIGHTING, global_samples=3))):
        scene = make()                  # owns the arrays the description points to
        desc = scene.frame(0.0)
        for integrator in (abi.INTEGRATOR_WHITTED, abi.INTEGRATOR_PATH):
            rd = scenes.render_desc(33, 17, integrator=integrator, camera=cam, n=2, light_n=2, max_depth=2,
                                    clear=(0.1, 0.2, 0.3), **kw)
            img, stats, ref, ref_stats = _both(gpu, desc, rd)
            _same_ray_counts(stats, ref_stats)
            assert stats.rays_primary == 33 * 17 * 4
            assert np.isfinite(img).all() and rel_err(img, ref).max() < REL_TOL, make.__name__


def _interleaved_rows_equal_the_frames(ds, rd_of, whole, W, H, ranks=4, group=8):
    """Every rank's share of the frame -- the groups of `group` rows g with g % ranks == rank, one
    tile of interleaved rows rendered in one call (dist.rank_rows) -- equals those rows of the frame
    rendered at once; returns the rays of all shares together."""
    import torch
    from raytracer_amd import dist as rdist
    rays = 0
    for rank in range(ranks):
        y0, rows, stride = rdist.rank_rows(H, ranks, rank, group)
        part = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda")
        rays += ds.render_device(rd_of(tile=(0, y0, W, rows), row_group=group, row_stride=stride), part.data_ptr()).rays
        frame_rows = torch.cat([torch.arange(y, min(y + group, H)) for y in range(y0, H, stride)]).to("cuda")
        assert torch.equal(part.view(torch.int32), whole[frame_rows].view(torch.int32)), rank
    return rays


@pytest.mark.parametrize("mesh", ["backpack", None], ids=["backpack", "procedural"])
def test_full_size_16spp_soft_shadow_frame_properties(gpu, mesh):
    """The 1080p 16-spp SquareLight frame of the bench's second leg at its full size (33 M
    camera rays, ~87 M shadow rays; minutes for the CPU oracle), on the scene the bench times
    (backpack model, 4096 x 4096 albedo map) and on the procedural mesh: the BVH walk, the culled KD
    walk and the reference-order walk must give the same image bit for bit and the same ray counts,
    the frame assembled from row bands and from interleaved groups of rows (the multi-GPU
    partition, here on one rank) must equal the frame rendered at once, and the image must be
    finite and lit."""
    import torch
    from raytracer_amd import dist as rdist
    scene = scenes.frame16_scene("whitted", mesh=mesh, texture_size=4096 if mesh else 256)
    desc = scene.frame(0.0)
    W, H = 1920, 1080

    def rd(flags=0, tile=None, row_group=0, row_stride=0):
        return scenes.frame16_render_desc("whitted", W, H, 6, flags=flags, tile=tile, row_group=row_group, row_stride=row_stride)

    with gpu.DeviceScene(desc, static=(mesh is not None)) as ds:
        fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        st = ds.render_device(rd(), fb.data_ptr())
        whole = fb.clone()
        assert st.rays_primary == W * H * 16 and st.rays_shadow % 4 == 0 and st.rays_shadow > 0
        for flags in (abi.FLAG_KD_CULLED, abi.FLAG_REFERENCE_ORDER):
            fb.zero_()
            st2 = ds.render_device(rd(flags), fb.data_ptr())
            assert (st2.rays_primary, st2.rays_shadow) == (st.rays_primary, st.rays_shadow)
            assert torch.equal(fb.view(torch.int32), whole.view(torch.int32)), flags
        rays = [0]

        def render_rect(x0, y0, w, h, out):
            rays[0] += ds.render_device(rd(tile=(x0, y0, w, h)), out.data_ptr()).rays
        banded = rdist.render_frame_bands(render_rect, W, H, torch.device("cuda"), band_rows=264)
        assert torch.equal(banded, whole) and rays[0] == st.rays
        assert _interleaved_rows_equal_the_frames(ds, rd, whole, W, H) == st.rays
        assert torch.isfinite(whole).all() and float(whole.max()) > 0.1


def test_full_size_16spp_path_ggx_frame_properties(gpu):
    """The 1080p 16-spp pathTrace frame of the bench (backpack model with its 4096 x 4096 albedo map,
    CookTorance / GGX Gp bounces to depth 8, emissive sphere; ~60 M rays) at its full size: the
    three walks give the same image bit for bit and the same ray counts, four tiles rendered
    separately compose to the frame, every rank's tile of interleaved 8-row groups equals its rows
    of the frame, and the image is finite and lit."""
    import torch
    scene = scenes.frame16_scene("path")
    desc = scene.frame(0.0)
    W, H = 1920, 1080

    def rd(flags=0, tile=None, row_group=0, row_stride=0):
        return scenes.frame16_render_desc("path", W, H, 4, flags=flags, tile=tile, row_group=row_group, row_stride=row_stride)

    with gpu.DeviceScene(desc, static=True) as ds:
        fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        st = ds.render_device(rd(), fb.data_ptr())
        whole = fb.clone()
        assert st.rays_primary == W * H * 16 and st.rays_secondary > 0.5 * st.rays_primary and st.rays_shadow == 0
        for flags in (abi.FLAG_KD_CULLED, abi.FLAG_REFERENCE_ORDER):
            fb.zero_()
            st2 = ds.render_device(rd(flags), fb.data_ptr())
            assert (st2.rays_primary, st2.rays_secondary) == (st.rays_primary, st.rays_secondary)
            assert torch.equal(fb.view(torch.int32), whole.view(torch.int32)), flags
        rays = 0
        for (x0, y0) in ((0, 0), (960, 0), (0, 536), (960, 536)):
            w, h = 960, (536 if y0 == 0 else H - 536)
            part = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
            rays += ds.render_device(rd(tile=(x0, y0, w, h)), part.data_ptr()).rays
            assert torch.equal(part.view(torch.int32), whole[y0:y0 + h, x0:x0 + w].contiguous().view(torch.int32))
        assert rays == st.rays
        assert _interleaved_rows_equal_the_frames(ds, rd, whole, W, H, ranks=8) == st.rays
        assert torch.isfinite(whole).all() and float(whole.max()) > 0.1


def test_full_size_path_depth8_lens_scene_properties(gpu):
    """Config 3 at 1080p, path depth 8: the three walks agree bit for bit, and two passes
    rendered in one call equal the same two passes rendered one call each (FP32 accumulation
    in pass order, Raytracer.cpp:710)."""
    import torch
    scene, cam = scenes.lens_scene()
    desc = scene.frame(0.0)
    W, H = 1920, 1080

    def rd(flags=0, passes=1, first_pass=0):
        return scenes.render_desc(W, H, integrator=abi.INTEGRATOR_PATH, camera=cam, n=1, max_depth=8, bias=1e-4,
                                  clear=(0.21, 0.78, 0.95), seed=2, flags=flags, passes=passes, first_pass=first_pass)

    with gpu.DeviceScene(desc) as ds:
        fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        st = ds.render_device(rd(passes=2), fb.data_ptr())
        both = fb.clone()
        for flags in (abi.FLAG_REFERENCE_ORDER, abi.FLAG_KD_CULLED):
            fb.zero_()
            s2 = ds.render_device(rd(flags, passes=2), fb.data_ptr())
            assert (s2.rays_primary, s2.rays_secondary) == (st.rays_primary, st.rays_secondary)
            assert torch.equal(fb, both)
        fb.zero_()
        ds.render_device(rd(first_pass=0), fb.data_ptr())
        ds.render_device(rd(first_pass=1), fb.data_ptr())
        assert torch.equal(fb, both)
        assert st.rays_primary == 2 * W * H and st.rays_secondary > st.rays_primary
        assert torch.isfinite(both).all() and float(both.max()) > 0.1


def test_sequence_prefetch_gives_the_same_frames(gpu):
    """Animated frames with the next frame's scene prepared on a worker thread while the
    current one renders: same images as preparing every frame in line, and an early stop
    leaves nothing behind."""
    from raytracer_amd import sequence
    s = scenes.animated_scene(copies=1, slices=40, stacks=20)
    times = [f / 30.0 for f in range(1, 5)]
    frames = {}
    for prefetch in (False, True):
        for k, t, ds, host_s in sequence.device_scenes(s, times, prefetch=prefetch):
            rd = scenes.render_desc(96, 54, integrator=abi.INTEGRATOR_PATH, n=2, max_depth=4, bias=1e-3, frame=k,
                                    camera=scenes.orbit_camera(t, radius=30.0, height=6.0))
            frames[(prefetch, k)], _ = ds.render(rd)
            assert host_s > 0.0
    for k in range(len(times)):
        assert np.array_equal(frames[(False, k)], frames[(True, k)])
    assert not np.array_equal(frames[(True, 0)], frames[(True, 3)])      # the emissive spheres moved
    for k, t, ds, _ in sequence.device_scenes(s, times):
        if k == 1:
            break
    # the scene is still usable afterwards
    desc = s.frame(times[0])
    with gpu.DeviceScene(desc) as ds:
        rd = scenes.render_desc(96, 54, integrator=abi.INTEGRATOR_PATH, n=2, max_depth=4, bias=1e-3, frame=0,
                                camera=scenes.orbit_camera(times[0], radius=30.0, height=6.0))
        again, _ = ds.render(rd)
    assert np.array_equal(again, frames[(True, 0)])


@pytest.mark.parametrize("kind", ["path", "path_sorted_shading", "whitted_refract", "noalloc"])
def test_direction_binned_walk_order_gives_identical_images(gpu, monkeypatch, kind):
    """Levels below the camera's are walked in the order of the rays' direction bins (forced here at
    a small size); RTX_FLAG_QUEUE_ORDER walks them as queued.  Same hits per ray, so the same image
    bit for bit, the same ray counts and the same traversal work, over chunks, passes and lanes."""
    import torch
    flags = abi.FLAG_COUNTERS
    if kind in ("path", "path_sorted_shading", "noalloc"):
        scene, cam = scenes.mesh_scene(slices=60, stacks=30, ggx=True)
        scene.sphere((0.0, 10.0, 0.0), 2.0, scene.material("PlainWhiteTees", emission=(8.0, 8.0, 8.0)))
        kw = dict(integrator=abi.INTEGRATOR_PATH, n=2, max_depth=6, bias=1e-3, passes=2, seed=8)
        if kind == "path_sorted_shading":
            flags |= abi.FLAG_SORT_MATERIAL
    else:
        scene, cam = scenes.whitted_scene()
        kw = dict(integrator=abi.INTEGRATOR_WHITTED, n=2, max_depth=5)
    desc = scene.frame(0.2)
    monkeypatch.setenv("RTX_BIN_RAYS_MIN", "1")
    monkeypatch.setenv("RTX_BIN_SHAPES_MIN", "0")
    monkeypatch.setenv("RTX_TWO_LANE_MIN", "128")
    monkeypatch.setenv("RTX_MAX_PATHS", "16384")
    with gpu.DeviceScene(desc) as ds:
        if kind == "noalloc":
            rd = scenes.render_desc(150, 90, camera=cam, **kw)
            a = torch.zeros((90, 150, 3), dtype=torch.float32, device="cuda")
            b = torch.zeros_like(a)
            ds.render_device_noalloc(rd, a.data_ptr())
            rd.flags |= abi.FLAG_QUEUE_ORDER
            ds.render_device_noalloc(rd, b.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and float(a.max()) > 0
            return
        a, sa = ds.render(scenes.render_desc(150, 90, camera=cam, flags=flags, **kw))
        b, sb = ds.render(scenes.render_desc(150, 90, camera=cam, flags=flags | abi.FLAG_QUEUE_ORDER, **kw))
    _same_ray_counts(sa, sb)
    assert sa.rays_secondary > 0
    for k in ("content_tests", "triangle_tests", "sphere_tests", "aabb_tests", "hits_accepted", "fallback_rays"):
        assert getattr(sa, k) == getattr(sb, k), k
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.isfinite(a).all() and a.max() > 0
