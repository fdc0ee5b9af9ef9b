"""GPU parity of batched shootRay (rtx_trace_closest) against the CPU oracle:
primary-hit ids, depths, full hit records, KD leaf sequences and work counters
must match bit for bit."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes
from oracle import binding as orc
from helpers import camera_rays_np, compare_trace, id_hash

pytestmark = pytest.mark.gpu


def _random_rays(rng, n, box=12.0):
    o = rng.uniform(-box, box, size=(n, 3))
    d = rng.normal(size=(n, 3))
    return o, d


@pytest.mark.parametrize("t,want", [(0.0, 0x316a73d0000316fa), (1.7, 0x1157010466b0eb89)])
def test_live_scene_hit_hash(gpu, t, want):
    """Known-answer hash of the primary-hit ids of the reference's live scene."""
    scene, cam = scenes.live_scene()
    desc = scene.frame(t)
    rd = scenes.render_desc(320, 240, camera=cam)
    o, d = camera_rays_np(rd, sub_sample=(0, 0))
    with gpu.DeviceScene(desc) as ds:
        got = ds.trace_closest(o, d, leaf_seq_cap=16, want_counters=True)
    assert id_hash(got["prim"]) == want
    compare_trace(got, orc.trace_closest(desc, o, d, leaf_seq_cap=16, want_counters=True), seq_cap=16, desc=desc)


def test_live_scene_random_and_edge_rays(gpu):
    scene, _ = scenes.live_scene()
    desc = scene.frame(0.4)
    rng = np.random.default_rng(1)
    o, d = _random_rays(rng, 20000, box=30.0)
    # axis-parallel rays (infinite slabs), rays starting on box planes, rays that miss everything
    axis = np.eye(3)[rng.integers(0, 3, 3000)] * rng.choice([-1.0, 1.0], size=(3000, 1))
    o2 = rng.uniform(-20, 20, size=(3000, 3))
    o2[:500, 1] = 0.0            # on the floor plane
    o2[500:1000, 0] = -1000.0    # on the scene box
    o3 = rng.uniform(-5, 5, size=(1000, 3)) + np.array([0, 5000.0, 0])
    d3 = np.tile([0.0, 1.0, 0.0], (1000, 1))
    o = np.vstack([o, o2, o3])
    d = np.vstack([d, axis, d3])
    with gpu.DeviceScene(desc) as ds:
        got = ds.trace_closest(o, d, leaf_seq_cap=24, want_counters=True)
        lin = ds.trace_closest(o, d, flags=abi.FLAG_LINEAR_SCAN, want_counters=True)
    compare_trace(got, orc.trace_closest(desc, o, d, leaf_seq_cap=24, want_counters=True), seq_cap=24, desc=desc)
    compare_trace(lin, orc.trace_closest(desc, o, d, flags=abi.FLAG_LINEAR_SCAN, want_counters=True), desc=desc)
    assert (got["prim"][-1000:] == -1).all()


def test_mesh_scene_primary_and_shadow_like_rays(gpu):
    """Triangle soup through the KD tree (about 10k triangles so the oracle stays fast)."""
    scene, cam = scenes.mesh_scene(slices=100, stacks=50)
    desc = scene.frame(0.0)
    rd = scenes.render_desc(192, 108, camera=cam)
    o, d = camera_rays_np(rd)
    rng = np.random.default_rng(2)
    o2, d2 = _random_rays(rng, 30000, box=6.0)
    o = np.vstack([o, o2])
    d = np.vstack([d, d2])
    with gpu.DeviceScene(desc) as ds:
        got = ds.trace_closest(o, d, leaf_seq_cap=64, want_counters=True)
    want = orc.trace_closest(desc, o, d, leaf_seq_cap=64, want_counters=True)
    compare_trace(got, want, seq_cap=64, desc=desc)
    assert (got["prim"] >= 0).mean() > 0.3


def test_lens_scene_biconvex(gpu):
    scene, cam = scenes.lens_scene()
    desc = scene.frame(0.0)
    assert desc.n_prims == desc.n_shapes + 2
    rd = scenes.render_desc(200, 150, camera=cam)
    o, d = camera_rays_np(rd)
    rng = np.random.default_rng(3)
    o2, d2 = _random_rays(rng, 20000, box=8.0)
    o = np.vstack([o, o2])
    d = np.vstack([d, d2])
    with gpu.DeviceScene(desc) as ds:
        got = ds.trace_closest(o, d, leaf_seq_cap=8, want_counters=True)
    want = orc.trace_closest(desc, o, d, leaf_seq_cap=8, want_counters=True)
    compare_trace(got, want, seq_cap=8, desc=desc)
    assert (got["prim"] >= desc.n_shapes).any(), "no ray reported an inner lens sphere"


def test_empty_and_single_ray(gpu):
    scene, _ = scenes.whitted_scene()
    desc = scene.frame(0.0)
    with gpu.DeviceScene(desc) as ds:
        got = ds.trace_closest(np.zeros((0, 3)), np.zeros((0, 3)))
        assert got["prim"].size == 0
        one = ds.trace_closest(np.array([[0.0, 3.0, 9.0]]), np.array([[0.0, -0.2, -1.0]]), want_counters=True)
    want = orc.trace_closest(desc, np.array([[0.0, 3.0, 9.0]]), np.array([[0.0, -0.2, -1.0]]), want_counters=True)
    compare_trace(one, want, desc=desc)


def test_malformed_scene_is_rejected(gpu):
    scene, _ = scenes.whitted_scene()
    desc = scene.frame(0.0)
    bad = abi.SceneDesc()
    import ctypes as C
    C.memmove(C.byref(bad), C.byref(desc), C.sizeof(bad))
    bad.n_leaf_refs = desc.n_leaf_refs
    refs = np.ctypeslib.as_array(desc.leaf_refs, shape=(desc.n_leaf_refs,)).copy()
    refs[0] = desc.n_shapes + 5
    bad.leaf_refs = refs.ctypes.data_as(C.POINTER(C.c_uint32))
    with pytest.raises(gpu.RtxError) as e:
        gpu.DeviceScene(bad)
    assert e.value.code == abi.RTX_ERR_INVALID


def _scenes_for_walks():
    yield "live", lambda: scenes.live_scene()[0], 0.7, 30.0
    yield "mesh", lambda: scenes.mesh_scene(slices=100, stacks=50)[0], 0.0, 8.0
    yield "lens", lambda: scenes.lens_scene()[0], 0.0, 8.0
    yield "whitted", lambda: scenes.whitted_scene()[0], 0.0, 8.0


@pytest.mark.parametrize("name,make,t,box", list(_scenes_for_walks()), ids=[c[0] for c in _scenes_for_walks()])
@pytest.mark.parametrize("walk", ["bvh", "bvh_static", "kd_culled"])
def test_render_walks_find_the_reference_hit(gpu, name, make, t, box, walk):
    """The render path's walks (BVH candidates + membership verification; culled KD walk)
    must report exactly the hit the reference-order walk reports: same primitive id (ties
    included), same depth bits, same hit record -- on incoherent random rays, rays starting
    on surfaces, and axis-parallel rays (which take the reference walk internally)."""
    scene = make()
    desc = scene.frame(t)
    rng = np.random.default_rng(5)
    o, d = _random_rays(rng, 60000, box=box)
    # rays that start on geometry (like shadow and bounce rays do)
    with gpu.DeviceScene(desc, static=walk == "bvh_static") as ds:      # rtx_scene_create_static: list searches skipped
        first = ds.trace_closest(o, d)
        hit = first["prim"] >= 0
        o2 = first["position"][hit] + 1e-3 * first["normal"][hit]
        d2 = rng.normal(size=o2.shape)
        axis = np.eye(3)[rng.integers(0, 3, 2000)] * rng.choice([-1.0, 1.0], size=(2000, 1))
        o3 = rng.uniform(-box, box, size=(2000, 3))
        oo = np.vstack([o, o2, o3])
        dd = np.vstack([d, d2, axis])
        flags = abi.FLAG_RENDER_WALK | (abi.FLAG_KD_CULLED if walk == "kd_culled" else 0)
        got = ds.trace_closest(oo, dd, flags=flags)
    want = orc.trace_closest(desc, oo, dd)
    for key in ("prim", "depth", "position", "normal", "bary"):
        g, w = got[key], want[key]
        same = (g == w) | ((g != g) & (w != w)) if g.dtype.kind == "f" else g == w
        assert same.all(), f"{key}: {np.count_nonzero(~same)} of {same.size} differ"
    assert (want["prim"] >= 0).mean() > 0.05


def _tie_scene():
    """Random triangles each present twice (two shapes, identical vertices: every hit is an
    exact depth tie that only the reference's visit order resolves), in both windings (so rays
    from either side hit, and triangles behind a ray origin are hits too, Triangle.cpp has no
    t >= 0 test), plus two coplanar overlapping quads."""
    from raytracer_amd import host
    rng = np.random.default_rng(11)
    scene = host.HostScene()
    m = scene.material("PlainWhiteTees")
    n = 400
    c = rng.uniform(-5, 5, size=(n, 1, 3))
    tri = c + rng.normal(scale=0.8, size=(n, 3, 3))
    flipped = tri[:, ::-1, :]
    order = rng.permutation(4 * n)
    pos = np.concatenate([tri, tri, flipped, flipped])[order]
    nrm = np.tile([0.0, 1.0, 0.0], (4 * n, 3, 1))
    uv = np.tile([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]], (4 * n, 1, 1))
    scene.triangles(pos, nrm, uv, m)
    quad = np.array([[[-6, -6, 0.25], [6, -6, 0.25], [6, 6, 0.25]], [[-6, -6, 0.25], [6, 6, 0.25], [-6, 6, 0.25]],
                     [[-3, -6, 0.25], [9, -6, 0.25], [9, 6, 0.25]], [[-3, -6, 0.25], [9, 6, 0.25], [-3, 6, 0.25]]], dtype=np.float64)
    scene.triangles(quad, nrm[:4], uv[:4], m)
    return scene


@pytest.mark.parametrize("walk", ["bvh", "kd_culled"])
def test_exact_depth_ties_follow_the_reference_visit_order(gpu, walk):
    scene = _tie_scene()
    desc = scene.frame(0.0)
    rng = np.random.default_rng(12)
    o, d = _random_rays(rng, 40000, box=7.0)
    flags = abi.FLAG_RENDER_WALK | (abi.FLAG_KD_CULLED if walk == "kd_culled" else 0)
    with gpu.DeviceScene(desc) as ds:
        got = ds.trace_closest(o, d, flags=flags)
        ref = ds.trace_closest(o, d)
    want = orc.trace_closest(desc, o, d)
    assert np.array_equal(ref["prim"], want["prim"])
    assert np.array_equal(got["prim"], want["prim"]), f"{np.count_nonzero(got['prim'] != want['prim'])} ids differ"
    assert np.array_equal(got["depth"], want["depth"])
    assert (want["prim"] >= 0).mean() > 0.3


def test_bvh_walk_falls_back_on_ties_only(gpu):
    """Work counters of a render: on the tie scene the BVH walk must send rays through the
    reference-order walk (fallback_rays > 0) and still produce the reference image; on an
    ordinary mesh no ray needs it."""
    from raytracer_amd import host
    for make, expect in ((_tie_scene, True), (lambda: scenes.mesh_scene(slices=60, stacks=30)[0], False)):
        scene = make()
        if expect:
            scene.directional_light((-1.0, -1.0, -1.0))
        desc = scene.frame(0.0)
        cam = host.make_camera((0.0, 0.5, 12.0), (0.0, 0.0, 0.0))
        imgs = {}
        with gpu.DeviceScene(desc) as ds:
            for name, fl in (("bvh", 0), ("ref", abi.FLAG_REFERENCE_ORDER)):
                rd = scenes.render_desc(160, 120, camera=cam, max_depth=2, bias=1e-3, flags=fl | abi.FLAG_COUNTERS)
                imgs[name], st = ds.render(rd)
                if name == "bvh":
                    assert (st.fallback_rays > 0) == expect, st.fallback_rays
        assert np.array_equal(imgs["bvh"], imgs["ref"])


def test_bounce_rays_are_certified_without_walking_the_kd_tree(gpu):
    """A bounce ray starts 1e-3 off the surface it left, and Triangle::rayHit (Triangle.cpp:8-72, no
    t >= 0 test) reports front-facing triangles behind that origin as hits; their `origin + direction
    * depth` is no point of the ray.  The membership certificate must still settle them from the leaf
    cell alone: the KD-branch steps of its slow proofs stay a small fraction of the rays, and the image
    equals the reference-order one."""
    from raytracer_amd import host
    scene = scenes.mesh_scene(ggx=True, mesh="teapot")[0]
    scene.sphere((0.0, 10.0, 0.0), 2.0, scene.material("PlainWhiteTees", emission=(8.0, 8.0, 8.0)))
    desc = scene.frame(0.0)
    cam = host.make_camera((0.0, 0.5, 12.0), (0.0, 0.0, 0.0))
    imgs = {}
    with gpu.DeviceScene(desc, static=True) as ds:
        for name, fl in (("bvh", 0), ("ref", abi.FLAG_REFERENCE_ORDER)):
            rd = scenes.render_desc(240, 180, integrator=abi.INTEGRATOR_PATH, camera=cam, max_depth=4, bias=1e-3,
                                    flags=fl | abi.FLAG_COUNTERS)
            imgs[name], st = ds.render(rd)
            if name == "bvh":
                assert st.fallback_rays == 0
                assert st.rays > 60000
                assert st.branch_visits < 0.05 * st.rays, (st.branch_visits, st.rays)
    assert np.array_equal(imgs["bvh"], imgs["ref"])


def test_kd_tree_that_leaves_shapes_out_of_leaves_they_reach(gpu):
    """The reference's builder lists a shape in every leaf its box reaches, and for a scene created
    with rtx_scene_create_static the membership certificate skips the search of a shape's leaf list
    where the library has checked exactly that.
    A caller's tree need not be so: here a quarter of the leaves lose their last shape (the entry is
    overwritten with a copy of the leaf's first), so hits inside those cells belong to no visited leaf
    there.  The render walk must still report what the reference-order walk reports on this tree."""
    import ctypes as C
    scene = scenes.mesh_scene(slices=80, stacks=40)[0]
    desc = scene.frame(0.0)
    first = np.ctypeslib.as_array(desc.leaf_first, shape=(desc.n_leaves,))
    count = np.ctypeslib.as_array(desc.leaf_count, shape=(desc.n_leaves,))
    refs = np.ctypeslib.as_array(desc.leaf_refs, shape=(desc.n_leaf_refs,)).copy()
    rng = np.random.default_rng(21)
    cut = np.flatnonzero((count >= 2) & (rng.random(desc.n_leaves) < 0.25))
    refs[first[cut] + count[cut] - 1] = refs[first[cut]]
    assert cut.size > 100
    thin = abi.SceneDesc()
    C.memmove(C.byref(thin), C.byref(desc), C.sizeof(thin))
    thin.leaf_refs = refs.ctypes.data_as(C.POINTER(C.c_uint32))
    o, d = _random_rays(rng, 200000, box=4.0)
    with gpu.DeviceScene(thin, static=True) as ds:
        ref = ds.trace_closest(o, d)
        hit = ref["prim"] >= 0
        o2 = ref["position"][hit] + 1e-3 * ref["normal"][hit]
        oo = np.vstack([o, o2])
        dd = np.vstack([d, rng.normal(size=o2.shape)])
        ref = ds.trace_closest(oo, dd)
        got = ds.trace_closest(oo, dd, flags=abi.FLAG_RENDER_WALK)
    with gpu.DeviceScene(desc) as ds:
        whole = ds.trace_closest(oo, dd)
    assert np.array_equal(got["prim"], ref["prim"]), f"{np.count_nonzero(got['prim'] != ref['prim'])} ids differ"
    assert np.array_equal(got["depth"], ref["depth"])
    # the cut must matter: some rays see another shape (or none) than with the whole tree
    assert np.count_nonzero(whole["prim"] != ref["prim"]) > 20


@pytest.mark.parametrize("static", [False, True])
def test_single_leaf_kd_tree_that_omits_a_shape(gpu, static):
    """A KD tree that is one leaf (fewer than seven shapes, KDTree.cpp:69) and does not list one of
    the scene's shapes: the reference's walk never tests that shape, so rays aimed at it see what lies
    behind.  The candidate BVH is built over all shapes and proposes it; the certificate must refuse
    it (a tree of one leaf has no branch to locate a point with), and the render walk must report
    what the reference-order walk reports, in traces and in pictures."""
    import ctypes as C
    scene, cam = scenes.whitted_scene()
    desc = scene.frame(0.0)
    assert desc.n_branches == 0 and desc.n_leaves == 1 and desc.kd_root < 0
    count = np.array([desc.n_shapes - 1], dtype=np.uint32)         # the last shape (the mirror sphere) is left out
    thin = abi.SceneDesc()
    C.memmove(C.byref(thin), C.byref(desc), C.sizeof(thin))
    thin.leaf_count = count.ctypes.data_as(C.POINTER(C.c_uint32))
    rd = scenes.render_desc(160, 120, integrator=abi.INTEGRATOR_WHITTED, camera=cam, max_depth=2)
    o, d = orc.camera_rays(rd)
    with gpu.DeviceScene(thin, static=static) as ds:
        ref = ds.trace_closest(o, d)
        got = ds.trace_closest(o, d, flags=abi.FLAG_RENDER_WALK)
        img, st = ds.render(rd)
        rd.flags = abi.FLAG_REFERENCE_ORDER
        img_ref, st_ref = ds.render(rd)
    with gpu.DeviceScene(desc) as ds:
        whole = ds.trace_closest(o, d)
    want = orc.trace_closest(thin, o, d)
    assert np.array_equal(ref["prim"], want["prim"]) and np.array_equal(ref["depth"], want["depth"])
    assert np.array_equal(got["prim"], ref["prim"]), f"{np.count_nonzero(got['prim'] != ref['prim'])} ids differ"
    assert np.array_equal(got["depth"], ref["depth"])
    assert not np.any(ref["prim"] == desc.n_shapes - 1)
    assert np.count_nonzero(whole["prim"] == desc.n_shapes - 1) > 100, "the omitted shape should be in view"
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (st_ref.rays_primary, st_ref.rays_secondary, st_ref.rays_shadow)
    assert np.array_equal(img.view(np.uint32), img_ref.view(np.uint32))


def _degenerate_scene(offset, scale=1.0):
    """Geometry chosen to sit on KD cell faces and split planes and to stress the culls'
    margins: axis-aligned quads on integer planes (their boxes are flat and coincide with
    split values), needle and zero-area triangles, shapes listed twice, a sphere touching
    a wall -- everything translated by `offset` so that coordinates are large compared with
    the features."""
    from raytracer_amd import host
    rng = np.random.default_rng(21)
    scene = host.HostScene()
    m = scene.material("PlainWhiteTees")
    tris = []
    for axis in range(3):
        for plane in (-4.0, -1.0, 0.0, 2.0, 4.0):
            u, v = [a for a in range(3) if a != axis]
            for (u0, v0) in ((-4.0, -4.0), (0.0, -4.0), (-4.0, 0.0), (0.0, 0.0)):
                q = np.zeros((4, 3))
                q[:, axis] = plane
                q[:, u] = [u0, u0 + 4.0, u0 + 4.0, u0]
                q[:, v] = [v0, v0, v0 + 4.0, v0 + 4.0]
                tris += [q[[0, 1, 2]], q[[0, 2, 3]], q[[2, 1, 0]], q[[3, 2, 0]]]     # both windings
    c = rng.uniform(-4, 4, size=(150, 1, 3))
    needle = c + np.concatenate([np.zeros((150, 1, 3)), rng.normal(scale=1.0, size=(150, 1, 3)),
                                 rng.normal(scale=1e-9, size=(150, 1, 3))], axis=1)
    needle[:, 2] += needle[:, 1] - c[:, 0]                      # third vertex 1e-9 off the second
    tris += list(needle)
    zero = np.repeat(rng.uniform(-4, 4, size=(30, 1, 3)), 3, axis=1)
    tris += list(zero)
    small = rng.uniform(-4, 4, size=(300, 1, 3)) + rng.normal(scale=0.3, size=(300, 3, 3))
    tris += list(small) + list(small[:40])                      # forty shapes present twice
    pos = (np.array(tris) + np.asarray(offset)) * scale
    n = len(pos)
    nrm = np.tile([0.0, 1.0, 0.0], (n, 3, 1))
    uv = np.tile([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]], (n, 1, 1))
    scene.triangles(pos, nrm, uv, m)
    scene.sphere(tuple((np.asarray(offset) + [1.0, 1.0, 3.0]) * scale), 1.0 * scale, m)      # touches the plane z = 4 and x = 2
    return scene


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (5000.0, -3000.0, 7000.0)], ids=["origin", "far"])
@pytest.mark.parametrize("walk", ["bvh", "kd_culled"])
def test_degenerate_geometry_walks_match_reference(gpu, walk, offset):
    """Flat boxes on split planes, needles, zero-area triangles, duplicates, large coordinates:
    both render-path walks must still report the reference-order walk's hit bit for bit, for
    random rays, rays lying in the quads' planes, rays starting on the quads, nearly
    axis-parallel rays and exactly axis-parallel ones."""
    scene = _degenerate_scene(offset)
    desc = scene.frame(0.0)
    rng = np.random.default_rng(22)
    off = np.asarray(offset)
    o1, d1 = _random_rays(rng, 30000, box=6.0)
    # origins on the integer planes, directions inside those planes (grazing) or leaving them
    o2 = rng.uniform(-4, 4, size=(9000, 3))
    ax = rng.integers(0, 3, 9000)
    o2[np.arange(9000), ax] = rng.choice([-4.0, -1.0, 0.0, 2.0, 4.0], 9000)
    d2 = rng.normal(size=(9000, 3))
    d2[np.arange(3000), ax[:3000]] = 0.0                        # in-plane
    # nearly axis-parallel: one dominant component, the others ~1e-12 .. 1e-17
    d3_ = rng.normal(size=(6000, 3)) * 10.0 ** rng.uniform(-17, -12, size=(6000, 1))
    d3_[np.arange(6000), rng.integers(0, 3, 6000)] = rng.choice([-1.0, 1.0], 6000)
    o3 = rng.uniform(-5, 5, size=(6000, 3))
    d4 = np.eye(3)[rng.integers(0, 3, 2000)] * rng.choice([-1.0, 1.0], size=(2000, 1))
    o4 = np.round(rng.uniform(-5, 5, size=(2000, 3)))           # exactly on the integer planes
    oo = np.vstack([o1, o2, o3, o4]) + off
    dd = np.vstack([d1, d2, d3_, d4])
    flags = abi.FLAG_RENDER_WALK | (abi.FLAG_KD_CULLED if walk == "kd_culled" else 0)
    with gpu.DeviceScene(desc) as ds:
        got = ds.trace_closest(oo, dd, flags=flags)
        ref = ds.trace_closest(oo, dd)
    want = orc.trace_closest(desc, oo, dd)
    for key in ("prim", "depth", "position", "normal", "bary"):
        for name, res in (("reference-order", ref), (walk, got)):
            g, w = res[key], want[key]
            same = (g == w) | ((g != g) & (w != w)) if g.dtype.kind == "f" else g == w
            assert same.all(), f"{name} {key}: {np.count_nonzero(~same)} of {same.size} differ"
    assert (want["prim"] >= 0).mean() > 0.3


@pytest.mark.parametrize("case", ["huge scene", "tiny direction components", "far origins"])
def test_bvh_walk_beyond_the_range_of_its_single_precision_box_test(gpu, case):
    """The BVH walk tests child boxes in single precision with headroom checks: a scene whose
    coordinates exceed 2^100 is not walked that way at all, an axis with |1/d| > 2^60 or |o/d| > 2^100
    refuses no box.  In all three cases the walk must still report the reference-order walk's hit."""
    rng = np.random.default_rng(77)
    scale = 1e31 if case == "huge scene" else 1.0
    scene = _degenerate_scene((0.0, 0.0, 0.0), scale)
    desc = scene.frame(0.0)
    o, d = _random_rays(rng, 20000, box=6.0)
    if case == "tiny direction components":
        # one or two components of 1e-19 .. 1e-30 next to a dominant one: |1/d| up to 1e30
        d = rng.normal(size=(20000, 3)) * 10.0 ** rng.uniform(-30, -19, size=(20000, 1))
        d[np.arange(20000), rng.integers(0, 3, 20000)] = rng.choice([-1.0, 1.0], 20000)
    if case == "far origins":
        # origins 1e25 .. 1e33 away, aimed at the scene: |o/d| beyond 2^100 on some axes
        target = rng.uniform(-4, 4, size=(20000, 3))
        o = target + rng.normal(size=(20000, 3)) * 10.0 ** rng.uniform(25, 33, size=(20000, 1))
        d = target - o
    o = o * scale
    with gpu.DeviceScene(desc) as ds:
        got = ds.trace_closest(o, d, flags=abi.FLAG_RENDER_WALK)
        ref = ds.trace_closest(o, d)
    want = orc.trace_closest(desc, o, d)
    for key in ("prim", "depth"):
        for name, res in (("reference-order", ref), ("bvh", got)):
            g, w = res[key], want[key]
            same = (g == w) | ((g != g) & (w != w)) if g.dtype.kind == "f" else g == w
            assert same.all(), f"{case}: {name} {key}: {np.count_nonzero(~same)} of {same.size} differ"
    if case != "far origins":
        assert (want["prim"] >= 0).mean() > 0.2
