"""Host scene API: KD construction (KDTree.cpp:68-126), flattening, and the oracle's KD walk
against its own linear scan (the reference's KDTRACE toggle, Raytracer.cpp:114-118)."""
import os

import numpy as np
import pytest

from raytracer_amd import abi, scenes
from raytracer_amd.host import HostScene, desc_array
from oracle import binding as orc
from helpers import camera_rays_np


def _tree_arrays(desc):
    return dict(
        lesser=desc_array(desc, "branch_lesser", desc.n_branches, np.int32),
        greater=desc_array(desc, "branch_greater", desc.n_branches, np.int32),
        axis=desc_array(desc, "branch_axis", desc.n_branches, np.int32),
        value=desc_array(desc, "branch_value", desc.n_branches, np.float64),
        bmin=desc_array(desc, "branch_min", desc.n_branches * 3, np.float64).reshape(-1, 3),
        bmax=desc_array(desc, "branch_max", desc.n_branches * 3, np.float64).reshape(-1, 3),
        first=desc_array(desc, "leaf_first", desc.n_leaves, np.uint32),
        count=desc_array(desc, "leaf_count", desc.n_leaves, np.uint32),
        refs=desc_array(desc, "leaf_refs", desc.n_leaf_refs, np.uint32),
    )


def test_small_scene_is_a_single_leaf():
    """Fewer than 7 shapes -> buildKDTree returns a KDLeaf (KDTree.cpp:69)."""
    scene, _ = scenes.whitted_scene()
    desc = scene.frame(0.0)
    assert desc.kd_root == -1 and desc.n_branches == 0 and desc.n_leaves == 1
    assert desc_array(desc, "leaf_refs", desc.n_leaf_refs, np.uint32).tolist() == [0, 1, 2, 3]


def test_tree_structure_invariants():
    scene, _ = scenes.mesh_scene(slices=40, stacks=20)
    desc = scene.frame(0.0)
    t = _tree_arrays(desc)
    st = scene.kd_stats()
    assert st["depth"] <= 22 and st["branches"] == desc.n_branches
    # leaves are numbered in walk order (lesser first) and their ranges tile leaf_refs
    assert (t["first"][1:] == t["first"][:-1] + t["count"][:-1]).all()
    order = []

    def walk(node):
        if node < 0:
            order.append(~node)
            return
        ax, v = t["axis"][node], t["value"][node]
        # child boxes clip the parent box at the split value (KDTree.cpp:85-86,107-108)
        for child, lo in ((t["lesser"][node], True), (t["greater"][node], False)):
            if child >= 0:
                if lo:
                    assert t["bmax"][child][ax] == v
                else:
                    assert t["bmin"][child][ax] == v
                assert t["axis"][child] == (ax + 1) % 3
        walk(t["lesser"][node])
        walk(t["greater"][node])

    import sys
    sys.setrecursionlimit(10000)
    walk(desc.kd_root)
    assert order == list(range(desc.n_leaves))
    # every shape is referenced by at least one leaf; the root box contains the origin (AABB(0,0) + encompass)
    assert set(t["refs"].tolist()) == set(range(desc.n_shapes))
    assert (t["bmin"][0] <= 0).all() and (t["bmax"][0] >= 0).all()


def test_kd_walk_equals_linear_scan_up_to_depth_ties():
    scene, cam = scenes.mesh_scene(slices=60, stacks=30)
    desc = scene.frame(0.0)
    rd = scenes.render_desc(160, 90, camera=cam)
    o, d = camera_rays_np(rd)
    kd = orc.trace_closest(desc, o, d)
    lin = orc.trace_closest(desc, o, d, flags=abi.FLAG_LINEAR_SCAN)
    assert np.array_equal(kd["depth"], lin["depth"])
    differ = kd["prim"] != lin["prim"]
    assert differ.mean() < 0.01          # exact depth ties on shared edges go to the first shape in walk order


def test_animated_spheres_move_between_frames():
    scene, _ = scenes.live_scene()
    a = desc_array(scene.frame(0.0), "sphere_center", 24, np.float64).copy()
    b = desc_array(scene.frame(1.0), "sphere_center", 24, np.float64).copy()
    assert np.array_equal(a[:15], b[:15])            # noMovement spheres
    assert not np.array_equal(a[15:], b[15:])        # circle0/1/2
    assert np.allclose(b[15:].reshape(3, 3)[:, 1], 15.0)


def test_biconvex_hidden_spheres_and_box():
    s = HostScene()
    s.sphere((5, 0, 0), 1.0, s.material("Red"))
    s.biconvex((0.0, 2.0, 4.0), (0.0, 0.0, 2.0), 3.0, 50.0)
    desc = s.frame(0.0)
    assert desc.n_shapes == 2 and desc.n_prims == 4 and desc.n_lenses == 1
    ptype = desc_array(desc, "prim_type", 4, np.uint8).tolist()
    assert ptype == [abi.PRIM_SPHERE, abi.PRIM_BICONVEX, abi.PRIM_SPHERE, abi.PRIM_SPHERE]
    assert desc_array(desc, "lens_spheres", 2, np.uint32).tolist() == [2, 3]
    # d = n r^2 / (f (n-1)^2) with n = 1.54 (Biconvex.cpp:3-10,31); centres at origin -/+ forward*(r-d)
    d = 1.54 * 9.0 / (50.0 * 0.54 * 0.54)
    c = desc_array(desc, "sphere_center", 9, np.float64).reshape(3, 3)
    assert np.allclose(c[1], [0.0, 2.0, 4.0 - (3.0 - d)]) and np.allclose(c[2], [0.0, 2.0, 4.0 + (3.0 - d)])
    half = np.sqrt((3.0 - d / 2) * 8.0 * d) / 2.0
    bmin = desc_array(desc, "prim_aabb_min", 12, np.float64).reshape(4, 3)[1]
    assert np.allclose(bmin, np.array([0.0, 2.0, 4.0]) - half)
    # the glass of the inner spheres is the preset (Biconvex.cpp:30)
    glass = desc.materials[desc_array(desc, "prim_material", 4, np.uint32)[2]]
    assert (glass.ni, glass.transparency) == (1.54, 0.8)


def test_unknown_preset_is_an_error_not_a_crash():
    s = HostScene()
    with pytest.raises(RuntimeError):
        s.material("NoSuchPreset")


def test_texture_and_sampler_descriptors():
    s = HostScene()
    tex = s.texture(scenes.checker_texture(size=16, cells=2))
    m = s.material_custom((1, 1, 1), samplers={"color": (abi.SAMPLER_TEXTURE_VEC3, tex),
                                              "smoothness": (abi.SAMPLER_DOUBLE_CHECKERS, 0)})
    s.sphere((0, 0, 0), 1.0, m)
    desc = s.frame(0.0)
    assert desc.n_textures == 1 and desc.textures[0].width == 16 and desc.n_texel_bytes == 16 * 16 * 3
    mat = desc.materials[0]
    assert mat.color_fn.kind == abi.SAMPLER_TEXTURE_VEC3 and mat.smoothness_fn.kind == abi.SAMPLER_DOUBLE_CHECKERS
    assert mat.ns_fn.kind == abi.SAMPLER_CONST


def _kd_arrays(scene, desc):
    from raytracer_amd import host
    d = desc
    out = {"root": d.kd_root, "stats": scene.kd_stats()}
    for name, cnt, dt in (("branch_min", 3 * d.n_branches, np.float64), ("branch_max", 3 * d.n_branches, np.float64),
                          ("branch_value", d.n_branches, np.float64), ("branch_axis", d.n_branches, np.int32),
                          ("branch_lesser", d.n_branches, np.int32), ("branch_greater", d.n_branches, np.int32),
                          ("leaf_first", d.n_leaves, np.uint32), ("leaf_count", d.n_leaves, np.uint32),
                          ("leaf_refs", d.n_leaf_refs, np.uint32)):
        out[name] = host.desc_array(d, name, cnt, dt).copy()
    return out


@pytest.mark.parametrize("name", ["live", "whitted", "lens", "mesh", "ties", "animated"])
def test_flat_kd_builder_equals_buildkdtree(name):
    """buildFlatKD (compact records, partial sorts, threads) and buildKDTree's node objects run
    through the flattening walk must give exactly the arrays the reference's construction gives
    (restated in oracle/kd_oracle.cpp): same branches, same leaf contents in the same order, same
    numbering, whatever the number of threads."""
    from raytracer_amd import scenes, host

    def make():
        if name == "live":
            return scenes.live_scene()[0], 0.7
        if name == "whitted":
            return scenes.whitted_scene()[0], 0.0
        if name == "lens":
            return scenes.lens_scene()[0], 0.0
        if name == "mesh":
            return scenes.mesh_scene(slices=80, stacks=40)[0], 0.0
        if name == "animated":
            return scenes.animated_scene(copies=2, slices=60, stacks=30), 1.3
        # many exactly equal keys: duplicated shapes and an axis-aligned grid of quads
        rng = np.random.default_rng(3)
        s = host.HostScene()
        m = s.material("PlainWhiteTees")
        tri = rng.integers(-4, 5, size=(300, 3, 3)).astype(np.float64)
        gx, gz = np.meshgrid(np.arange(-5.0, 5.0), np.arange(-5.0, 5.0))
        quads = np.array([[[x, 0.0, z], [x + 1, 0.0, z], [x + 1, 0.0, z + 1]] for x, z in zip(gx.ravel(), gz.ravel())])
        pos = np.concatenate([tri, tri, quads, quads[::-1]])
        nrm = np.tile([0.0, 1.0, 0.0], (len(pos), 3, 1))
        uv = np.zeros((len(pos), 3, 2))
        s.triangles(pos, nrm, uv, m)
        return s, 0.0

    from oracle import binding as orc
    got = {}
    for builder, threads in (("nodes", 0), ("flat", 1), ("flat", 3), ("flat", 8)):
        scene, t = make()
        scene.set_kd_builder(builder, threads)
        desc = scene.frame(t)
        got[(builder, threads)] = _kd_arrays(scene, desc)
    # the reference's recursion (KDTree.cpp:68-126) restated on the same shape boxes
    n = desc.n_shapes
    bmin = host.desc_array(desc, "prim_aabb_min", 3 * n, np.float64).reshape(n, 3)
    bmax = host.desc_array(desc, "prim_aabb_max", 3 * n, np.float64).reshape(n, 3)
    smin, smax = scene.scene_box()
    ref = orc.build_kd(bmin, bmax, smin, smax)
    for key, arrays in got.items():
        assert arrays["stats"]["depth"] == ref["depth"], key
        for field, want in ref.items():
            if field == "depth":
                continue
            have = arrays[field]
            same = np.array_equal(want, have) if isinstance(want, np.ndarray) else want == have
            assert same, (key, field)


def test_png_writer_round_trip(tmp_path):
    """rth_write_png output decodes (PIL) to the bytes handed in; FrameWriter names files as
    the reference's OUTPUTFRAMES loop does."""
    from PIL import Image
    from raytracer_amd import host, output
    rng = np.random.default_rng(9)
    for (h, w) in ((1, 1), (7, 13), (120, 200)):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        p = tmp_path / f"t_{h}x{w}.png"
        host.write_png(str(p), img)
        back = np.asarray(Image.open(p).convert("RGB"))
        assert back.shape == img.shape and np.array_equal(back, img)
    with pytest.raises(OSError):
        host.write_png(str(tmp_path / "no_such_dir" / "x.png"), img)
    fw = output.FrameWriter(root=str(tmp_path / "out"), stamp="2026-01-02-03-04-05")
    assert fw.path(7).endswith(os.path.join("out", "2026-01-02-03-04-05", "7.png"))


def test_oracle_present_known_values():
    """sRGB framebuffer conversion of the present step on values with known encodings."""
    from oracle import binding as orc
    lin = np.array([[[0.0, 1.0, 0.5], [0.0031308, 0.2140411, -1.0], [2.0, np.nan, 1e-9]]], dtype=np.float32)
    got = orc.present(lin, passes=1, tonemap=0, flip=False)
    assert got[0, 0].tolist() == [0, 255, 188]
    assert got[0, 1].tolist() == [10, 127, 0]
    assert got[0, 2].tolist() == [255, 0, 0]
    # cbrt tonemap, 2 passes: cbrt(0.25 / 2) = 0.5 -> 188
    got = orc.present(np.full((2, 1, 3), 0.25, dtype=np.float32), passes=2, tonemap=1)
    assert (got == 188).all()
    # vertical flip
    rows = np.zeros((2, 1, 3), dtype=np.float32); rows[1] = 1.0
    assert orc.present(rows, tonemap=0, flip=True)[0, 0, 0] == 255
    assert orc.present(rows, tonemap=0, flip=False)[0, 0, 0] == 0


def test_fbx_reader_on_fixture():
    """tests/golden/pyramid.fbx (written by tests/golden/make_fbx_fixture.py): quad base +
    4 triangles, deflated vertex array, Model node with Rx(-90), x100 scaling, translation."""
    from raytracer_amd import host
    here = os.path.dirname(os.path.abspath(__file__))
    n, names = host.fbx_probe(os.path.join(here, "golden", "pyramid.fbx"))
    assert n == 6 and names == ["stone.png"]
    # Model wants <root>/<name>/<name>.fbx: golden/pyramid.fbx is reached as root=tests, name=golden -> use a link-free copy
    import shutil, tempfile
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "pyramid"))
        shutil.copy(os.path.join(here, "golden", "pyramid.fbx"), os.path.join(root, "pyramid", "pyramid.fbx"))
        s = HostScene()
        white = s.material("PlainWhiteTees")
        first, count = s.model(root, "pyramid", material=white)
        assert (first, count) == (0, 6)
        s2 = HostScene()
        first2, _ = s2.model(root, "pyramid", pos=(1.0, 2.0, 3.0))          # own materials, texture file absent
        d2 = s2.frame(0.0)
        assert d2.n_triangles == 6 and d2.n_materials == 1
        moved = desc_array(d2, "tri_position", 54, np.float64).reshape(6, 3, 3)
    d = s.frame(0.0)
    pos = desc_array(d, "tri_position", 54, np.float64).reshape(6, 3, 3)
    nrm = desc_array(d, "tri_normal", 54, np.float64).reshape(6, 3, 3)
    uv = desc_array(d, "tri_uv", 36, np.float64).reshape(6, 3, 2)
    V = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], [0, 0, 1.5]], dtype=np.float64)
    UV = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.25]], dtype=np.float64)
    tris = [(3, 2, 1), (3, 1, 0), (0, 1, 4), (1, 2, 4), (2, 3, 4), (3, 0, 4)]      # fans of the five polygons
    for t, idx in enumerate(tris):
        for k, v in enumerate(idx):
            x, y, z = V[v]
            want = np.array([x, z + 0.5, -y])           # Rx(-90), scale 100 * 0.01, translation 50 * 0.01
            assert np.allclose(pos[t, k], want, atol=1e-6), (t, k, pos[t, k], want)
            assert np.allclose(uv[t, k], [UV[v][0], 1.0 - UV[v][1]], atol=1e-7)
            assert abs(np.linalg.norm(nrm[t, k]) - 1.0) < 1e-6
        g = np.cross(pos[t, 1] - pos[t, 0], pos[t, 2] - pos[t, 0])
        for k in range(3):
            # smooth normals stay on the side faces' side (at the base's sharp rim they lean
            # away from the base's own normal: every face at a position counts, whatever the angle)
            assert t < 2 or np.dot(g, nrm[t, k]) > 0.0
    # the apex normal is shared by the four side faces: straight up after the rotation
    assert np.allclose(nrm[2, 2], [0.0, 1.0, 0.0], atol=1e-6)
    # Model transform (Model.cpp:19-25): translation applied before the node matrix
    assert np.allclose(moved, pos + np.array([1.0, 2.0, 3.0]), atol=1e-5)


REFERENCE_MODELS = "/root/reference/Raytracer/Models"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_MODELS), reason="the reference's model files are not on this machine")
def test_fbx_reference_models_triangle_counts():
    """Triangle counts after triangulation of the reference's FBX files (data files, read in
    place), as recorded when the reference was surveyed."""
    from raytracer_amd import host
    want = {"teapot": 15704, "bunny": 30338, "backpack": 67908, "lemon": 8298, "brick": 12, "cube": 12,
            "brick2": 72, "shapes": 92}
    for name, count in want.items():
        n, names = host.fbx_probe(os.path.join(REFERENCE_MODELS, name, name + ".fbx"))
        assert n == count, (name, n)
    s = HostScene()
    first, n = s.model(REFERENCE_MODELS, "teapot", material=s.material("PlainWhiteTees"))
    s.point_light((5.0, 5.0, 5.0))
    d = s.frame(0.0)
    assert d.n_triangles == 15704
    pos = desc_array(d, "tri_position", 9 * n, np.float64).reshape(-1, 3)
    nrm = desc_array(d, "tri_normal", 9 * n, np.float64).reshape(-1, 3)
    assert np.isfinite(pos).all() and np.isfinite(nrm).all()
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1.0).max() < 1e-5
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    assert (hi - lo).max() < 1e4 and (hi - lo).min() > 1e-3
    # and it renders: rays from a camera looking at the model's centre hit it
    cam = __import__("raytracer_amd.host", fromlist=["make_camera"]).make_camera(tuple((lo + hi) / 2 + np.array([0, 0, 1.1 * (hi - lo).max()])), tuple((lo + hi) / 2))
    rd = scenes.render_desc(48, 36, camera=cam, max_depth=1)
    o, dd = camera_rays_np(rd)
    got = orc.trace_closest(d, o, dd)
    assert (got["prim"] >= 0).mean() > 0.1
