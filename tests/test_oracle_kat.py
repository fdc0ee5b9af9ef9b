"""Pins the CPU oracle (and the host scene classes) to the known-answer values that were
recorded from the reference's own functions (tests/golden/reference_kat.json).  Every
comparison is bit-exact: the values were printed with %.17g, which round-trips a double."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from raytracer_amd import abi, scenes
from raytracer_amd.host import HostScene, desc_array
from oracle import binding as orc
from helpers import camera_rays_np, id_hash

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))
PD = C.POINTER(C.c_double)


def arr(v):
    a = np.ascontiguousarray(v, dtype=np.float64)
    return a, a.ctypes.data_as(PD)


def out(n):
    a = np.zeros(n)
    return a, a.ctypes.data_as(PD)


def normalized(v):
    """glm::normalize through Ray::Ray's arithmetic."""
    d, pd = out(3)
    _, po = arr([0.0, 0.0, 0.0])
    _, pv = arr(v)
    orc.lib().orc_ray_init(po, pv, pd, None)
    return d


def copper():
    s = HostScene()
    m = s.material("Copper")
    s.sphere((0, 0, 0), 1.0, m)
    desc = s.frame(0.0)
    return s, desc, desc.materials[0]


def test_ray_constructor():
    k = KAT["ray_ctor"]
    assert normalized(k["dir"]).tolist() == k["direction"]


@pytest.mark.parametrize("name", ["refraction_entering", "refraction_tir"])
def test_refraction_ray(name):
    k = KAT[name]
    d = normalized(k["d_unnormalised"])
    o, po = out(3)
    io = C.c_int(0)
    _, pn = arr(k["n"])
    orc.lib().orc_refraction_ray(pn, d.ctypes.data_as(PD), k["ior"], k["entering"], po, C.byref(io))
    assert o.tolist() == k["out"]
    assert io.value == k["internal_only"]


def test_triangle_hits():
    k = KAT["floor_triangle"]
    L = orc.lib()
    nrm = k["normal"] * 3

    def hit(order, origin, direction):
        pos = sum((k[v] for v in order), [])
        uvs = sum((k["uv_" + v] for v in order), [])
        p, pp = out(3); n, pn = out(3); uv, puv = out(2); b, pb = out(3); dep, pdep = out(1)
        ok = L.orc_triangle_hit(arr(pos)[1], arr(nrm)[1], arr(uvs)[1], arr(origin)[1], arr(direction)[1],
                                pp, pn, puv, pb, pdep)
        return ok, p, n, uv, b, dep[0]

    ok, p, n, uv, b, dep = hit("acb", k["ray_origin"], k["ray_dir"])
    want = k["acb"]
    assert ok == 1
    assert p.tolist() == want["position"]
    assert b.tolist() == want["bary"]
    assert dep == want["depth"]
    assert uv.tolist() == want["uv"]           # evaluated in FP32 (mat3x2 is a float matrix)
    ok, *_ = hit("adc", k["ray_origin"], k["ray_dir"])
    assert ok == 0
    # no t >= 0 test: a front-facing triangle behind the origin is reported
    bo = k["behind_origin"]
    ok, p, n, uv, b, dep = hit("acb", bo["ray_origin"], bo["ray_dir"])
    assert ok == bo["hit"] and dep == bo["depth"] and p[1] == bo["position_y"]


def test_sphere_hit():
    k = KAT["sphere"]
    ident = np.eye(4).reshape(-1)
    p, pp = out(3); n, pn = out(3); uv, puv = out(2); dep, pdep = out(1)
    ok = orc.lib().orc_sphere_hit(arr(k["center"])[1], k["radius"] ** 2, arr(ident)[1], arr(k["ray_origin"])[1],
                                  arr(k["ray_dir"])[1], pp, pn, puv, pdep)
    assert ok == 1
    assert dep[0] == k["depth"]
    assert n.tolist() == k["normal"]
    # uv.x goes through atan2: libm of this host must agree to the last bit with the recorded one
    assert uv.tolist() == pytest.approx(k["uv"], abs=2e-16)


def test_sphere_redo_aabb_through_host_classes():
    k = KAT["sphere_redo_aabb"]
    s = HostScene()
    s.sphere((0.0, 0.0, 0.0), k["radius"], s.material("PlainWhiteTees"), motion=k["motion"])
    desc = s.frame(k["t"])
    assert desc_array(desc, "prim_aabb_min", 3, np.float64).tolist() == k["box_min"]


@pytest.mark.parametrize("name", ["aabb_enter_exit", "aabb_axis_parallel"])
def test_aabb_enter_exit(name):
    k = KAT[name]
    e, pe = out(3); x, px = out(3)
    hit = orc.lib().orc_aabb_enter_exit(arr(k["bmin"])[1], arr(k["bmax"])[1], arr(k["ray_origin"])[1],
                                        arr(k["ray_dir"])[1], pe, px)
    assert hit == k["hit"]
    assert e.tolist() == k["enter"] and x.tolist() == k["exit"]
    assert orc.lib().orc_aabb_hit(arr(k["bmin"])[1], arr(k["bmax"])[1], arr(k["ray_origin"])[1],
                                  arr(k["ray_dir"])[1]) == 1


def test_cook_torrance():
    k = KAT["cook_torrance"]
    s, desc, m = copper()
    color = np.array(m.color[:])
    f0 = np.zeros(3) * (1.0 - m.metalness) + color * m.metalness
    o, po = out(3); ks, pks = out(3)
    uv = np.zeros(2)
    orc.lib().orc_cook_torrance(arr(k["in_dir"])[1], arr(k["out_dir"])[1], arr(k["normal"])[1], arr(uv)[1],
                                arr(k["down"])[1], C.byref(m), C.byref(desc), arr(f0)[1], po, pks)
    assert o.tolist() == k["out"]
    assert ks.tolist() == k["ks"]


def test_lighting_function():
    k = KAT["lighting_function"]
    s, desc, m = copper()
    o, po = out(3)
    orc.lib().orc_lighting(arr(k["in_dir"])[1], arr(k["light_dir"])[1], arr(k["normal"])[1], arr(np.zeros(2))[1],
                           k["attenuation"], C.byref(m), C.byref(desc), arr(k["light_color"])[1], po)
    assert o.tolist() == k["out"]


def test_point_and_directional_lights_through_host_classes():
    L = orc.lib()
    k = KAT["point_light"]
    s = HostScene()
    s.sphere((0, 0, 0), 1.0, s.material("PlainWhiteTees"))
    s.point_light(k["position"])
    s.directional_light(KAT["directional_light"]["direction"])
    desc = s.frame(0.0)
    pl, dl = desc.lights[0], desc.lights[1]
    d, pd = out(3)
    L.orc_light_ray(C.byref(pl), arr(k["point"])[1], 0.0, 0.0, pd)
    assert d.tolist() == k["dir"]
    dist = L.orc_light_distance(C.byref(pl), arr(k["point"])[1], pd)
    assert dist == k["distance"]
    assert L.orc_light_attenuation(C.byref(pl), dist, arr(k["attenuation_clq"])[1]) == k["attenuation"]
    k = KAT["directional_light"]
    L.orc_light_ray(C.byref(dl), arr(k["point"])[1], 0.0, 0.0, pd)
    assert d.tolist() == k["dir"]        # aims at a finite far point: 1e-13 skew
    assert L.orc_light_distance(C.byref(dl), arr(k["point"])[1], pd) == 1e13
    assert L.orc_light_attenuation(C.byref(dl), 5.0, arr([1.0, 0.09, 0.032])[1]) == 1.0


@pytest.mark.parametrize("t,key", [(0.0, "t0"), (1.7, "t1_7")])
def test_live_scene_primary_hit_hash(t, key):
    """Host KD builder + flattener + oracle traversal + camera set-up, end to end, against the
    hash of primary-hit ids recorded from the reference's code on its own scene."""
    k = KAT["live_scene_hit_hash"]
    scene, cam = scenes.live_scene()
    desc = scene.frame(t)
    st = scene.kd_stats()
    assert {n: st[n] for n in ("branches", "leaves", "leaf_refs", "depth")} == k["kd"]
    rd = scenes.render_desc(k["width"], k["height"], camera=cam)
    o, d = camera_rays_np(rd, sub_sample=(0, 0))
    r = orc.trace_closest(desc, o, d)
    assert (r["prim"] >= 0).all()
    assert "%016x" % id_hash(r["prim"]) == k[key]
    lin = orc.trace_closest(desc, o, d, flags=abi.FLAG_LINEAR_SCAN)
    assert np.array_equal(lin["prim"], r["prim"])       # KD walk == linear scan on this scene


def test_camera_rays_numpy_equals_scalar_code():
    _, cam = scenes.live_scene()
    for n, sub in [(1, (0, 0)), (1, (1, 1)), (4, (3, 2))]:
        rd = scenes.render_desc(48, 36, camera=cam, n=n, tile=(5, 7, 30, 20))
        o1, d1 = orc.camera_rays(rd, sub)
        o2, d2 = camera_rays_np(rd, sub)
        assert np.array_equal(o1, o2) and np.array_equal(d1, d2)


def test_mirror_direction_uses_sin_pi():
    """-rotate(d, pi, n): sin(pi) = 1.22e-16 leaks into the result (Raytracer.cpp:284)."""
    assert math.sin(math.pi) == 1.2246467991473532e-16 and math.cos(math.pi) == -1.0
    d = normalized([0.3, -1.0, 0.2])
    o, po = out(3)
    orc.lib().orc_mirror_direction(d.ctypes.data_as(PD), arr([0.0, 1.0, 0.0])[1], po)
    ideal = d - 2.0 * d[1] * np.array([0.0, 1.0, 0.0])
    assert np.allclose(o, ideal, atol=1e-15)
    assert not np.array_equal(o, ideal)


def test_entering_is_a_threshold_on_the_dot_product():
    """`entering = acos(dot(n, -d)) < pi / 2.0` (Raytracer.cpp:248-249): pi / 2.0 is the double just
    below the real pi/2, so the C library's acos compares below it exactly for
    dot >= 1.7225464241988333e-16 (and <= 1: acos is NaN beyond).  The HIP code tests the dot
    product against that constant instead of calling acos; here the oracle, which calls acos, is
    swept over the bracket of the threshold and over the whole range."""
    L = orc.lib()
    L.orc_entering_for_dot.argtypes = [C.c_double]
    T = np.array([0x3ca8d313198a2e04], dtype=np.uint64).view(np.float64)[0]
    assert T == 1.7225464241988333e-16
    text = open(os.path.join(os.path.dirname(__file__), "..", "raytracer_amd", "csrc", "rtx_device.hpp")).read()
    assert "kEnteringMin = 1.7225464241988333e-16" in text

    def expected(x):
        return int(x >= T and x <= 1.0)

    bits = np.uint64(0x3ca8d313198a2e04)
    for k in range(-5000, 5001):
        x = np.array([bits + np.uint64(k) if k >= 0 else bits - np.uint64(-k)], dtype=np.uint64).view(np.float64)[0]
        assert L.orc_entering_for_dot(float(x)) == expected(x), (k, x)
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-1.5, 1.5, 20000), 10.0 ** rng.uniform(-300, 0.2, 20000), -10.0 ** rng.uniform(-300, 0.2, 2000),
                         [0.0, -0.0, 1.0, np.nextafter(1.0, 2.0), np.nextafter(1.0, 0.0), -1.0, np.nan, np.inf, -np.inf,
                          5e-324, 2.2250738585072014e-308]])
    for x in xs:
        assert L.orc_entering_for_dot(float(x)) == expected(x), x


def test_fresnel_fifth_power_by_multiplication_stays_close_to_pow():
    """The device evaluates pow(1 - dot(vh, vi), 5.0) of fresnelSchlick (CookTorrance.cpp:39-42) as
    (b*b)*(b*b)*b; the oracle calls pow as the reference does.  Three roundings against the C
    library's one: the product's relative distance from pow stays below 3 * 2^-53 + pow's own
    half unit (5.6e-16), i.e. within three units in the last place, over [0, 1] (sampled densely near 0 and
    1 as well) -- eleven orders below the 1e-4 image tolerance however many bounces multiply it."""
    rng = np.random.default_rng(5)
    b = np.concatenate([rng.random(400_000), rng.random(100_000) ** 8, 1.0 - rng.random(100_000) ** 8,
                        np.array([0.0, 1.0, 0.5, 2.0 ** -20, 1.0 - 2.0 ** -53])])
    b2 = b * b
    product = (b2 * b2) * b
    exact = np.power(b, 5.0)
    normal = exact > 1e-300
    assert np.all(np.abs(product - exact)[normal] <= 3.0 * np.spacing(exact)[normal])
    assert np.all(np.abs(product - exact)[normal] <= 5.6e-16 * exact[normal])
    assert np.all(product[~normal] <= 1e-300)


def test_specular_power_by_squaring_stays_close_to_pow():
    """The device evaluates pow(x, ns) of lightingFunction (Raytracer.cpp:191) by square-and-multiply when
    ns is a whole number up to 1 024 (the presets' 2, 50, 100); the oracle calls pow as the reference
    does.  The product's relative distance from pow stays below 0.45 * ns units in the last place plus
    pow's own half unit, over [0, 1], sampled densely near both ends: 9e-15 at ns = 100."""
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.random(200_000), 1.0 - 10.0 ** rng.uniform(-14, -1, 50_000), 10.0 ** rng.uniform(-4, 0, 50_000),
                        np.array([0.0, 1.0, 0.5, 1.0 - 2.0 ** -53])])

    def by_squaring(b, e):
        r = np.ones_like(b)
        b = b.copy()
        while e:
            if e & 1:
                r = r * b
            e >>= 1
            if e:
                b = b * b
        return r

    for ns in (0, 1, 2, 3, 50, 100, 128, 1000, 1024):
        product = by_squaring(x, ns)
        exact = np.power(x, float(ns))
        normal = exact > 1e-290
        bound = (0.45 * ns + 1.0) * 2.0 ** -52
        assert np.all(np.abs(product - exact)[normal] <= bound * exact[normal]), ns
        assert np.all(product[~normal] <= 1e-289), ns
    assert by_squaring(np.array([0.0]), 0)[0] == 1.0 and by_squaring(np.array([0.0]), 7)[0] == 0.0
