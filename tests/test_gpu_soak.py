"""The default walk of the render path (candidates from the library's BVH with the single-precision
box test, certified against the reference's KD tree) against the reference-order walk
(traverseKDTree, KDTree.cpp:158-220) on the same GPU: rtx_trace_closest on 1.5 million rays per
model -- random, aimed from far away, nearly axis-parallel, grazing, and leaving surfaces the way
bounce and shadow rays do (where triangles behind the origin are hits, Triangle.cpp:8-72) -- over
the scenes the bench times (backpack model) and the teapot and procedural meshes.  Hit ids and
depths must be equal bit for bit.  tools/soak_walks.py runs the same at ten times the size."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


def soak_batches(rng, n):
    """Five batches of n rays; the fifth needs `surface` points (set from the first batch's hits)."""
    surface = [None]

    def make(kind):
        if kind == 0:       # random origins around the model, random directions
            return rng.uniform(-6, 6, (n, 3)), rng.normal(size=(n, 3))
        if kind == 1:       # origins on a far sphere aimed at the model
            t = rng.normal(size=(n, 3))
            t /= np.linalg.norm(t, axis=1, keepdims=True)
            o = t * 10.0 ** rng.uniform(1, 6, (n, 1))
            return o, rng.uniform(-3, 3, (n, 3)) - o
        if kind == 2:       # nearly axis-parallel
            d = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-16, -2, (n, 1))
            d[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
            return rng.uniform(-4, 4, (n, 3)), d
        if kind == 3:       # origins close to the surface region, grazing directions
            o = rng.uniform(-2.5, 2.5, (n, 3))
            o[:, 1] = rng.uniform(0, 4, n)
            d = rng.normal(size=(n, 3))
            d[:, 1] *= 0.05
            return o, d
        # leaving the surfaces the first batch hit, 1e-3 off them, in any direction
        return surface[0][rng.integers(0, len(surface[0]), n)], rng.normal(size=(n, 3))
    return make, surface


@pytest.mark.parametrize("static", [True, False], ids=["static", "per-frame"])
@pytest.mark.parametrize("mesh", ["backpack", "teapot", None], ids=["backpack", "teapot", "procedural"])
def test_render_walk_equals_reference_order_walk(gpu, mesh, static):
    scene = scenes.frame16_scene("path", mesh=mesh, texture_size=256)
    desc = scene.frame(0.0)
    rng = np.random.default_rng(11 + (1 if static else 0))
    n = 300_000
    make, surface = soak_batches(rng, n)
    hits = 0
    with gpu.DeviceScene(desc, static=static) as ds:
        for kind in range(5):
            o, d = make(kind)
            got = ds.trace_closest(o, d, flags=abi.FLAG_RENDER_WALK)
            ref = ds.trace_closest(o, d)
            if kind == 0:
                hit = ref["prim"] >= 0
                surface[0] = ref["position"][hit] + 1e-3 * ref["normal"][hit]
                assert len(surface[0]) > 1000
            same = (got["prim"] == ref["prim"]) & ((got["depth"] == ref["depth"]) | (np.isnan(got["depth"]) & np.isnan(ref["depth"])))
            assert same.all(), f"kind {kind}: {np.count_nonzero(~same)} of {n} rays differ"
            hits += int((ref["prim"] >= 0).sum())
    assert hits > n // 2
