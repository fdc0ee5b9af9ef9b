#!/usr/bin/env python3
"""Soak run: the default walk of the render path (BVH candidates with the single-precision box test, KD
certificate) against the reference-order walk on the same GPU, rtx_trace_closest on 15 million rays
per seed -- random, aimed from far away, nearly axis-parallel, grazing, leaving surfaces the way bounce
and shadow rays do (where triangles behind the origin are hits, Triangle.cpp:8-72) -- over the backpack, teapot and
procedural mesh scenes.  Prints the number of rays whose hit id or depth differ (must be 0).

    python tools/soak_walks.py [seed [static]]      (static: scenes made with rtx_scene_create_static)
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import abi, device, scenes
device.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
total = bad = 0
for mesh in ("backpack", "teapot", None):
    scene, _ = scenes.mesh_scene(ggx=True, square_light=True, mesh=mesh)
    desc = scene.frame(0.0)
    with device.DeviceScene(desc, static=(len(sys.argv) > 2 and sys.argv[2] == "static")) as ds:
        surface = None
        for rep in range(5):
            n = 1_000_000
            kind = rep
            if kind == 0:       # random origins around the model, random directions
                o = rng.uniform(-6, 6, (n, 3)); d = rng.normal(size=(n, 3))
            elif kind == 1:     # origins on a far sphere aimed at the model
                t = rng.normal(size=(n, 3)); t /= np.linalg.norm(t, axis=1, keepdims=True)
                o = t * 10.0 ** rng.uniform(1, 6, (n, 1)); d = rng.uniform(-3, 3, (n, 3)) - o
            elif kind == 2:     # nearly axis-parallel
                d = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-16, -2, (n, 1))
                d[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
                o = rng.uniform(-4, 4, (n, 3))
            elif kind == 4:     # leaving the surfaces the first batch hit, 1e-3 off them, in any direction
                o = surface[rng.integers(0, len(surface), n)]; d = rng.normal(size=(n, 3))
            else:               # origins close to the surface region, grazing directions
                o = rng.uniform(-2.5, 2.5, (n, 3)); o[:, 1] = rng.uniform(0, 4, n)
                d = rng.normal(size=(n, 3)); d[:, 1] *= 0.05
            got = ds.trace_closest(o, d, flags=abi.FLAG_RENDER_WALK)
            ref = ds.trace_closest(o, d)
            if kind == 0:
                hit = ref["prim"] >= 0
                surface = ref["position"][hit] + 1e-3 * ref["normal"][hit]
            same = (got["prim"] == ref["prim"]) & ((got["depth"] == ref["depth"]) | (np.isnan(got["depth"]) & np.isnan(ref["depth"])))
            total += n; bad += int((~same).sum())
            print(json.dumps({"mesh": mesh or "procedural", "kind": kind, "rays": n, "hits": float((ref["prim"] >= 0).mean()), "differ": int((~same).sum())}), flush=True)
print(json.dumps({"total": total, "differ": bad}))
