#!/usr/bin/env python3
"""Triangle soups of the reference's FBX models as compact fixtures (container only).

Reads <reference>/Raytracer/Models/<name>/<name>.fbx with the library's own FBX reader
(raytracer_amd/host/fbx.cpp, the stand-in for the assimp import of Model.cpp:63-200), with
Model's default placement (position 0, rotation 0), and stores what the reader hands to
`new Triangle(...)` -- per-corner position, normal, uv -- as tests/golden/mesh_<name>.npz:
unique corner records (8 float64 each) plus a [triangles, 3] index array, so the soup is
rebuilt bit for bit by `records[index]`.  The fixture is data; nothing of the reference's
source is stored.  Run:  python tools/make_mesh_fixture.py teapot backpack
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODELS = "/root/reference/Raytracer/Models"


def soup_of(name):
    from raytracer_amd.host import HostScene, desc_array
    s = HostScene()
    mat = s.material("PlainWhiteTees")
    first, n = s.model(MODELS, name, material=mat)
    desc = s.frame(0.0)
    assert desc.n_triangles == n and first == 0
    pos = desc_array(desc, "tri_position", n * 9, np.float64).reshape(n, 3, 3).copy()
    nrm = desc_array(desc, "tri_normal", n * 9, np.float64).reshape(n, 3, 3).copy()
    uv = desc_array(desc, "tri_uv", n * 6, np.float64).reshape(n, 3, 2).copy()
    return pos, nrm, uv


def main():
    for name in sys.argv[1:] or ["teapot"]:
        pos, nrm, uv = soup_of(name)
        n = len(pos)
        corners = np.concatenate([pos, nrm, uv], axis=2).reshape(n * 3, 8)
        # exact de-duplication on the bit patterns
        keys = np.ascontiguousarray(corners).view(np.dtype((np.void, 64))).ravel()
        _, first_at, inverse = np.unique(keys, return_index=True, return_inverse=True)
        records = corners[first_at]
        index = inverse.reshape(n, 3).astype(np.uint32)
        assert np.array_equal(records[index].view(np.uint64), corners.reshape(n, 3, 8).view(np.uint64))
        out = os.path.join(ROOT, "tests", "golden", f"mesh_{name}.npz")
        np.savez_compressed(out, records=records, index=index)
        print(f"{name}: {n} triangles, {len(records)} unique corners -> {out} ({os.path.getsize(out) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
