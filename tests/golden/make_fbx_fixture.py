#!/usr/bin/env python3
"""Writes tests/golden/pyramid.fbx: a small binary FBX 7400 file made here (not taken from
anywhere) for the FBX reader tests.  One mesh -- a square pyramid whose base is a quad, so the
reader has to triangulate --, one Model node with a local rotation and the x100 scaling that
Blender exports carry, one material with a diffuse texture name, UVs by polygon vertex with an
index array; the Vertices array is stored deflated, the others raw.

    python tests/golden/make_fbx_fixture.py
"""
import os
import struct
import zlib

VERTS = [-1.0, -1.0, 0.0, 1.0, -1.0, 0.0, 1.0, 1.0, 0.0, -1.0, 1.0, 0.0, 0.0, 0.0, 1.5]
POLYS = [[3, 2, 1, 0], [0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]]
UV = [0.0, 0.0, 1.0, 0.0, 1.0, 1.0, 0.0, 1.0, 0.5, 0.25]
ROTATION = (-90.0, 0.0, 0.0)
SCALING = (100.0, 100.0, 100.0)
TRANSLATION = (0.0, 50.0, 0.0)


def prop(v):
    if isinstance(v, bool):
        return b"C" + struct.pack("<b", v)
    if isinstance(v, int):
        return (b"I" + struct.pack("<i", v)) if -2**31 <= v < 2**31 and not getattr(v, "long", False) else b"L" + struct.pack("<q", v)
    if isinstance(v, float):
        return b"D" + struct.pack("<d", v)
    if isinstance(v, (bytes, str)):
        b = v.encode() if isinstance(v, str) else v
        return b"S" + struct.pack("<I", len(b)) + b
    kind, values, deflate = v
    fmt = {"d": "<%dd", "i": "<%di"}[kind] % len(values)
    raw = struct.pack(fmt, *values)
    data = zlib.compress(raw) if deflate else raw
    return kind.encode() + struct.pack("<III", len(values), 1 if deflate else 0, len(data)) + data


class Long(int):
    long = True


def node(name, props=(), children=(), at=0):
    """Serialises a node that starts at absolute offset `at`."""
    pbytes = b"".join(prop(p) for p in props)
    head = 12 + 1 + len(name)
    body = b""
    pos = at + head + len(pbytes)
    for c in children:
        cb = c(pos)
        body += cb
        pos += len(cb)
    if children:
        body += b"\0" * 13
        pos += 13
    return struct.pack("<IIIB", pos, len(props), len(pbytes), len(name)) + name.encode() + pbytes + body


def N(name, props=(), children=()):
    return lambda at: node(name, props, children, at)


def P(*values):
    return N("P", values)


def main():
    index = []
    for p in POLYS:
        index += p[:-1] + [~p[-1]]
    uv_index = []
    for p in POLYS:
        uv_index += p
    geometry = N("Geometry", (Long(1001), b"Pyramid\x00\x01Geometry", b"Mesh"), (
        N("Vertices", (("d", VERTS, True),)),
        N("PolygonVertexIndex", (("i", index, False),)),
        N("LayerElementNormal", (0,), (N("Normals", (("d", [0.0] * (3 * len(index)), False),)),)),
        N("LayerElementUV", (0,), (
            N("MappingInformationType", (b"ByPolygonVertex",)),
            N("ReferenceInformationType", (b"IndexToDirect",)),
            N("UV", (("d", UV, False),)),
            N("UVIndex", (("i", uv_index, False),)))),
        N("LayerElementMaterial", (0,), (
            N("MappingInformationType", (b"AllSame",)),
            N("ReferenceInformationType", (b"IndexToDirect",)),
            N("Materials", (("i", [0], False),)))),
    ))
    model = N("Model", (Long(2002), b"Pyramid\x00\x01Model", b"Mesh"), (
        N("Properties70", (), (
            P(b"Lcl Translation", b"Lcl Translation", b"", b"A", *TRANSLATION),
            P(b"Lcl Rotation", b"Lcl Rotation", b"", b"A", *ROTATION),
            P(b"Lcl Scaling", b"Lcl Scaling", b"", b"A", *SCALING))),
    ))
    material = N("Material", (Long(3003), b"Stone\x00\x01Material", b""), ())
    texture = N("Texture", (Long(4004), b"Albedo\x00\x01Texture", b""), (
        N("RelativeFilename", (b"textures\\stone.png",)),))
    objects = N("Objects", (), (geometry, model, material, texture))
    connections = N("Connections", (), (
        N("C", (b"OO", Long(2002), Long(0))),
        N("C", (b"OO", Long(1001), Long(2002))),
        N("C", (b"OO", Long(3003), Long(2002))),
        N("C", (b"OP", Long(4004), Long(3003), b"DiffuseColor")),
    ))
    out = b"Kaydara FBX Binary  \x00\x1a\x00" + struct.pack("<I", 7400)
    for top in (N("FBXHeaderExtension", (), (N("Creator", (b"raytracer_amd test fixture",)),)), objects, connections):
        out += top(len(out))
    out += b"\0" * 13
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyramid.fbx")
    with open(path, "wb") as f:
        f.write(out)
    print(path, len(out), "bytes")


if __name__ == "__main__":
    main()
