"""Shared helpers for the test-suite (test infrastructure; may use the oracle)."""
import numpy as np


def camera_rays_np(rd, sub_sample=(1, 1)):
    """Vectorised restatement of the per-pixel ray set-up (Raytracer.cpp:676-700).

    numpy evaluates the same IEEE operations in the same order as the scalar
    code, so the rays are bit-identical to orc_camera_ray's (checked in
    tests/test_oracle_kat.py)."""
    sx, sy = sub_sample
    c = rd.camera
    xs = np.arange(rd.tile_x0, rd.tile_x0 + rd.tile_w, dtype=np.float64)
    ys = np.arange(rd.tile_y0, rd.tile_y0 + rd.tile_h, dtype=np.float64)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    nx = (X / float(rd.width)) - 0.5
    ny = (Y / float(rd.height)) - 0.5
    eye = np.array(c.eye[:])
    right, up, fwd = np.array(c.right[:]), np.array(c.up[:]), np.array(c.forward[:])
    on = ((right * nx[..., None]) + (up * ny[..., None])) + eye
    on = on + (fwd * c.focal)
    cx = 1.0 / (float(rd.width) - 1.0)
    cy = 1.0 / (float(rd.height) - 1.0)
    n = rd.pixel_multisample_n
    off = np.array([float(sx) * (cx / float(n + 1)), float(sy) * (cy / float(n + 1)), 0.0])
    v = (on + off) - eye
    tx, ty, tz = v[..., 0] * v[..., 0], v[..., 1] * v[..., 1], v[..., 2] * v[..., 2]
    inv = 1.0 / np.sqrt((tx + ty) + tz)
    d = v * inv[..., None]
    o = np.broadcast_to(eye, d.shape)
    return np.ascontiguousarray(o.reshape(-1, 3)), np.ascontiguousarray(d.reshape(-1, 3))


def id_hash(prims):
    """FNV-style hash of a primitive-id image (id+1, 0 on a miss), row-major."""
    h = 1469598103934665603
    for p in prims:
        h = ((h ^ ((int(p) + 1) & 0xFFFFFFFFFFFFFFFF)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def psnr(a, b, peak=None):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return np.inf
    if peak is None:
        peak = max(np.abs(b).max(), 1e-12)
    return 10.0 * np.log10(peak * peak / mse)


def rel_err(a, b, floor=1e-3):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def compare_trace(got, want, seq_cap=0, desc=None):
    """Comparison of two trace_closest results: everything bit-exact, except that the
    uv of a SPHERE hit goes through atan2 (Sphere.cpp:75), whose last bits differ
    between the host's and the device's libm; those may differ by a few ulp."""
    for key in ("prim", "depth", "leaf_count", "leaf_hash", "position", "normal", "bary", "counters"):
        g, w = got[key], want[key]
        if g.dtype.kind == "f":
            same = (g == w) | (np.isnan(g) & np.isnan(w))
        else:
            same = g == w
        assert same.all(), f"{key}: {np.count_nonzero(~same)} of {same.size} entries differ"
    g, w = got["uv"], want["uv"]
    same = (g == w) | (np.isnan(g) & np.isnan(w))
    if not same.all():
        assert desc is not None, "uv differs and no scene description was given to tell spheres apart"
        prim = want["prim"]
        ptype = np.ctypeslib.as_array(desc.prim_type, shape=(desc.n_prims,))
        is_tri = np.zeros(prim.shape, bool)
        hit = prim >= 0
        is_tri[hit] = ptype[prim[hit]] == 0
        assert same[is_tri].all(), "uv of triangle hits must be bit-exact"
        assert np.abs(g - w)[~same].max() < 1e-14, "sphere uv differs by more than libm rounding"
    if seq_cap:
        assert np.array_equal(got["leaf_seq"], want["leaf_seq"]), "leaf sequences differ"
