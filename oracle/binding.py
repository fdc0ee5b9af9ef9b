"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the raytracer_amd package.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "librt_oracle.so")
_lib = None

PD = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_LIB):
        raise RuntimeError(f"{ORACLE_LIB} is missing: run `make -C oracle`")
    L = C.CDLL(ORACLE_LIB)
    L.orc_trace_closest.argtypes = [
        C.POINTER(abi.SceneDesc), PD, PD, C.c_size_t, C.c_uint32, C.POINTER(C.c_int32), PD, PD, PD, PD, PD,
        C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
    L.orc_render.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RenderDesc), C.POINTER(C.c_float), PD,
                             C.POINTER(abi.Stats), C.c_int, C.POINTER(C.c_int)]
    L.orc_camera_ray.argtypes = [C.POINTER(abi.RenderDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, PD, PD]
    L.orc_uniform.restype = C.c_double
    L.orc_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.orc_ray_init.argtypes = [PD, PD, PD, PD]
    L.orc_aabb_enter_exit.argtypes = [PD] * 6
    L.orc_aabb_hit.argtypes = [PD] * 4
    L.orc_triangle_hit.argtypes = [PD] * 10
    L.orc_sphere_hit.argtypes = [PD, C.c_double, PD, PD, PD, PD, PD, PD, PD]
    L.orc_solve_quadratic.argtypes = [C.c_double, C.c_double, C.c_double, PD, PD]
    L.orc_refraction_ray.argtypes = [PD, PD, C.c_double, C.c_int, PD, C.POINTER(C.c_int)]
    L.orc_mirror_direction.argtypes = [PD, PD, PD]
    L.orc_lighting.argtypes = [PD, PD, PD, PD, C.c_double, C.POINTER(abi.Material), C.POINTER(abi.SceneDesc), PD, PD]
    L.orc_cook_torrance.argtypes = [PD, PD, PD, PD, PD, C.POINTER(abi.Material), C.POINTER(abi.SceneDesc), PD, PD, PD]
    L.orc_light_ray.argtypes = [C.POINTER(abi.Light), PD, C.c_double, C.c_double, PD]
    L.orc_light_distance.restype = C.c_double
    L.orc_light_distance.argtypes = [C.POINTER(abi.Light), PD, PD]
    L.orc_light_attenuation.restype = C.c_double
    L.orc_light_attenuation.argtypes = [C.POINTER(abi.Light), C.c_double, PD]
    L.orc_material_color.argtypes = [C.POINTER(abi.Material), C.POINTER(abi.SceneDesc), PD, PD]
    _lib = L
    return L


def _in(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(PD)


def _out(n):
    a = np.zeros(n, dtype=np.float64)
    return a, a.ctypes.data_as(PD)


def trace_closest(desc, origin, direction, flags=0, leaf_seq_cap=0, want_counters=False):
    """Batched shootRay on the CPU.  Returns a dict of numpy arrays."""
    L = lib()
    origin, po = _in(origin)
    direction, pd = _in(direction)
    n = origin.size // 3
    prim = np.zeros(n, np.int32)
    depth, pdepth = _out(n)
    position, ppos = _out(n * 3)
    normal, pnrm = _out(n * 3)
    uv, puv = _out(n * 2)
    bary, pbary = _out(n * 3)
    leaf_count = np.zeros(n, np.uint32)
    leaf_hash = np.zeros(n, np.uint64)
    leaf_seq = np.full((n, max(leaf_seq_cap, 1)), 0xFFFFFFFF, np.uint32)
    counters = np.zeros((n, 6), np.uint32)
    r = L.orc_trace_closest(
        C.byref(desc), po, pd, n, flags, prim.ctypes.data_as(C.POINTER(C.c_int32)), pdepth, ppos, pnrm, puv,
        pbary, leaf_count.ctypes.data_as(C.POINTER(C.c_uint32)), leaf_hash.ctypes.data_as(C.POINTER(C.c_uint64)),
        leaf_seq.ctypes.data_as(C.POINTER(C.c_uint32)) if leaf_seq_cap else None, leaf_seq_cap,
        counters.ctypes.data_as(C.POINTER(C.c_uint32)) if want_counters else None)
    if r != 0:
        raise RuntimeError(f"orc_trace_closest failed: {r}")
    return dict(prim=prim, depth=depth, position=position.reshape(n, 3), normal=normal.reshape(n, 3),
                uv=uv.reshape(n, 2), bary=bary.reshape(n, 3), leaf_count=leaf_count, leaf_hash=leaf_hash,
                leaf_seq=leaf_seq, counters=counters)


def render(desc, rd, accum=None, threads=0, want_rgb64=False):
    """Per-pixel loop on the CPU.  Returns (accum float32 [h,w,3], rgb64 or None, Stats, threads used)."""
    L = lib()
    n = rd.tile_w * rd.tile_h
    if accum is None:
        accum = np.zeros((rd.tile_h, rd.tile_w, 3), np.float32)
    assert accum.dtype == np.float32 and accum.size == n * 3 and accum.flags.c_contiguous
    rgb64 = np.zeros((rd.tile_h, rd.tile_w, 3), np.float64) if want_rgb64 else None
    stats = abi.Stats()
    used = C.c_int(0)
    r = L.orc_render(C.byref(desc), C.byref(rd), accum.ctypes.data_as(C.POINTER(C.c_float)),
                     rgb64.ctypes.data_as(PD) if want_rgb64 else None, C.byref(stats), threads, C.byref(used))
    if r != 0:
        raise RuntimeError(f"orc_render failed: {r}")
    return accum, rgb64, stats, used.value


def set_reference_costs(on):
    """orc_set_reference_costs: the render / trace calls repeat the work the reference repeats
    (leaf vector copies, per-test and per-hit sphere matrix work); same results, slower."""
    lib().orc_set_reference_costs(int(bool(on)))


def camera_rays(rd, sub_sample=None):
    """Origins/directions of the camera rays of rd's tile (one per pixel; sub_sample=(sx,sy), default (1,1))."""
    L = lib()
    sx, sy = sub_sample or (1, 1)
    n = rd.tile_w * rd.tile_h
    o = np.zeros((n, 3))
    d = np.zeros((n, 3))
    oo = (C.c_double * 3)()
    dd = (C.c_double * 3)()
    k = 0
    for y in range(rd.tile_y0, rd.tile_y0 + rd.tile_h):
        for x in range(rd.tile_x0, rd.tile_x0 + rd.tile_w):
            L.orc_camera_ray(C.byref(rd), x, y, sx, sy, oo, dd)
            o[k] = oo[:]
            d[k] = dd[:]
            k += 1
    return o, d


def present(accum, passes=1, tonemap=1, flip=True):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    h, w, _ = accum.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    L = lib()
    L.orc_present.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint8)]
    L.orc_present.restype = None
    L.orc_present(accum.ctypes.data_as(C.POINTER(C.c_float)), w, h, passes, tonemap, int(flip), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def uniform(seed, frame, pixel, sample, branch, draw):
    return lib().orc_uniform(seed, frame, pixel, sample, branch, draw)


_kd_lib = None


def build_kd(box_min, box_max, scene_min, scene_max):
    """The reference's buildKDTree (oracle/kd_oracle.cpp) over shapes 0..n-1 given by their
    bounding boxes ([n,3] each; mid = (min + max) / 2.0 as AABB's two-argument constructor sets
    it, Shape.hpp:17-28), flattened in pre-order.  Returns a dict of the rtx_scene_desc KD arrays."""
    global _kd_lib
    if _kd_lib is None:
        path = os.path.join(_HERE, "libkd_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C oracle`")
        K = C.CDLL(path)
        K.orc_kd_build.restype = C.c_void_p
        K.orc_kd_build.argtypes = [C.c_uint32, PD, PD, PD, PD, PD]
        K.orc_kd_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        K.orc_kd_copy.argtypes = [C.c_void_p, PD, PD, PD] + [C.POINTER(C.c_int32)] * 3 + [C.POINTER(C.c_uint32)] * 3
        K.orc_kd_free.argtypes = [C.c_void_p]
        _kd_lib = K
    K = _kd_lib
    box_min, pmin = _in(box_min)
    box_max, pmax = _in(box_max)
    mid, pmid = _in((box_min + box_max) / 2.0)
    _, psmin = _in(scene_min)
    _, psmax = _in(scene_max)
    h = C.c_void_p(K.orc_kd_build(box_min.size // 3, pmin, pmax, pmid, psmin, psmax))
    counts = (C.c_int64 * 5)()
    K.orc_kd_counts(h, counts)
    nb, nl, nr, depth, root = list(counts)
    out = dict(branch_min=np.zeros(nb * 3), branch_max=np.zeros(nb * 3), branch_value=np.zeros(nb),
               branch_axis=np.zeros(nb, np.int32), branch_lesser=np.zeros(nb, np.int32), branch_greater=np.zeros(nb, np.int32),
               leaf_first=np.zeros(nl, np.uint32), leaf_count=np.zeros(nl, np.uint32), leaf_refs=np.zeros(nr, np.uint32))
    i32, u32 = C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    K.orc_kd_copy(h, out["branch_min"].ctypes.data_as(PD), out["branch_max"].ctypes.data_as(PD),
                  out["branch_value"].ctypes.data_as(PD), out["branch_axis"].ctypes.data_as(i32),
                  out["branch_lesser"].ctypes.data_as(i32), out["branch_greater"].ctypes.data_as(i32),
                  out["leaf_first"].ctypes.data_as(u32), out["leaf_count"].ctypes.data_as(u32),
                  out["leaf_refs"].ctypes.data_as(u32))
    K.orc_kd_free(h)
    out["root"] = root
    out["depth"] = depth
    return out
