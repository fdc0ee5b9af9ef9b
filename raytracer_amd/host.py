"""Python face of the host scene API (raytracer_amd/host/*.cpp).

A HostScene is assembled the way the reference's main() assembles its scene
(Raytracer/Raytracer.cpp:448-557): materials, shapes, lights; then `frame(t)`
does what the head of the reference's frame loop does (redoAABBs + buildKDTree,
Raytracer.cpp:634-636) and returns the flattened rtx_scene_desc that both the
HIP library and the CPU oracle consume.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(_HERE, "host", "librth_host.so")

_lib = None


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOST_LIB):
        raise RuntimeError(f"{HOST_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(HOST_LIB)
    PD = C.POINTER(C.c_double)
    L.rth_scene_new.restype = C.c_void_p
    L.rth_scene_free.argtypes = [C.c_void_p]
    L.rth_scene_error.restype = C.c_char_p
    L.rth_scene_error.argtypes = [C.c_void_p]
    L.rth_texture_add.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int]
    L.rth_material_preset.argtypes = [C.c_void_p, C.c_char_p, PD, C.c_uint32, C.c_int]
    L.rth_material_custom.argtypes = [C.c_void_p, PD, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, PD, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.rth_material_set_color.argtypes = [C.c_void_p, C.c_int, PD]
    L.rth_add_triangles.argtypes = [C.c_void_p, PD, PD, PD, C.c_size_t, C.c_int]
    L.rth_add_sphere.argtypes = [C.c_void_p, PD, C.c_double, C.c_int, C.c_int]
    L.rth_add_biconvex.argtypes = [C.c_void_p, PD, PD, C.c_double, C.c_double]
    L.rth_add_point_light.argtypes = [C.c_void_p, PD, PD]
    L.rth_add_directional_light.argtypes = [C.c_void_p, PD, PD]
    L.rth_add_square_light.argtypes = [C.c_void_p, PD, PD, C.c_double, C.c_double, PD]
    L.rth_frame.restype = C.POINTER(abi.SceneDesc)
    L.rth_frame.argtypes = [C.c_void_p, C.c_double]
    L.rth_kd_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.rth_scene_box.argtypes = [C.c_void_p, PD]
    L.rth_frame_timings.argtypes = [C.c_void_p, PD]
    L.rth_set_kd_builder.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.rth_make_camera.argtypes = [PD, PD, C.c_double, C.POINTER(abi.Camera)]
    L.rth_add_model.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, PD, PD, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.rth_fbx_probe.restype = C.c_long
    L.rth_fbx_probe.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.rth_write_png.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]
    _lib = L
    return L


def make_camera(eye, lookat, fov_degrees=90.0):
    """Camera basis of Raytracer.cpp:642-660."""
    cam = abi.Camera()
    _, pe = _d(eye)
    _, pl = _d(lookat)
    lib().rth_make_camera(pe, pl, float(fov_degrees), C.byref(cam))
    return cam


class HostScene:
    def __init__(self):
        self._L = lib()
        self._h = C.c_void_p(self._L.rth_scene_new())
        self.desc = None

    def close(self):
        if self._h:
            self._L.rth_scene_free(self._h)
            self._h = None
            self.desc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, r):
        if r < 0:
            raise RuntimeError(self._L.rth_scene_error(self._h).decode())
        return r

    # ---- materials / textures
    def texture(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        return self._check(self._L.rth_texture_add(self._h, rgb.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                   rgb.shape[1], rgb.shape[0]))

    def material(self, premade, emission=(0.0, 0.0, 0.0), color_sampler=abi.SAMPLER_CONST, texture=0, color=None):
        """Material(premade, emission, colorFn) [+ ->setColor(color)]."""
        _, pe = _d(emission)
        m = self._check(self._L.rth_material_preset(self._h, premade.encode(), pe, color_sampler, texture))
        if color is not None:
            _, pc = _d(color)
            self._check(self._L.rth_material_set_color(self._h, m, pc))
        return m

    def material_custom(self, color, ns=100.0, ni=1.0, transparency=0.0, metalness=0.5, smoothness=0.5,
                        emission=(0.0, 0.0, 0.0), samplers=None):
        """Material(color, ns, ni, transparency, metalness, smoothness, emission, fns...).

        samplers: optional dict channel -> (kind, texture) for the channels
        color, ns, ni, transparency, metalness, smoothness.
        """
        order = ["color", "ns", "ni", "transparency", "metalness", "smoothness"]
        kinds = (C.c_uint32 * 6)()
        texs = (C.c_int * 6)()
        for i, name in enumerate(order):
            k, t = (samplers or {}).get(name, (abi.SAMPLER_CONST, 0))
            kinds[i], texs[i] = k, t
        _, pc = _d(color)
        _, pe = _d(emission)
        return self._check(self._L.rth_material_custom(self._h, pc, ns, ni, transparency, metalness,
                                                       smoothness, pe, kinds, texs))

    # ---- shapes
    def triangles(self, position, normal, uv, material):
        position, pp = _d(position)
        n = position.size // 9
        normal, pn = _d(normal)
        uv, pu = _d(uv)
        assert normal.size == n * 9 and uv.size == n * 6
        return self._check(self._L.rth_add_triangles(self._h, pp, pn, pu, n, material))

    def model(self, models_root, modelname, pos=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0), material=None,
              load_textures=True):
        """addModel(shapes, modelname, pos, rot) (Raytracer.cpp:362-366): the triangles of
        <models_root>/<modelname>/<modelname>.fbx.  With `material` every triangle gets it;
        otherwise the model's own materials are used, their diffuse textures decoded here
        (PIL) when load_textures is set and the files exist.  Returns (first shape, count)."""
        path = os.path.join(models_root, modelname, modelname + ".fbx")
        n, names = fbx_probe(path)
        ids = []
        if material is None and load_textures:
            for name in names:
                tex = -1
                f = os.path.join(models_root, modelname, name) if name else None
                if f and os.path.exists(f):
                    from PIL import Image
                    tex = self.texture(np.asarray(Image.open(f).convert("RGB")))
                ids.append(tex)
        arr = (C.c_int * max(len(ids), 1))(*ids) if ids else None
        _, pp = _d(pos)
        _, pr = _d(rotation)
        first = self._check(self._L.rth_add_model(self._h, os.fsencode(models_root), modelname.encode(), pp, pr,
                                                  -1 if material is None else material, arr, len(ids)))
        return first, n

    def sphere(self, origin, radius, material, motion="noMovement"):
        _, po = _d(origin)
        return self._check(self._L.rth_add_sphere(self._h, po, float(radius), material, abi.MOTION[motion]))

    def biconvex(self, origin, forward, lens_radius, focal_length):
        _, po = _d(origin)
        _, pf = _d(forward)
        return self._check(self._L.rth_add_biconvex(self._h, po, pf, float(lens_radius), float(focal_length)))

    # ---- lights
    def point_light(self, position, color=(1.0, 1.0, 1.0)):
        _, pp = _d(position)
        _, pc = _d(color)
        return self._check(self._L.rth_add_point_light(self._h, pp, pc))

    def directional_light(self, direction, color=(1.0, 1.0, 1.0)):
        _, pd = _d(direction)
        _, pc = _d(color)
        return self._check(self._L.rth_add_directional_light(self._h, pd, pc))

    def square_light(self, position, normal, width, height, color=(1.0, 1.0, 1.0)):
        _, pp = _d(position)
        _, pn = _d(normal)
        _, pc = _d(color)
        return self._check(self._L.rth_add_square_light(self._h, pp, pn, float(width), float(height), pc))

    # ---- per frame
    def frame(self, current_time=0.0):
        """redoAABBs + buildKDTree + flatten; returns the rtx_scene_desc (valid until the next call
        or close()).  The description points into arrays this scene owns, so it keeps a reference
        to the scene: holding the description alone is enough to keep them alive."""
        p = self._L.rth_frame(self._h, float(current_time))
        if not p:
            raise RuntimeError(self._L.rth_scene_error(self._h).decode())
        self.desc = p.contents
        self.desc._owner = self
        return self.desc

    def set_kd_builder(self, builder="flat", threads=0):
        """'flat': the tree is built straight into the flattened arrays, on `threads` threads
        (0 = automatic); 'nodes': through buildKDTree's KDNode objects and a flattening walk, the
        way a driver written against the reference's classes gets it.  Both give the same arrays."""
        self._L.rth_set_kd_builder(self._h, {"flat": 0, "nodes": 1}[builder], int(threads))

    def kd_stats(self):
        out = (C.c_uint32 * 5)()
        self._L.rth_kd_stats(self._h, out)
        return dict(zip(["branches", "leaves", "leaf_refs", "depth", "max_leaf"], list(out)))

    def frame_timings(self):
        """Seconds the last frame() spent in redoAABBs, buildKDTree, flatten, freeing the old tree."""
        out = (C.c_double * 4)()
        self._L.rth_frame_timings(self._h, out)
        return dict(zip(["redo_aabbs", "build_kd_tree", "flatten", "free_old_tree"], list(out)))

    def scene_box(self):
        out = (C.c_double * 6)()
        self._L.rth_scene_box(self._h, out)
        return np.array(out[:3]), np.array(out[3:])


def fbx_probe(path):
    """(triangles after triangulation, diffuse texture file name per material slot) of a binary FBX file."""
    buf = C.create_string_buffer(1 << 16)
    n = lib().rth_fbx_probe(os.fsencode(path), buf, len(buf))
    if n < 0:
        raise RuntimeError(buf.value.decode(errors="replace"))
    text = buf.value.decode(errors="replace")
    return n, text.split("\n")[:-1] if text else []


def write_png(path, rgb8):
    """[H, W, 3] uint8, top row first -> PNG file (the stand-in for stbi_write_png, Raytracer.cpp:104)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, c = rgb8.shape
    assert c == 3
    if lib().rth_write_png(os.fsencode(path), rgb8.ctypes.data_as(C.POINTER(C.c_uint8)), w, h) != 0:
        raise OSError(f"cannot write {path}")


def desc_array(desc, name, count, dtype):
    """View one of the flattened arrays of an rtx_scene_desc as numpy (no copy)."""
    ptr = getattr(desc, name)
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(count,)).view(dtype)
