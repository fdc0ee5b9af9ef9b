"""Python face of the C ABI in include/rtx.h (raytracer_amd/csrc/librtx_hip.so).

This module only marshals arguments; every computation happens in the HIP
library.  If the library is missing or no MI355X is usable the calls raise --
there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTX_HIP_LIB names another build of the library (kernel experiments)
HIP_LIB = os.environ.get("RTX_HIP_LIB") or os.path.join(_HERE, "csrc", "librtx_hip.so")
_lib = None

PD = C.POINTER(C.c_double)

# every symbol include/rtx.h declares
EXPORTS = ["rtx_init", "rtx_device_count", "rtx_scene_create", "rtx_scene_create_static", "rtx_scene_set_stream", "rtx_scene_destroy",
           "rtx_scene_set_async", "rtx_scene_wait", "rtx_scene_clone",
           "rtx_trace_closest", "rtx_render", "rtx_render_device", "rtx_render_device_noalloc", "rtx_present", "rtx_present_device", "rtx_uniform",
           "rtx_last_error"]


class RtxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtx error {code}: {message}")
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HIP_LIB):
        raise RuntimeError(f"{HIP_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(HIP_LIB)
    L.rtx_init.argtypes = [C.c_int]
    L.rtx_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_void_p)]
    L.rtx_scene_create_static.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_void_p)]
    L.rtx_scene_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.rtx_scene_destroy.argtypes = [C.c_void_p]
    L.rtx_scene_set_async.argtypes = [C.c_void_p, C.c_int]
    L.rtx_scene_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.rtx_scene_wait.argtypes = [C.c_void_p, C.POINTER(abi.Stats)]
    L.rtx_scene_destroy.restype = None
    L.rtx_trace_closest.argtypes = [
        C.c_void_p, PD, PD, C.c_size_t, C.c_uint32, C.POINTER(C.c_int32), PD, PD, PD, PD, PD,
        C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
    L.rtx_render.argtypes = [C.c_void_p, C.POINTER(abi.RenderDesc), C.POINTER(C.c_float), C.POINTER(abi.Stats)]
    L.rtx_render_device.argtypes = [C.c_void_p, C.POINTER(abi.RenderDesc), C.c_void_p, C.POINTER(abi.Stats)]
    L.rtx_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(abi.RenderDesc), C.c_void_p]
    L.rtx_present.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                              C.POINTER(C.c_uint8)]
    L.rtx_present_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    L.rtx_uniform.restype = C.c_double
    L.rtx_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.rtx_last_error.restype = C.c_char_p
    _lib = L
    return L


def _check(rc):
    if rc != abi.RTX_OK:
        raise RtxError(rc, lib().rtx_last_error().decode())


def init(device=-1):
    _check(lib().rtx_init(device))


def device_count():
    return lib().rtx_device_count()


class DeviceScene:
    """One frame's scene resident in HBM (rtx_scene_create .. rtx_scene_destroy)."""

    def __init__(self, desc, stream=None, static=False, _clone_of=None):
        """static: rtx_scene_create_static -- for a scene rendered many times (more host time once,
        a cheaper membership check per hit; same results)."""
        self._L = lib()
        h = C.c_void_p()
        if _clone_of is not None:
            _check(self._L.rtx_scene_clone(_clone_of._h, C.byref(h)))
        elif static:
            _check(self._L.rtx_scene_create_static(C.byref(desc), C.byref(h)))
        else:
            _check(self._L.rtx_scene_create(C.byref(desc), C.byref(h)))
        self._h = h
        self._async = False
        if stream is not None:
            _check(self._L.rtx_scene_set_stream(self._h, C.c_void_p(stream)))

    def clone(self, stream=None):
        """rtx_scene_clone: a handle with its own stream, queues and statistics on the same
        device copy of the scene (frames in flight on several streams)."""
        return DeviceScene(None, stream=stream, _clone_of=self)

    def set_async(self, enabled=True):
        """rtx_scene_set_async: render_device / present_device return once their work is queued;
        wait() returns the statistics of the frames since the last hand-out."""
        _check(self._L.rtx_scene_set_async(self._h, int(bool(enabled))))
        self._async = bool(enabled)

    def wait(self):
        """rtx_scene_wait: wait for everything queued on the scene's stream; summed Stats."""
        stats = abi.Stats()
        _check(self._L.rtx_scene_wait(self._h, C.byref(stats)))
        return stats

    def close(self):
        if getattr(self, "_h", None):
            self._L.rtx_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def trace_closest(self, origin, direction, flags=0, leaf_seq_cap=0, want_counters=False):
        """Batched shootRay.  Returns a dict of numpy arrays (same keys as the oracle binding)."""
        origin = np.ascontiguousarray(origin, dtype=np.float64)
        direction = np.ascontiguousarray(direction, dtype=np.float64)
        n = origin.size // 3
        assert direction.size == n * 3
        prim = np.zeros(n, np.int32)
        depth = np.zeros(n)
        position = np.zeros((n, 3))
        normal = np.zeros((n, 3))
        uv = np.zeros((n, 2))
        bary = np.zeros((n, 3))
        leaf_count = np.zeros(n, np.uint32)
        leaf_hash = np.zeros(n, np.uint64)
        leaf_seq = np.full((n, max(leaf_seq_cap, 1)), 0xFFFFFFFF, np.uint32)
        counters = np.zeros((n, 6), np.uint32)
        p = lambda a: a.ctypes.data_as(PD)
        _check(self._L.rtx_trace_closest(
            self._h, p(origin), p(direction), n, flags, prim.ctypes.data_as(C.POINTER(C.c_int32)), p(depth),
            p(position), p(normal), p(uv), p(bary), leaf_count.ctypes.data_as(C.POINTER(C.c_uint32)),
            leaf_hash.ctypes.data_as(C.POINTER(C.c_uint64)),
            leaf_seq.ctypes.data_as(C.POINTER(C.c_uint32)) if leaf_seq_cap else None, leaf_seq_cap,
            counters.ctypes.data_as(C.POINTER(C.c_uint32)) if want_counters else None))
        return dict(prim=prim, depth=depth, position=position, normal=normal, uv=uv, bary=bary,
                    leaf_count=leaf_count, leaf_hash=leaf_hash, leaf_seq=leaf_seq, counters=counters)

    def render(self, rd, accum=None, want_stats=True):
        """rtx_render on a host buffer.  Returns (accum float32 [tile_h, tile_w, 3], Stats)."""
        if accum is None:
            accum = np.zeros((rd.tile_h, rd.tile_w, 3), np.float32)
        assert accum.dtype == np.float32 and accum.size == rd.tile_w * rd.tile_h * 3 and accum.flags.c_contiguous
        stats = abi.Stats()
        _check(self._L.rtx_render(self._h, C.byref(rd), accum.ctypes.data_as(C.POINTER(C.c_float)),
                                  C.byref(stats) if want_stats else None))
        return accum, stats

    def render_device(self, rd, accum_ptr, want_stats=True):
        """rtx_render_device on device memory (e.g. a torch tensor's data_ptr()).  On an
        asynchronous scene nothing is returned: the statistics come from wait()."""
        if self._async:
            _check(self._L.rtx_render_device(self._h, C.byref(rd), C.c_void_p(accum_ptr), None))
            return None
        stats = abi.Stats()
        _check(self._L.rtx_render_device(self._h, C.byref(rd), C.c_void_p(accum_ptr),
                                         C.byref(stats) if want_stats else None))
        return stats

    def render_device_noalloc(self, rd, accum_ptr):
        """rtx_render_device_noalloc: queued on the scene's stream, nothing allocated, nothing
        returned; the caller synchronises the stream before reading the framebuffer."""
        _check(self._L.rtx_render_device_noalloc(self._h, C.byref(rd), C.c_void_p(accum_ptr)))

    def present(self, accum, passes=1, tonemap=abi.TONEMAP_CBRT, flip=True):
        """rtx_present: [H, W, 3] float accumulation buffer (row 0 = bottom) -> [H, W, 3] uint8 sRGB."""
        accum = np.ascontiguousarray(accum, dtype=np.float32)
        h, w, _ = accum.shape
        out = np.empty((h, w, 3), dtype=np.uint8)
        _check(self._L.rtx_present(self._h, accum.ctypes.data_as(C.POINTER(C.c_float)), w, h, passes, tonemap, int(flip),
                                   out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def present_device(self, accum_ptr, width, height, rgb8_ptr, passes=1, tonemap=abi.TONEMAP_CBRT, flip=True):
        """rtx_present_device on device memory."""
        _check(self._L.rtx_present_device(self._h, C.c_void_p(accum_ptr), width, height, passes, tonemap, int(flip),
                                          C.c_void_p(rgb8_ptr)))


def uniform(seed, frame, pixel, sample, branch, draw):
    return lib().rtx_uniform(seed, frame, pixel, sample, branch, draw)
