"""Frame sequences of an animated scene (the reference's OUTPUTFRAMES loop,
Raytracer/Raytracer.cpp:595-752): per frame redoAABBs + buildKDTree on the host
(:634-636), upload, then the pass loop on the GPU.

The host part of frame f+1 (redoAABBs, KD build, flatten, BVH build and upload inside
rtx_scene_create) does not depend on the GPU work of frame f, so a worker thread prepares
the next frame's device scene while the current one renders; the C calls release the GIL.
The reference rebuilds its tree inside the frame loop, serially.
"""
import threading
import time

from . import device


class _Prepared:
    def __init__(self):
        self.scene = None
        self.error = None
        self.host_s = 0.0
        self.done = threading.Event()


def _prepare(host_scene, t, stream, slot):
    try:
        t0 = time.perf_counter()
        desc = host_scene.frame(t)                  # valid until the next frame() call on this scene
        slot.scene = device.DeviceScene(desc, stream=stream)    # copies everything to HBM
        slot.host_s = time.perf_counter() - t0
    except Exception as e:                          # handed to the consumer
        slot.error = e
    finally:
        slot.done.set()


def device_scenes(host_scene, times, stream=None, prefetch=True):
    """Yield (index, t, DeviceScene, host_seconds) for every time in `times`, in order.

    With `prefetch` the next frame's scene is prepared on a worker thread while the caller
    works with the current one.  Each DeviceScene is closed when the caller asks for the next.
    Only one frame() call is in flight at a time, as the host scene is not thread-safe."""
    times = list(times)
    if not times:
        return
    nxt = _Prepared()
    worker = threading.Thread(target=_prepare, args=(host_scene, times[0], stream, nxt))
    worker.start()
    started = True
    try:
        for k, t in enumerate(times):
            if not started:
                worker.start()
                started = True
            worker.join()
            cur, nxt = nxt, None
            if cur.error is not None:
                raise cur.error
            if k + 1 < len(times):
                nxt = _Prepared()
                worker = threading.Thread(target=_prepare, args=(host_scene, times[k + 1], stream, nxt))
                started = prefetch
                if prefetch:
                    worker.start()
            try:
                yield k, t, cur.scene, cur.host_s
            finally:
                cur.scene.close()
    finally:
        # a consumer that stops early may leave one prepared scene behind
        if started and worker.is_alive():
            worker.join()
        if nxt is not None and nxt.scene is not None:
            nxt.scene.close()
