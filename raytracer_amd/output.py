"""Headless output stage: what the reference does between drawBuffer and the PNG files
(Raytracer/Raytracer.cpp:720-726 tonemap, :422 sRGB framebuffer, :90-105 saveImage,
:582-590 and :748 file naming), without a window.
"""
import os
import time

from . import abi, host


class FrameWriter:
    """out/<UTC %F-%H-%M-%S>/<frame>.png, as OUTPUTFRAMES names them (Raytracer.cpp:582-590,748)."""

    def __init__(self, root="out", stamp=None):
        self.directory = os.path.join(root, stamp or time.strftime("%Y-%m-%d-%H-%M-%S", time.gmtime()))
        os.makedirs(self.directory, exist_ok=True)

    def path(self, frame):
        return os.path.join(self.directory, f"{frame}.png")

    def save(self, device_scene, frame, accum, passes=1, tonemap=abi.TONEMAP_CBRT):
        """accum: [H, W, 3] float32 accumulation buffer of `passes` passes, row 0 = bottom."""
        rgb8 = device_scene.present(accum, passes=passes, tonemap=tonemap, flip=True)
        host.write_png(self.path(frame), rgb8)
        return self.path(frame)
