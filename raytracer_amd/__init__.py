"""MI355X-native trace loop for noahwhygodwhy/Raytracer scenes.

Layout:
  abi      ctypes mirrors of include/rtx.h
  host     host scene API (C++ classes in host/, flattened per frame)
  device   the HIP library behind the C ABI (csrc/)
  scenes   scene recipes for the benchmark configurations
  sequence animated frame sequences: the next frame's host work overlaps the current render
  dist     multi-GPU partitions (row bands of a frame, frames of a sequence) + gather
  output   present + PNG naming of the reference's OUTPUTFRAMES loop
"""
__all__ = ["abi", "host", "scenes"]
