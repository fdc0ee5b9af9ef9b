"""MI355X-native trace loop for noahwhygodwhy/Raytracer scenes.

Layout:
  abi      ctypes mirrors of include/rtx.h
  host     host scene API (C++ classes in host/, flattened per frame)
  device   the HIP library behind the C ABI (csrc/)
  scenes   scene recipes for the benchmark configurations
"""
__all__ = ["abi", "host", "scenes"]
