"""The C ABI: struct layouts of the ctypes mirrors match the C compiler's, the HIP library
loads without a GPU and exports every symbol include/rtx.h declares, and argument checks
that need no device report errors instead of crashing."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from raytracer_amd import abi, device

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_sizes_match_the_c_compiler(tmp_path):
    src = tmp_path / "sizes.c"
    names = ["rtx_sampler", "rtx_material", "rtx_texture", "rtx_light", "rtx_scene_desc", "rtx_camera",
             "rtx_render_desc", "rtx_stats"]
    body = "\n".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names)
    body += '\nprintf("off_camera %zu\\n", offsetof(rtx_render_desc, camera));'
    body += '\nprintf("off_lights %zu\\n", offsetof(rtx_scene_desc, lights));'
    body += '\nprintf("off_ms_total %zu\\n", offsetof(rtx_stats, ms_total));'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rtx.h"\nint main(void){\n' + body + "\nreturn 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    mirror = {"rtx_sampler": abi.Sampler, "rtx_material": abi.Material, "rtx_texture": abi.Texture,
              "rtx_light": abi.Light, "rtx_scene_desc": abi.SceneDesc, "rtx_camera": abi.Camera,
              "rtx_render_desc": abi.RenderDesc, "rtx_stats": abi.Stats}
    for n, cls in mirror.items():
        assert int(got[n]) == C.sizeof(cls), n
    assert int(got["off_camera"]) == abi.RenderDesc.camera.offset
    assert int(got["off_lights"]) == abi.SceneDesc.lights.offset
    assert int(got["off_ms_total"]) == abi.Stats.ms_total.offset


def test_flags_match_header():
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    for name, shift in re.findall(r"RTX_FLAG_(\w+)\s*=\s*1u << (\d+)", text):
        assert getattr(abi, "FLAG_" + name) == 1 << int(shift), name


def test_hip_library_loads_and_exports_every_declared_symbol():
    L = device.lib()
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    declared = set(re.findall(r"\b(rtx_[a-z_]+)\s*\(", text))
    assert declared == set(device.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_error_paths_without_a_gpu():
    L = device.lib()
    assert L.rtx_scene_create(None, C.byref(C.c_void_p())) == abi.RTX_ERR_INVALID
    assert L.rtx_scene_create_static(None, C.byref(C.c_void_p())) == abi.RTX_ERR_INVALID
    assert b"null" in L.rtx_last_error()
    bad = abi.SceneDesc()
    bad.abi_version = 99
    assert L.rtx_scene_create(C.byref(bad), C.byref(C.c_void_p())) == abi.RTX_ERR_INVALID
    assert b"abi_version" in L.rtx_last_error()
    assert L.rtx_render(None, None, None, None) == abi.RTX_ERR_INVALID


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure; nothing under raytracer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"import\s+oracle|from\s+oracle|librt_oracle|rt_oracle\.h|oracle/", text), (dirpath, f)


def test_no_gpu_means_loud_failure():
    if device.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(device.RtxError) as e:
        device.init(0)
    assert e.value.code == abi.RTX_ERR_NO_GPU
