"""The C-ABI libraries load without a GPU and export every entry point the headers declare."""
import ctypes
import os
import re

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    names = re.findall(r"\b(rt[a-z]*_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if not n.startswith("rtflat_")))


def test_rthip_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_HIP)
    names = declared_functions("rthip.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"librthip.so does not export {n}"


def test_host_library_exports_every_declared_symbol():
    capi.hip_lib()
    lib = ctypes.CDLL(capi.LIB_HOST)
    names = declared_functions("raytracer_c.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libraytracer_amd.so does not export {n}"


def test_record_layouts_are_64_bytes():
    text = open(os.path.join(ROOT, "include", "rtflat.h")).read()
    assert "RTFLAT_RECORD_BYTES 64u" in text
    assert capi.NODE_DTYPE.itemsize == capi.MODEL_NODE_DTYPE.itemsize == capi.TRI_DTYPE.itemsize == 64
    assert ctypes.sizeof(capi.FlatInfo) == 72
    assert ctypes.sizeof(capi.Camera) == 44  # the reference's Camera (src/util/Camera.h:4-12)


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.RtError) as e:
        capi.Renderer("hip")
    assert e.value.code == capi.RTHIP_E_NODEVICE


def test_lights_and_meshes_are_refused_after_generate():
    """Lights are uploaded with the scene (rt_generate): one added later would silently never shine, unlike the reference,
    which reads its light vectors live (src/cpu/CPUScene.cpp:31,39).  The call says so instead."""
    import numpy as np
    r = capi.Renderer("hip-offline")
    r.add_directional_light((0, -1, 0))
    r.add_mesh(capi.Mesh.cube())
    r.generate()
    import pytest
    with pytest.raises(capi.RtError) as e:
        r.add_directional_light((0, -1, 0))
    assert e.value.code == capi.RTHIP_E_STATE
    with pytest.raises(capi.RtError) as e:
        r.add_point_light()
    assert e.value.code == capi.RTHIP_E_STATE
    with pytest.raises(capi.RtError) as e:
        r.add_mesh(capi.Mesh.cube())
    assert e.value.code == capi.RTHIP_E_STATE
    r.close()
