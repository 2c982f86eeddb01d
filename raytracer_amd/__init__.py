"""raytracer_amd -- MI355X-native backend for the ray-scene intersection and shading loop of Dynamitos/RayTracer.

The product is native: raytracer_amd/csrc (hand-written gfx950 kernels + the C ABI of include/rthip.h) and
raytracer_amd/src (host C++ mirroring the reference's src/scene, src/util and a new src/hip backend, with the C ABI
of include/raytracer_c.h).  This Python package is only the ctypes binding used by tests, bench.py and the
multi-GPU launcher.
"""
from . import capi, scenes  # noqa: F401
from .capi import Camera, Mesh, Probe, Renderer, RtError, make_camera  # noqa: F401
