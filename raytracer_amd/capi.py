"""ctypes binding of the product's C ABIs (include/raytracer_c.h, include/rthip.h).

Plumbing only: every call goes straight into libraytracer_amd.so / librthip.so, which are built in-tree by
`make -C raytracer_amd` (see __graft_entry__.build).  There is no CPU fallback: if the libraries are missing the
import of this module raises, and a render without a GPU fails with RTHIP_E_NODEVICE.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_HOST = os.path.join(_HERE, "libraytracer_amd.so")
LIB_HIP = os.path.join(_HERE, "librthip.so")

RTHIP_E_ARG, RTHIP_E_STATE, RTHIP_E_BOUNDS, RTHIP_E_NODEVICE, RTHIP_E_LIMIT, RTHIP_E_COMM = -1, -2, -3, -4, -5, -6
RTHIP_COMM_ID_BYTES = 128
RTHIP_STATIC_MAX_RAYS = 16384  # include/rthip.h


class RtError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what}: {error_string(code)} ({code})")


class Camera(C.Structure):  # rthip_camera == the reference's Camera (src/util/Camera.h:4-12)
    _fields_ = [("position", C.c_float * 3), ("target", C.c_float * 3), ("sensorSize", C.c_float * 2),
                ("S_O", C.c_float), ("f", C.c_float), ("A", C.c_float)]


class FrameDesc(C.Structure):  # rthip_frame_desc
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("numSamples", C.c_uint32), ("maxDepth", C.c_uint32),
                ("aovDepths", C.c_uint32), ("tileSize", C.c_uint32), ("numTiles", C.c_uint32), ("pad", C.c_uint32),
                ("tileIds", C.POINTER(C.c_uint32))]


class Counters(C.Structure):  # rthip_counters
    _fields_ = [("extendPrimary", C.c_uint64), ("extendSecondary", C.c_uint64), ("shadow", C.c_uint64),
                ("nodesVisited", C.c_uint64), ("trisTested", C.c_uint64), ("nonFiniteAccepts", C.c_uint64),
                ("hits", C.c_uint64 * 16), ("nodesVisitedExtend", C.c_uint64), ("trisTestedExtend", C.c_uint64),
                ("waveNodeSteps", C.c_uint64), ("waveLeafSteps", C.c_uint64),
                ("recordsFetchedExtend", C.c_uint64), ("recordsFetchedShadow", C.c_uint64)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "hits"}
        d["hits"] = [int(x) for x in self.hits]
        return d


class FlatInfo(C.Structure):  # rtflat_info
    _fields_ = [("numModelNodes", C.c_uint32), ("numNodes", C.c_uint32), ("numTris", C.c_uint32), ("rootModelNode", C.c_uint32),
                ("sceneLo", C.c_float * 3), ("sceneHi", C.c_float * 3), ("padAbs", C.c_float), ("originBound", C.c_float),
                ("cullRel", C.c_float), ("cullAbs", C.c_float), ("maxDepthBlas", C.c_uint32), ("numDegenerate", C.c_uint32),
                ("pad", C.c_uint32 * 2)]


# numpy views of the 64-byte records of include/rtflat.h
NODE_DTYPE = np.dtype([("origin", "<f4", 3), ("exp", "u1", 3), ("numChildren", "u1"), ("child", "<u4", 4), ("qlo", "u1", (3, 4)),
                       ("qhi", "u1", (3, 4)), ("pad", "<u4", 2)])
MODEL_NODE_DTYPE = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("left", "<u4"), ("right", "<u4"), ("blas", "<u4"), ("triCount", "<u4"),
                             ("indicesOffset", "<u4"), ("orderBase", "<u4"), ("modelIndex", "<u4"), ("subtreeSize", "<u4"),
                             ("pad", "<u4", 2)])
TRI_DTYPE = np.dtype([("p0", "<f4", 3), ("e0", "<f4", 3), ("e1", "<f4", 3), ("n", "<f4", 3), ("triId", "<u4"), ("orderKey", "<u4"),
                      ("pad", "<u4", 2)])
assert NODE_DTYPE.itemsize == 64 and MODEL_NODE_DTYPE.itemsize == 64 and TRI_DTYPE.itemsize == 64
# the shading tables of include/rtflat.h (scene option "materials")
SHADE_DTYPE = np.dtype([("uv", "<f4", (3, 2)), ("material", "<u4"), ("pad", "<u4")])
MATERIAL_DTYPE = np.dtype([("baseColor", "<f4", 3), ("kind", "<u4"), ("emissive", "<f4", 3), ("pad0", "<u4"),
                           ("albedoTexel", "<u4"), ("albedoW", "<u4"), ("albedoH", "<u4"), ("pad1", "<u4"),
                           ("emissiveTexel", "<u4"), ("emissiveW", "<u4"), ("emissiveH", "<u4"), ("pad2", "<u4")])
assert SHADE_DTYPE.itemsize == 32 and MATERIAL_DTYPE.itemsize == 64
MAT_STUB, MAT_FACTORS = 0, 1

_host = None
_hip = None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def hip_lib():
    global _hip
    if _hip is None:
        if not os.path.exists(LIB_HIP):
            raise ImportError(f"{LIB_HIP} is missing: run `make -C raytracer_amd` (no CPU fallback exists)")
        L = C.CDLL(LIB_HIP, mode=C.RTLD_GLOBAL)
        fp, up, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
        L.rthip_error_string.restype = C.c_char_p
        L.rthip_error_string.argtypes = [C.c_int]
        L.rthip_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rthip_destroy.argtypes = [vp]
        L.rthip_probe_rand01.argtypes = [vp, up, C.c_uint32, fp]
        L.rthip_probe_sincos.argtypes = [vp, fp, C.c_uint32, fp, fp]
        L.rthip_probe_primary_rays.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, up, C.c_uint32, fp, fp]
        L.rthip_probe_bounce.argtypes = [vp, fp, fp, C.c_uint32, fp]
        L.rthip_probe_trace.argtypes = [vp, fp, fp, C.c_uint32, C.c_float, C.c_float, up, fp, C.POINTER(C.c_uint8)]
        L.rthip_trace_rays_static.argtypes = [vp, vp, vp, C.c_uint32, C.c_float, C.c_float, vp, vp, vp]
        L.rthip_comm_unique_id.argtypes = [C.c_char_p]
        L.rthip_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        L.rthip_comm_destroy.argtypes = [vp]
        L.rthip_gather.argtypes = [vp, vp, C.c_uint64, C.c_int]
        L.rthip_accumulator_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.rthip_untile.argtypes = [vp, vp, up, C.c_uint32, vp]
        L.rthip_copy_accumulator_to.argtypes = [vp, vp, C.c_uint64]
        L.rthip_render_passes.argtypes = [vp, C.c_uint32, C.c_uint32, fp]
        L.rthip_enqueue_passes.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.rthip_synchronize.argtypes = [vp]
        L.rthip_get_counters.argtypes = [vp, C.POINTER(Counters)]
        L.rthip_get_kernel_times.argtypes = [vp, fp, up]
        L.rthip_upload_materials.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint64]
        L.rthip_set_point_shadows.argtypes = [vp, C.c_int]
        L.rthip_set_build_quality.argtypes = [vp, C.c_int]
        L.rthip_refit.argtypes = [vp, vp, vp, fp]
        L.rthip_download_scene.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
        L.rthip_begin.argtypes = [vp, C.POINTER(Camera), C.POINTER(FrameDesc)]
        L.rthip_begin_dynamic.argtypes = [vp, C.POINTER(Camera), C.POINTER(FrameDesc)]
        L.rthip_enqueue_tiles.argtypes = [vp, up, C.c_uint32]
        L.rthip_wait_tiles.argtypes = [vp, C.c_uint32]
        L.rthip_get_local_tiles.argtypes = [vp, up, C.c_uint32, C.POINTER(C.c_uint32)]
        L.rthip_gather_tiles.argtypes = [vp, vp, C.c_uint64, up, C.c_uint32, up, C.c_int]
        L.rthip_set_refill_mode.argtypes = [vp, C.c_int]
        L.rthip_get_refill_mode.argtypes = [vp, C.POINTER(C.c_int), fp]
        L.rthip_get_grid_blocks.argtypes = [vp, C.POINTER(C.c_uint32)]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(LIB_HOST):
            raise ImportError(f"{LIB_HOST} is missing: run `make -C raytracer_amd` (no CPU fallback exists)")
        L = C.CDLL(LIB_HOST)
        fp, up, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
        for name in ("rt_mesh_cube", "rt_mesh_soup", "rt_mesh_heightfield", "rt_mesh_load_obj", "rt_mesh_from_arrays", "rt_context"):
            getattr(L, name).restype = vp
        L.rt_mesh_soup.argtypes = [C.c_uint32, C.c_uint64, C.c_float]
        L.rt_mesh_heightfield.argtypes = [C.c_uint32, C.c_uint64, C.c_float]
        L.rt_mesh_load_obj.argtypes = [C.c_char_p]
        L.rt_mesh_load_all.restype = C.c_uint32
        L.rt_mesh_load_all.argtypes = [C.c_char_p, C.POINTER(vp), C.c_uint32]
        L.rt_mesh_from_arrays.argtypes = [fp, C.c_uint32, fp, up, C.c_uint32]
        L.rt_mesh_num_positions.restype = C.c_uint32
        L.rt_mesh_num_positions.argtypes = [vp]
        L.rt_mesh_num_triangles.restype = C.c_uint32
        L.rt_mesh_num_triangles.argtypes = [vp]
        L.rt_mesh_copy.argtypes = [vp, fp, up]
        L.rt_mesh_copy_texcoords.argtypes = [vp, fp]
        L.rt_mesh_set_material.argtypes = [vp, C.c_uint32, fp, fp, fp, C.c_uint32, C.c_uint32, fp, C.c_uint32, C.c_uint32]
        L.rt_mesh_get_material.argtypes = [vp, up, fp, fp]
        L.rt_get_flat_shading.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), up, C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.rt_mesh_destroy.argtypes = [vp]
        L.rt_renderer_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.rt_renderer_destroy.argtypes = [vp]
        L.rt_add_directional_light.argtypes = [vp, fp, fp]
        L.rt_add_point_light.argtypes = [vp, fp, fp, C.c_float]
        L.rt_add_mesh.argtypes = [vp, vp, fp]
        L.rt_generate.argtypes = [vp]
        L.rt_set_option.argtypes = [vp, C.c_char_p, C.c_double]
        L.rt_set_tiles.argtypes = [vp, up, C.c_uint32]
        L.rt_start_render.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32]
        L.rt_begin_frame.argtypes = [vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32]
        L.rt_render_passes.argtypes = [vp, C.c_uint32, C.c_uint32, fp]
        L.rt_enqueue_passes.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.rt_synchronize.argtypes = [vp]
        L.rt_wait_render.argtypes = [vp]
        L.rt_cancel_render.argtypes = [vp]
        L.rt_get_sample_times.restype = C.c_uint32
        L.rt_get_sample_times.argtypes = [vp, fp, C.c_uint32]
        L.rt_read_accumulator.argtypes = [vp, fp]
        L.rt_read_image.argtypes = [vp, fp]
        L.rt_save_image.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32]
        L.rt_write_ppm.argtypes = [C.c_char_p, fp, C.c_uint32, C.c_uint32]
        L.rt_read_aov.argtypes = [vp, C.c_uint32, C.c_uint32, up, C.POINTER(C.c_uint8)]
        L.rt_get_counters.argtypes = [vp, C.POINTER(Counters)]
        L.rt_get_kernel_times.argtypes = [vp, fp, up]
        L.rt_get_flat.argtypes = [vp, C.POINTER(C.POINTER(FlatInfo)), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_double)]
        L.rt_context.argtypes = [vp]
        L.rt_get_build_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        L.rt_gather.argtypes = [vp, vp, C.c_uint64, C.c_int]
        _host = L
    return _host


def error_string(code):
    return hip_lib().rthip_error_string(int(code)).decode()


def _check(code, what):
    if code != 0:
        raise RtError(code, what)


def make_camera(position, target, S_O=20.0, f=0.7, A=0.35, sensor=(0.036, 0.024)):
    """Defaults as src/util/Camera.h:8-11; the reference's main.cpp:18-24 uses f=0, A=0, S_O=6."""
    c = Camera()
    c.position[:] = [float(x) for x in position]
    c.target[:] = [float(x) for x in target]
    c.sensorSize[:] = [float(np.float32(sensor[0])), float(np.float32(sensor[1]))]
    c.S_O, c.f, c.A = float(S_O), float(f), float(A)
    return c


IDENTITY = np.eye(4, dtype=np.float32)


class Mesh:
    """What ModelLoader::loadModel returns for one aiMesh (src/util/ModelLoader.cpp:14-41)."""

    def __init__(self, handle):
        if not handle:
            raise RtError(RTHIP_E_ARG, "mesh creation")
        self._h = handle
        self.material_spec = None  # the arguments of the last set_material (so a second consumer can be given the same)

    @classmethod
    def cube(cls):
        return cls(host_lib().rt_mesh_cube())

    @classmethod
    def soup(cls, num_triangles, seed, edge_radius=0.02):
        return cls(host_lib().rt_mesh_soup(int(num_triangles), int(seed), float(edge_radius)))

    @classmethod
    def heightfield(cls, quads_per_side, seed, amplitude=0.15):
        return cls(host_lib().rt_mesh_heightfield(int(quads_per_side), int(seed), float(amplitude)))

    @classmethod
    def load_obj(cls, path):
        return cls(host_lib().rt_mesh_load_obj(os.fsencode(path)))

    @classmethod
    def load_all(cls, path):
        """Every model of a file (.obj, .glb), in file order -- what ModelLoader::loadModel returns."""
        n = host_lib().rt_mesh_load_all(os.fsencode(path), None, 0)
        if n == 0:
            raise RtError(RTHIP_E_ARG, "no model could be read from %s" % path)
        arr = (C.c_void_p * n)()
        host_lib().rt_mesh_load_all(os.fsencode(path), arr, n)
        return [cls(h) for h in arr]

    @classmethod
    def from_arrays(cls, positions, indices, texcoords=None):
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        tc = None
        if texcoords is not None:
            t = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
            assert t.shape[0] == p.shape[0]
            tc = _fp(t)
        return cls(host_lib().rt_mesh_from_arrays(_fp(p), p.shape[0], tc, _up(i), i.shape[0]))

    def set_material(self, kind=MAT_FACTORS, base_color=(0, 1, 0), emissive=(0, 0, 0), albedo_texture=None, emissive_texture=None):
        """The mesh's Material (src/util/Material.h:7-14); textures are float32 arrays [h, w, 3] or None.  Read by the path
        only when the scene option "materials" is set."""
        def tex(t):
            if t is None:
                return None, 0, 0
            a = np.ascontiguousarray(t, np.float32)
            assert a.ndim == 3 and a.shape[2] == 3
            return _fp(a), a.shape[1], a.shape[0]
        b, e = np.asarray(base_color, np.float32), np.asarray(emissive, np.float32)
        ap, aw, ah = tex(albedo_texture)
        ep, ew, eh = tex(emissive_texture)
        _check(host_lib().rt_mesh_set_material(self._h, int(kind), _fp(b), _fp(e), ap, aw, ah, ep, ew, eh), "rt_mesh_set_material")
        self.material_spec = dict(kind=int(kind), base_color=b.copy(), emissive=e.copy(), albedo_texture=albedo_texture, emissive_texture=emissive_texture)

    def material(self):
        """(kind, base colour, emissive) -- for a loaded .glb mesh its primitive's glTF factors"""
        k = C.c_uint32()
        b, e = np.zeros(3, np.float32), np.zeros(3, np.float32)
        _check(host_lib().rt_mesh_get_material(self._h, C.byref(k), _fp(b), _fp(e)), "rt_mesh_get_material")
        return int(k.value), b, e

    def texcoords(self):
        t = np.zeros((self.num_positions, 2), np.float32)
        host_lib().rt_mesh_copy_texcoords(self._h, _fp(t))
        return t

    @property
    def num_positions(self):
        return host_lib().rt_mesh_num_positions(self._h)

    @property
    def num_triangles(self):
        return host_lib().rt_mesh_num_triangles(self._h)

    def arrays(self):
        p = np.zeros((self.num_positions, 3), np.float32)
        i = np.zeros((self.num_triangles, 3), np.uint32)
        host_lib().rt_mesh_copy(self._h, _fp(p), _up(i))
        return p, i

    def close(self):
        if self._h:
            host_lib().rt_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """The reference's Renderer interface (src/scene/Renderer.h:14-42) with the `hip` backend behind it."""

    def __init__(self, backend="hip", device=0):
        h = C.c_void_p()
        _check(host_lib().rt_renderer_create(backend.encode(), int(device), C.byref(h)), "rt_renderer_create")
        self._h = h
        self._size = None

    def close(self):
        if self._h:
            host_lib().rt_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_directional_light(self, direction, color=(1, 1, 1)):
        d, c = np.asarray(direction, np.float32), np.asarray(color, np.float32)
        _check(host_lib().rt_add_directional_light(self._h, _fp(d), _fp(c)), "rt_add_directional_light")

    def add_point_light(self, position=(0, 0, 0), color=(1, 1, 1), attenuation=1.0):
        p, c = np.asarray(position, np.float32), np.asarray(color, np.float32)
        _check(host_lib().rt_add_point_light(self._h, _fp(p), _fp(c), float(attenuation)), "rt_add_point_light")

    def add_mesh(self, mesh, transform=IDENTITY):
        t = np.ascontiguousarray(np.asarray(transform, np.float32).T)  # glm::mat4 is column-major
        _check(host_lib().rt_add_mesh(self._h, mesh._h, _fp(t)), "rt_add_mesh")

    def set_option(self, name, value):
        _check(host_lib().rt_set_option(self._h, name.encode(), float(value)), f"rt_set_option({name})")

    def set_tiles(self, tile_ids):
        t = np.ascontiguousarray(tile_ids, np.uint32)
        _check(host_lib().rt_set_tiles(self._h, _up(t), t.size), "rt_set_tiles")

    def generate(self):
        _check(host_lib().rt_generate(self._h), "rt_generate")

    def start_render(self, camera, width, height, num_samples):
        self._size = (int(height), int(width))
        _check(host_lib().rt_start_render(self._h, C.byref(camera), int(width), int(height), int(num_samples)), "rt_start_render")

    def wait(self):
        _check(host_lib().rt_wait_render(self._h), "render")

    def cancel(self):
        _check(host_lib().rt_cancel_render(self._h), "rt_cancel_render")

    def render(self, camera, width, height, num_samples):
        self.start_render(camera, width, height, num_samples)
        self.wait()

    def begin_frame(self, camera, width, height, num_samples):
        self._size = (int(height), int(width))
        _check(host_lib().rt_begin_frame(self._h, C.byref(camera), int(width), int(height), int(num_samples)), "rt_begin_frame")

    def enqueue_passes(self, begin, count):
        """Queues sample passes [begin, begin+count) without waiting; synchronize() or any read-back waits for them."""
        _check(host_lib().rt_enqueue_passes(self._h, int(begin), int(count)), "rt_enqueue_passes")

    def synchronize(self):
        _check(host_lib().rt_synchronize(self._h), "rt_synchronize")

    def render_passes(self, begin, count):
        """Runs sample passes [begin, begin+count) synchronously; returns the device milliseconds of the call."""
        ms = C.c_float(0)
        _check(host_lib().rt_render_passes(self._h, int(begin), int(count), C.byref(ms)), "rt_render_passes")
        return float(ms.value)

    def sample_times(self):
        n = host_lib().rt_get_sample_times(self._h, None, 0)
        out = np.zeros(max(n, 1), np.float32)
        host_lib().rt_get_sample_times(self._h, _fp(out), n)
        return out[:n]

    def accumulator(self):
        h, w = self._size
        out = np.zeros((h, w, 3), np.float32)
        _check(host_lib().rt_read_accumulator(self._h, _fp(out)), "rt_read_accumulator")
        return out

    def save_image(self, path):
        """Writes the tonemapped image of the current frame as a binary PPM."""
        h, w = self._size
        _check(host_lib().rt_save_image(self._h, os.fsencode(path), w, h), "rt_save_image")

    def image(self):
        h, w = self._size
        out = np.zeros((h, w, 3), np.float32)
        _check(host_lib().rt_read_image(self._h, _fp(out)), "rt_read_image")
        return out

    def aov(self, samp, depth):
        h, w = self._size
        tri = np.full((h, w), 0xFFFFFFFF, np.uint32)
        sh = np.full((h, w), 0xFF, np.uint8)
        _check(host_lib().rt_read_aov(self._h, int(samp), int(depth), _up(tri), sh.ctypes.data_as(C.POINTER(C.c_uint8))), "rt_read_aov")
        return tri, sh

    def counters(self):
        c = Counters()
        _check(host_lib().rt_get_counters(self._h, C.byref(c)), "rt_get_counters")
        return c.as_dict()

    def kernel_times(self):
        ms = np.zeros(6, np.float32)
        n = np.zeros(6, np.uint32)
        _check(host_lib().rt_get_kernel_times(self._h, _fp(ms), _up(n)), "rt_get_kernel_times")
        names = ("generate", "extend", "shade", "shadow", "accumulate", "compact")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def flat(self):
        """Views (no copy) of the flattened scene: info, model_nodes, nodes, tris, build_seconds."""
        info = C.POINTER(FlatInfo)()
        mn, nd, tr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        secs = C.c_double()
        _check(host_lib().rt_get_flat(self._h, C.byref(info), C.byref(mn), C.byref(nd), C.byref(tr), C.byref(secs)), "rt_get_flat")
        i = info.contents

        def view(ptr, count, dtype):
            buf = (C.c_char * (64 * count)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dtype, count=count)

        # the views alias memory owned by the renderer: keep it alive for as long as the dict lives
        return dict(_owner=self, info=i, info_bytes=bytes(C.string_at(C.addressof(i), C.sizeof(FlatInfo))),
                    model_nodes=view(mn, i.numModelNodes, MODEL_NODE_DTYPE), nodes=view(nd, max(i.numNodes, 1), NODE_DTYPE),
                    tris=view(tr, max(i.numTris, 1), TRI_DTYPE), build_seconds=secs.value)

    def flat_shading(self):
        """Views (no copy) of the shading tables handed to the device: shade[numTris], materials[numModels], texels[n, 3];
        None unless the scene option "materials" was set."""
        sh, mt, tx = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nm, nt = C.c_uint32(), C.c_uint64()
        _check(host_lib().rt_get_flat_shading(self._h, C.byref(sh), C.byref(mt), C.byref(nm), C.byref(tx), C.byref(nt)), "rt_get_flat_shading")
        if not sh.value:
            return None
        n_tris = self.flat()["info"].numTris

        def view(ptr, count, dtype):
            buf = (C.c_char * (dtype.itemsize * count)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dtype, count=count)

        texels = view(tx, 3 * nt.value, np.dtype("<f4")).reshape(-1, 3) if tx.value else np.zeros((0, 3), np.float32)
        return dict(_owner=self, shade=view(sh, n_tris, SHADE_DTYPE), materials=view(mt, nm.value, MATERIAL_DTYPE), texels=texels)

    # ---- dynamic tile ownership (include/rthip.h: rthip_begin_dynamic / rthip_enqueue_tiles / rthip_wait_tiles)
    def begin_dynamic(self, camera, width, height, num_samples, max_depth, capacity=0, tile_size=64):
        """A frame whose tiles arrive while it renders (at most `capacity` of them on this rank; 0 = all tiles of the image)."""
        self._size = (int(height), int(width))
        d = FrameDesc(int(width), int(height), int(num_samples), int(max_depth), 0, int(tile_size), int(capacity), 0, None)
        _check(hip_lib().rthip_begin_dynamic(self.context, C.byref(camera), C.byref(d)), "rthip_begin_dynamic")

    def enqueue_tiles(self, tile_ids, max_in_flight=None):
        """Queues every sample pass of these tiles (no waiting); with max_in_flight, then blocks until at most that many such
        calls are unfinished -- the back-pressure that makes drawing tiles from a shared cursor balance the ranks."""
        t = np.ascontiguousarray(tile_ids, np.uint32)
        _check(hip_lib().rthip_enqueue_tiles(self.context, _up(t), t.size), "rthip_enqueue_tiles")
        if max_in_flight is not None:
            _check(hip_lib().rthip_wait_tiles(self.context, int(max_in_flight)), "rthip_wait_tiles")

    def local_tiles(self):
        n = C.c_uint32(0)
        _check(hip_lib().rthip_get_local_tiles(self.context, None, 0, C.byref(n)), "rthip_get_local_tiles")
        ids = np.zeros(max(1, n.value), np.uint32)
        _check(hip_lib().rthip_get_local_tiles(self.context, _up(ids), ids.size, C.byref(n)), "rthip_get_local_tiles")
        return ids[: n.value]

    def refit(self, moved):
        """rthip_refit: the scene takes the vertex positions of `moved` -- a renderer (any backend, "hip-offline" will do) holding
        the SAME meshes, triangle for triangle, with moved vertices -- without rebuilding its triangle-level tree.  Returns the
        device milliseconds."""
        f = moved.flat()
        order = np.argsort(f["tris"]["triId"], kind="stable")  # leaf order -> pool order
        tris = np.ascontiguousarray(f["tris"][order])
        nodes = np.ascontiguousarray(f["model_nodes"])
        ms = C.c_float(0)
        _check(hip_lib().rthip_refit(self.context, nodes.ctypes.data, tris.ctypes.data, C.byref(ms)), "rthip_refit")
        return float(ms.value)

    def refill_mode(self):
        """(traversal mode in use: 0 low / 1 high refill thresholds / 2 packet traversal of camera and first shadow rays / 3 of camera
        rays only, [ms of the last calibration: settings 0, 1, 2; the fourth is unused] -- zeros when none was taken)"""
        m = C.c_int()
        ms = np.zeros(4, np.float32)
        _check(hip_lib().rthip_get_refill_mode(self.context, C.byref(m), _fp(ms)), "rthip_get_refill_mode")
        return int(m.value), ms

    def grid_blocks(self):
        """resident blocks per CU of the traversal kernel's launches over slot lists (depth >= 1): 3 unless RTHIP_LIST_BLOCKS says otherwise"""
        b = C.c_uint32()
        _check(hip_lib().rthip_get_grid_blocks(self.context, C.byref(b)), "rthip_get_grid_blocks")
        return int(b.value)

    def build_times(self):
        """(host seconds of the flattener, device milliseconds of the BVH build -- 0 unless option deviceBvh was set)"""
        a, b = C.c_double(), C.c_double()
        _check(host_lib().rt_get_build_times(self._h, C.byref(a), C.byref(b)), "rt_get_build_times")
        return a.value, b.value

    @property
    def context(self):
        return C.c_void_p(host_lib().rt_context(self._h))


class Probe:
    """Device-arithmetic probes of rthip.h (tests compare them bit for bit with the CPU oracle)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def rand01(self, xyz):
        a = np.ascontiguousarray(xyz, np.uint32).reshape(-1, 3)
        out = np.zeros((a.shape[0], 3), np.float32)
        _check(hip_lib().rthip_probe_rand01(self.ctx, _up(a), a.shape[0], _fp(out)), "rthip_probe_rand01")
        return out

    def sincos(self, x):
        a = np.ascontiguousarray(x, np.float32)
        s, c = np.zeros_like(a), np.zeros_like(a)
        _check(hip_lib().rthip_probe_sincos(self.ctx, _fp(a), a.size, _fp(s), _fp(c)), "rthip_probe_sincos")
        return s, c

    def primary_rays(self, camera, width, height, whs):
        a = np.ascontiguousarray(whs, np.uint32).reshape(-1, 3)
        o, d = np.zeros((a.shape[0], 3), np.float32), np.zeros((a.shape[0], 3), np.float32)
        _check(hip_lib().rthip_probe_primary_rays(self.ctx, C.byref(camera), width, height, _up(a), a.shape[0], _fp(o), _fp(d)),
               "rthip_probe_primary_rays")
        return o, d

    def bounce(self, normals, rnd):
        n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        r = np.ascontiguousarray(rnd, np.float32).reshape(-1, 3)
        out = np.zeros_like(n)
        _check(hip_lib().rthip_probe_bounce(self.ctx, _fp(n), _fp(r), n.shape[0], _fp(out)), "rthip_probe_bounce")
        return out

    def trace(self, origins, dirs, tmin=1e-4, tmax=1e20):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        tri, t, anyh = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros(n, np.uint8)
        _check(hip_lib().rthip_probe_trace(self.ctx, _fp(o), _fp(d), n, float(np.float32(tmin)), float(np.float32(tmax)), _up(tri), _fp(t),
                                           anyh.ctypes.data_as(C.POINTER(C.c_uint8))), "rthip_probe_trace")
        return tri, t, anyh


def write_ppm(path, rgb):
    """Row-major (H, W, 3) floats in [0, 1] -> binary PPM (the encoder behind Renderer.save_image)."""
    a = np.ascontiguousarray(rgb, np.float32)
    _check(host_lib().rt_write_ppm(os.fsencode(path), _fp(a), a.shape[1], a.shape[0]), "rt_write_ppm")
