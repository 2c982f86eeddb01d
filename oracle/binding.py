"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/liboracle_cpu.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (raytracer_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_cpu.so")


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("target", C.c_float * 3), ("sensorSize", C.c_float * 2),
                ("S_O", C.c_float), ("f", C.c_float), ("A", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "numSamples", "maxDepth", "x0", "y0", "x1", "y1",
                                          "sampBegin", "sampEnd", "numWorkers", "aovDepths", "useFlat", "pad")]


class Counters(C.Structure):
    _fields_ = [("extendPrimary", C.c_uint64), ("extendSecondary", C.c_uint64), ("shadow", C.c_uint64),
                ("nodesVisited", C.c_uint64), ("trisTested", C.c_uint64), ("nonFiniteAccepts", C.c_uint64),
                ("hits", C.c_uint64 * 16)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "hits"}
        d["hits"] = [int(x) for x in self.hits]
        return d


def build(force=False):
    """Compile the oracle with its committed Makefile (g++ only, no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_add_directional_light.argtypes = [C.c_void_p, fp, fp]
        L.orc_add_point_light.argtypes = [C.c_void_p, fp, fp, C.c_float]
        L.orc_add_model.argtypes = [C.c_void_p, fp, C.c_uint32, fp, C.POINTER(C.c_uint32), C.c_uint32, fp]
        L.orc_generate.argtypes = [C.c_void_p]
        L.orc_set_model_material.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, fp, fp, fp, C.c_uint32, C.c_uint32, fp, C.c_uint32, C.c_uint32]
        L.orc_set_shading.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_flat_shading.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_num_tris.restype = C.c_uint32
        L.orc_num_tris.argtypes = [C.c_void_p]
        L.orc_num_models.restype = C.c_uint32
        L.orc_num_models.argtypes = [C.c_void_p]
        L.orc_get_tri_operands.argtypes = [C.c_void_p, C.c_uint32, fp, fp, fp, fp]
        L.orc_get_model_tree.restype = C.c_uint32
        L.orc_get_model_tree.argtypes = [C.c_void_p, fp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]
        L.orc_set_flat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), fp, fp, C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint8), fp, C.POINTER(Counters)]
        L.orc_rand01.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, fp]
        L.orc_primary_ray.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, fp, fp, fp]
        L.orc_sincos.argtypes = [C.c_float, fp, fp]
        L.orc_bounce_dir.argtypes = [fp, fp, fp]
        L.orc_aabb_intersects.restype = C.c_int
        L.orc_aabb_intersects.argtypes = [fp, fp, fp, fp, C.c_float, C.c_float]
        L.orc_closest_hit.restype = C.c_uint32
        L.orc_closest_hit.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_float, C.c_float, fp]
        L.orc_any_hit.restype = C.c_int
        L.orc_any_hit.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_float, C.c_float]
        _lib = L
    return _lib


def _f32(x):
    return float(np.float32(x))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_camera(position, target, S_O=20.0, f=0.7, A=0.35, sensor=(0.036, 0.024)):
    """Defaults are Camera.h:8-11; the reference's main.cpp:18-24 uses f=0, A=0, S_O=6."""
    c = Camera()
    c.position[:] = [float(x) for x in position]
    c.target[:] = [float(x) for x in target]
    c.sensorSize[:] = [np.float32(sensor[0]), np.float32(sensor[1])]
    c.S_O, c.f, c.A = float(S_O), float(f), float(A)
    return c


IDENTITY = np.eye(4, dtype=np.float32)


class OracleScene:
    def __init__(self, fix_edge_offset=False):
        self._h = lib().orc_scene_create(1 if fix_edge_offset else 0)
        self._keep = []

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def add_directional_light(self, direction, color=(1, 1, 1)):
        lib().orc_add_directional_light(self._h, _f3(direction), _f3(color))

    def add_point_light(self, position=(0, 0, 0), color=(1, 1, 1), attenuation=1.0):
        lib().orc_add_point_light(self._h, _f3(position), _f3(color), float(attenuation))

    def add_model(self, positions, indices, transform=IDENTITY, texcoords=None):
        p = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        # glm::mat4 is column-major: element [c][r]; numpy matrices here are written row-major (math convention)
        t = np.ascontiguousarray(np.asarray(transform, dtype=np.float32).T)
        tc = None
        if texcoords is not None:
            tca = np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
            tc = _fptr(tca)
        lib().orc_add_model(self._h, _fptr(p), p.shape[0], tc, i.ctypes.data_as(C.POINTER(C.c_uint32)), i.shape[0], _fptr(t))

    def set_model_material(self, model_index, kind=1, base_color=(0, 1, 0), emissive=(0, 0, 0), albedo_texture=None,
                           emissive_texture=None):
        """Textures: float32 arrays [h, w, 3] or None."""
        def tex(t):
            if t is None:
                return None, 0, 0
            a = np.ascontiguousarray(t, dtype=np.float32)
            assert a.ndim == 3 and a.shape[2] == 3
            return _fptr(a), a.shape[1], a.shape[0]
        ap, aw, ah = tex(albedo_texture)
        ep, ew, eh = tex(emissive_texture)
        lib().orc_set_model_material(self._h, int(model_index), int(kind), _f3(base_color), _f3(emissive), ap, aw, ah, ep, ew, eh)

    def set_shading(self, materials=False, point_shadows=False):
        lib().orc_set_shading(self._h, 1 if materials else 0, 1 if point_shadows else 0)

    def set_flat_shading(self, shade, materials, texels):
        """The flattened shading tables (numpy arrays of 32-byte / 64-byte records, float32 texels) for variant B."""
        keep = [np.ascontiguousarray(shade), np.ascontiguousarray(materials),
                np.ascontiguousarray(texels if texels is not None and len(texels) else np.zeros(3, np.float32))]
        self._keep_shading = keep
        lib().orc_set_flat_shading(self._h, *[a.ctypes.data for a in keep])

    def generate(self):
        lib().orc_generate(self._h)

    @property
    def num_tris(self):
        return lib().orc_num_tris(self._h)

    @property
    def num_models(self):
        return lib().orc_num_models(self._h)

    def tri_operands(self, tri_id):
        out = [np.zeros(3, np.float32) for _ in range(4)]
        lib().orc_get_tri_operands(self._h, int(tri_id), *[_fptr(o) for o in out])
        return out

    def all_tri_operands(self):
        n = self.num_tris
        out = np.zeros((n, 4, 3), np.float32)
        for k in range(n):
            a = self.tri_operands(k)
            for j in range(4):
                out[k, j] = a[j]
        return out

    def model_tree(self):
        cap = 2 * max(1, self.num_models)
        boxes = np.zeros((cap, 6), np.float32)
        topo = np.zeros((cap, 2), np.uint32)
        model_of = np.zeros(cap, np.uint32)
        n = lib().orc_get_model_tree(self._h, _fptr(boxes), topo.ctypes.data_as(C.POINTER(C.c_uint32)),
                                     model_of.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
        return boxes[:n], topo[:n], model_of[:n]

    def set_flat(self, model_nodes, nodes, tris, info_bytes):
        """model_nodes / nodes / tris: numpy arrays of 64-byte records; info_bytes: the rtflat_info struct bytes."""
        self._keep = [np.ascontiguousarray(model_nodes), np.ascontiguousarray(nodes), np.ascontiguousarray(tris),
                      np.frombuffer(bytes(info_bytes), dtype=np.uint8).copy()]
        lib().orc_set_flat(self._h, *[a.ctypes.data for a in self._keep])

    def render(self, camera, width, height, num_samples, max_depth=12, window=None, samples=None, workers=0, aov_depths=0,
               flat=False):
        """Returns dict(accumulator[H,W,3], image[H,W,3], tri_id[N,D,H,W], shadow[N,D,H,W], sample_times[N], counters)."""
        prm = RenderParams()
        prm.width, prm.height, prm.numSamples, prm.maxDepth = width, height, num_samples, max_depth
        x0, y0, x1, y1 = window if window else (0, 0, width, height)
        prm.x0, prm.y0, prm.x1, prm.y1 = x0, y0, x1, y1
        s0, s1 = samples if samples else (0, num_samples)
        prm.sampBegin, prm.sampEnd = s0, s1
        prm.numWorkers, prm.aovDepths, prm.useFlat = workers, aov_depths, 1 if flat else 0
        acc = np.zeros((height, width, 3), np.float32)
        img = np.zeros((height, width, 3), np.float32)
        tri = shadow = None
        tp = sp = None
        if aov_depths:
            # planes [0, s1): the oracle only touches those of the passes it renders
            tri = np.zeros((s1, aov_depths, height, width), np.uint32)
            shadow = np.zeros((s1, aov_depths, height, width), np.uint8)
            tp = tri.ctypes.data_as(C.POINTER(C.c_uint32))
            sp = shadow.ctypes.data_as(C.POINTER(C.c_uint8))
        times = np.zeros(num_samples, np.float32)
        cnt = Counters()
        rc = lib().orc_render(self._h, C.byref(camera), C.byref(prm), _fptr(acc), _fptr(img), tp, sp, _fptr(times), C.byref(cnt))
        assert rc == 0
        return dict(accumulator=acc, image=img, tri_id=tri, shadow=shadow, sample_times=times, counters=cnt.as_dict())

    def closest_hit(self, origin, direction, tmin=1e-4, tmax=1e20, flat=False):
        t = C.c_float(0)
        tid = lib().orc_closest_hit(self._h, 1 if flat else 0, _f3(origin), _f3(direction), _f32(tmin), _f32(tmax), C.byref(t))
        return int(tid), float(t.value)

    def any_hit(self, origin, direction, tmin=1e-4, tmax=1e20, flat=False):
        return bool(lib().orc_any_hit(self._h, 1 if flat else 0, _f3(origin), _f3(direction), _f32(tmin), _f32(tmax)))


def rand01(x, y, z):
    out = np.zeros(3, np.float32)
    lib().orc_rand01(int(x), int(y), int(z), _fptr(out))
    return out


def primary_ray(camera, width, height, w, h, samp):
    o, d, r = (np.zeros(3, np.float32) for _ in range(3))
    lib().orc_primary_ray(C.byref(camera), width, height, w, h, samp, _fptr(o), _fptr(d), _fptr(r))
    return o, d, r


def sincos(x):
    s, c = C.c_float(0), C.c_float(0)
    lib().orc_sincos(_f32(x), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def bounce_dir(normal_light, rnd):
    out = np.zeros(3, np.float32)
    lib().orc_bounce_dir(_f3(normal_light), _f3(rnd), _fptr(out))
    return out


def aabb_intersects(lo, hi, o, d, tmin=1e-4, tmax=1e20):
    return bool(lib().orc_aabb_intersects(_f3(lo), _f3(hi), _f3(o), _f3(d), _f32(tmin), _f32(tmax)))
