"""Scene recipes of the benchmark workloads and the tests: meshes, lights and camera of each named scene, fed to the
product renderer through its C ABI.  (The tests extend these specs with the CPU oracle's side in tests/scenes.py; nothing
here imports test infrastructure.)"""
import os

import numpy as np

from . import capi as _capi
from .capi import Mesh, Renderer, make_camera

# fixture DATA (arrays of the shipped stanford-bunny.obj and the .glb meshes; /root/reference does not travel to the GPU box)
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main_light_dir():
    """glm::normalize(vec3(-0.4,-0.3,-0.2)) of the reference's main.cpp:10, in fp32 with glm's operation order."""
    d = np.array([-0.4, -0.3, -0.2], np.float32)
    s = np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
    return d * (np.float32(1.0) / np.sqrt(s))


def translate(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m


def scale(s):
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = m[1, 1] = m[2, 2] = s
    return m


class SceneSpec:
    def __init__(self, name, meshes, camera, dir_lights=None, point_lights=None):
        self.name = name
        self.meshes = meshes  # list of (Mesh, 4x4 transform)
        self.camera = camera  # dict(position, target, S_O, f, A)
        self.dir_lights = [(main_light_dir(), (1, 1, 1))] if dir_lights is None else dir_lights
        self.point_lights = [((0, 0, 0), (1, 1, 1), 1.0)] if point_lights is None else point_lights

    def product_camera(self):
        return make_camera(**self.camera)

    def build_product(self, backend="hip", options=None, device=0):
        r = Renderer(backend, device)
        for d, c in self.dir_lights:
            r.add_directional_light(d, c)
        for p, c, a in self.point_lights:
            r.add_point_light(p, c, a)
        for mesh, t in self.meshes:
            r.add_mesh(mesh, t)
        for k, v in (options or {}).items():
            r.set_option(k, v)
        r.generate()
        return r


MAIN_CAMERA = dict(position=(5, 1, 2), target=(0, 0, 0), S_O=6, f=0, A=0)  # main.cpp:18-24
IDENTITY = _capi.IDENTITY


def cube_scene():
    """The reference's own scene (main.cpp:9-24) with the procedural stand-in for cube.fbx."""
    return SceneSpec("cube", [(Mesh.cube(), IDENTITY)], MAIN_CAMERA)


def bunny_mesh():
    z = np.load(os.path.join(GOLDEN, "bunny_mesh.npz"))
    return Mesh.from_arrays(z["positions"], z["indices"])


def bunny_scene():
    return SceneSpec("bunny", [(bunny_mesh(), IDENTITY)],
                     dict(position=(0.05, 0.18, 0.35), target=(-0.02, 0.1, 0.0), S_O=6, f=0, A=0),
                     point_lights=[((0.0, 0.1, 0.1), (1, 1, 1), 1.0)])


def soup_scene(num_tris=20000, seed=0x5EED0001, radius=0.05, lens=False):
    cam = dict(position=(0, 0, 4), target=(0, 0, 0), S_O=4, f=0.05 if lens else 0, A=0.02 if lens else 0)
    return SceneSpec(f"soup{num_tris}", [(Mesh.soup(num_tris, seed, radius), IDENTITY)], cam)


def heightfield_scene(quads=96, seed=0x5EED0002):
    return SceneSpec(f"hf{quads}", [(Mesh.heightfield(quads, seed, 0.15), IDENTITY)],
                     dict(position=(0.3, 1.2, 2.2), target=(0, 0, 0), S_O=6, f=0, A=0),
                     dir_lights=[(main_light_dir(), (1, 1, 1)), (np.array([0.6, -0.64, 0.48], np.float32), (0.5, 0.25, 0.125))])


def multi_model_scene():
    """Three models: exercises the model-level tree, the DFS tie order and the reference's edge-pool indexing."""
    meshes = [(Mesh.cube(), translate(-1.6, 0, 0) @ scale(0.7)),
              (Mesh.soup(300, 7, 0.4), IDENTITY),
              (Mesh.cube(), translate(1.4, 0.3, -0.2) @ scale(0.5)),
              (Mesh.heightfield(8, 3, 0.3), translate(0, -1.2, 0))]
    return SceneSpec("multi", meshes, dict(position=(1.0, 2.0, 6.0), target=(0, 0, 0), S_O=6, f=0, A=0))


def glb_meshes(name):
    """Meshes of the reference's res/models/<name>.glb (fixture: tests/golden/glb_meshes.npz, one per mesh primitive), each
    with its texture coordinates and its primitive's glTF material factors (read only with the scene option "materials")."""
    z = np.load(os.path.join(GOLDEN, "glb_meshes.npz"))
    out = []
    for k in range(int(z[name + "_count"])):
        m = Mesh.from_arrays(z["%s_%d_positions" % (name, k)], z["%s_%d_indices" % (name, k)], z["%s_%d_texcoords" % (name, k)])
        m.set_material(_capi.MAT_FACTORS, z["%s_%d_base_color" % (name, k)], z["%s_%d_emissive" % (name, k)])
        out.append(m)
    return out


def checker_texture(w, h, a=(0.9, 0.2, 0.1), b=(0.1, 0.3, 0.9), cells=4):
    """A w x h RGB checkerboard with a per-texel ramp on top (so neighbouring texels differ and an off-by-one shows)."""
    y, x = np.mgrid[0:h, 0:w]
    on = (((x * cells) // w + (y * cells) // h) & 1).astype(bool)
    t = np.where(on[..., None], np.asarray(a, np.float32), np.asarray(b, np.float32)).astype(np.float32)
    t *= (np.float32(0.5) + np.float32(0.5) * ((x + 3 * y) % 17).astype(np.float32)[..., None] / np.float32(16))
    return np.ascontiguousarray(t, np.float32)


def quad_mesh(size=1.0, uv_scale=1.0, tilt=0.02):
    """A square facing +y, two triangles, texture coordinates covering [0, uv_scale]^2.  It rises by 2 * tilt * size along x:
    an exactly flat model has a zero-thickness box, which the reference's strict slab test (AABB.h:59) never passes."""
    s = float(size)
    r = float(tilt) * s
    p = np.array([[-s, -r, -s], [s, r, -s], [s, r, s], [-s, -r, s]], np.float32)
    t = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * np.float32(uv_scale)
    return Mesh.from_arrays(p, np.array([[0, 2, 1], [0, 3, 2]], np.uint32), t)


def materials_scene():
    """Every material feature at once: a textured floor (albedo + emissive textures, texture coordinates running past 1 so
    the edge clamp is hit), a cube with the reference's stub material (albedo = texture coordinates), a soup with plain
    factors, an emissive panel, a model left at the default material; one directional and two point lights (one of them
    behind the cube as seen from the floor, so its shadow rays matter)."""
    floor = quad_mesh(3.0, 1.25)
    floor.set_material(_capi.MAT_FACTORS, (0.9, 0.8, 0.7), (0.05, 0.1, 0.02), checker_texture(16, 8), checker_texture(5, 7, (1, 1, 0), (0, 1, 1), 3))
    cube = Mesh.cube()
    cube.set_material(_capi.MAT_STUB)
    soup = Mesh.soup(400, 11, 0.35)
    soup.set_material(_capi.MAT_FACTORS, (0.25, 0.5, 0.95), (0, 0, 0))
    panel = quad_mesh(0.6)
    panel.set_material(_capi.MAT_FACTORS, (0.2, 0.2, 0.2), (2.0, 1.5, 0.5))
    plain = Mesh.cube()
    meshes = [(floor, translate(0, -1.0, 0)), (cube, translate(-0.9, -0.3, 0.2) @ scale(0.6)), (soup, translate(0.8, 0.2, -0.3)),
              (panel, translate(0.0, 1.6, 0.0)), (plain, translate(1.7, -0.6, 0.9) @ scale(0.35))]
    return SceneSpec("materials", meshes, dict(position=(2.5, 2.2, 4.5), target=(0, -0.2, 0), S_O=6, f=0, A=0),
                     point_lights=[((-0.9, 0.9, 0.2), (1.0, 0.9, 0.8), 4.0), ((1.5, 0.4, 1.5), (0.3, 0.6, 1.0), 3.0)])


def box_glb_scene():
    """res/models/box.glb as ModelLoader::loadModel + addModel(identity) would set it up: a cube inside five wall quads at
    +-10, every mesh a model of its own.  Two walls are exactly flat (invisible to the reference's strict slab test), three
    are 2e-6 thick."""
    return SceneSpec("box_glb", [(m, IDENTITY) for m in glb_meshes("box")], MAIN_CAMERA)


def town_hall_scene():
    """res/models/town_hall.glb (two meshes, 58 triangles)."""
    return SceneSpec("town_hall", [(m, IDENTITY) for m in glb_meshes("town_hall")],
                     dict(position=(16.0, 7.0, 15.0), target=(2.5, 2.0, 0.0), S_O=6, f=0, A=0),
                     point_lights=[((2.0, 2.0, 0.0), (1.0, 0.8, 0.6), 12.0)])


# ---- BASELINE.json configs: the scenes and frame parameters bench.py and the full-size GPU tests share
SOUP1M = dict(num_tris=1 << 20, seed=0x5EED0001, radius=0.02)  # configs[2]
HF10M = dict(quads=2236, seed=0x5EED0002)                      # configs[3] / [4]: 2 * 2236^2 = 9,999,392 triangles


def soup1m_scene():
    return soup_scene(**SOUP1M)


def hf10m_scene():
    return heightfield_scene(**HF10M)
