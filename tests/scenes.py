"""Scene recipes shared by the tests: every recipe feeds the SAME arrays, lights and camera to the product renderer
(raytracer_amd.scenes, through the C ABI) and to the CPU oracle (oracle/, test infrastructure).  The recipes themselves live
in the package; this module only adds the oracle's side to each spec."""
import functools

import raytracer_amd as ra  # noqa: F401  (tests reach ra.* through this module's importers)
from oracle import binding as ob
from raytracer_amd import scenes as _ps
from raytracer_amd.scenes import GOLDEN, MAIN_CAMERA, main_light_dir, scale, translate  # noqa: F401


class SceneSpec(_ps.SceneSpec):
    def oracle_camera(self):
        return ob.make_camera(**self.camera)

    def build_oracle(self, fix_edge_offset=False, materials=False, point_shadows=False):
        s = ob.OracleScene(fix_edge_offset=fix_edge_offset)
        for d, c in self.dir_lights:
            s.add_directional_light(d, c)
        for p, c, a in self.point_lights:
            s.add_point_light(p, c, a)
        for k, (mesh, t) in enumerate(self.meshes):
            pos, idx = mesh.arrays()
            s.add_model(pos, idx, t, texcoords=mesh.texcoords())
            if mesh.material_spec is not None:
                s.set_model_material(k, **mesh.material_spec)
        s.set_shading(materials, point_shadows)
        s.generate()
        return s


def _with_oracle(recipe):
    @functools.wraps(recipe)
    def make(*args, **kwargs):
        spec = recipe(*args, **kwargs)
        spec.__class__ = SceneSpec
        return spec
    return make


cube_scene = _with_oracle(_ps.cube_scene)
bunny_scene = _with_oracle(_ps.bunny_scene)
soup_scene = _with_oracle(_ps.soup_scene)
heightfield_scene = _with_oracle(_ps.heightfield_scene)
multi_model_scene = _with_oracle(_ps.multi_model_scene)
box_glb_scene = _with_oracle(_ps.box_glb_scene)
town_hall_scene = _with_oracle(_ps.town_hall_scene)
soup1m_scene = _with_oracle(_ps.soup1m_scene)
hf10m_scene = _with_oracle(_ps.hf10m_scene)
materials_scene = _with_oracle(_ps.materials_scene)
bunny_mesh = _ps.bunny_mesh
glb_meshes = _ps.glb_meshes
