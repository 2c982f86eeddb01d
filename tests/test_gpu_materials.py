"""GPU parity for materials on the hot path and shadowed point lights: the HIP path, through the C ABI,
against the CPU oracle on the same scenes -- hit ids, shadow masks (directional bits, then point-light bits), ray counts and
the accumulator, bit for bit.  What the oracle itself rests on for this feature is stated in tests/test_materials_cpu.py."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import capi
from tests import scenes

pytestmark = pytest.mark.gpu

SCENES = {"materials": scenes.materials_scene, "box_glb": scenes.box_glb_scene, "town_hall": scenes.town_hall_scene,
          "multi": scenes.multi_model_scene}


def _compare(spec, W, H, N, depth, materials, point_shadows, extra=None, flat_oracle=False, passes_per_call=1):
    aov = min(depth, 3)
    opts = dict(maxDepth=depth, aovDepths=aov)
    if materials:
        opts["materials"] = 1
    if point_shadows:
        opts["pointShadows"] = 1
    opts.update(extra or {})
    r = spec.build_product(options=opts)
    oracle = spec.build_oracle(materials=materials, point_shadows=point_shadows)
    if flat_oracle:
        flat = r.flat()
        oracle.set_flat(flat["model_nodes"], flat["nodes"], flat["tris"], flat["info_bytes"])
        if materials:
            sh = r.flat_shading()
            oracle.set_flat_shading(sh["shade"], sh["materials"], sh["texels"])
    want = oracle.render(spec.oracle_camera(), W, H, N, max_depth=depth, aov_depths=aov, workers=14, flat=flat_oracle)
    assert want["counters"]["nonFiniteAccepts"] == 0
    r.begin_frame(spec.product_camera(), W, H, N)
    for s0 in range(0, N, passes_per_call):
        n = min(passes_per_call, N - s0)
        r.render_passes(s0, n)
        for s in range(s0, s0 + n):
            for d in range(aov):
                tri, sh = r.aov(s, d)
                assert np.array_equal(tri, want["tri_id"][s, d]), (s, d, int((tri != want["tri_id"][s, d]).sum()))
                assert np.array_equal(sh, want["shadow"][s, d]), (s, d, int((sh != want["shadow"][s, d]).sum()))
    cnt, oc = r.counters(), want["counters"]
    assert cnt["nonFiniteAccepts"] == 0
    assert cnt["extendPrimary"] == oc["extendPrimary"] and cnt["shadow"] == oc["shadow"] and cnt["hits"][:depth] == oc["hits"][:depth]
    acc = r.accumulator()
    assert np.array_equal(acc.view(np.uint32), want["accumulator"].view(np.uint32)), \
        f"accumulator differs: max abs {np.abs(acc - want['accumulator']).max()}, pixels {(acc != want['accumulator']).any(axis=2).sum()}"
    r.close()
    return want


@pytest.mark.parametrize("name,depth,materials,point_shadows", [
    ("materials", 8, True, True), ("materials", 8, True, False), ("materials", 8, False, True), ("materials", 1, True, True),
    ("materials", 2, True, True), ("box_glb", 8, True, True), ("town_hall", 8, True, True), ("multi", 7, True, True)])
def test_materials_and_point_shadows_match_oracle_bitexact(name, depth, materials, point_shadows):
    want = _compare(SCENES[name](), 176, 128, 3, depth, materials, point_shadows)
    if point_shadows and name == "materials":
        hit = want["shadow"][:, 0] != 0xFF
        assert (want["shadow"][:, 0][hit] >> 1).any(), "the scene should have occluded point-light shadow rays"


@pytest.mark.parametrize("name,depth,materials", [("materials", 8, True), ("materials", 1, False), ("box_glb", 5, True), ("multi", 4, True)])
def test_point_shadow_packets_match_oracle_bitexact(name, depth, materials):
    """The packet kernel's point-light passes (k_trace_packet<SHADOW_PT>: the first shadow launch when packet traversal is in
    use) forced on, with waves of 64 different pixels -- the least coherent input it can get."""
    _compare(SCENES[name](), 176, 128, 3, depth, materials, True, extra=dict(refillMode=2))


def test_points_only_shadowed_by_packets():
    base = scenes.materials_scene()
    spec = scenes.SceneSpec("points_only", base.meshes, base.camera, dir_lights=[], point_lights=base.point_lights)
    _compare(spec, 160, 120, 2, 6, True, True, extra=dict(refillMode=2))


def test_points_only_shadowed():
    """No directional light: every shadow ray is a point-light ray (the SHADOW_PT launch starts at a point light)."""
    base = scenes.materials_scene()
    spec = scenes.SceneSpec("points_only", base.meshes, base.camera, dir_lights=[], point_lights=base.point_lights)
    _compare(spec, 160, 120, 2, 6, True, True)


def test_batched_unrecorded_frames_and_device_bvh():
    """The same through the paths the AOV runs do not take: several passes per call, two slots in flight, 64-sample batches
    (one pixel per micro-block), and a tree built on the device (the shading table is indexed by triId, not by record)."""
    spec = scenes.materials_scene()
    W, H, N, depth = 200, 144, 8, 8
    want = spec.build_oracle(materials=True, point_shadows=True).render(spec.oracle_camera(), W, H, N, max_depth=depth, workers=14)
    for extra in (dict(), dict(deviceBvh=1), dict(pipelining=0)):
        r = spec.build_product(options=dict(maxDepth=depth, materials=1, pointShadows=1, **extra))
        r.begin_frame(spec.product_camera(), W, H, N)
        r.enqueue_passes(0, 3)
        r.enqueue_passes(3, 5)
        acc = r.accumulator()
        assert np.array_equal(acc.view(np.uint32), want["accumulator"].view(np.uint32)), extra
        # a second frame on the same renderer (slots and parity flip)
        r.begin_frame(spec.product_camera(), W, H, N)
        r.render_passes(0, N)
        assert np.array_equal(r.accumulator().view(np.uint32), want["accumulator"].view(np.uint32)), extra
        r.close()


def test_textured_soup_matches_flat_oracle():
    """20 000 triangles with a texture and an emissive texture, against the oracle's walk of the same flattened arrays."""
    spec = scenes.soup_scene(20000, radius=0.05)
    mesh = spec.meshes[0][0]
    pos, _ = mesh.arrays()
    uv = np.ascontiguousarray((pos[:, :2] * np.float32(0.37) + np.float32(0.5)), np.float32)  # some coordinates fall outside [0,1)
    p, i = mesh.arrays()
    textured = ra.Mesh.from_arrays(p, i, uv)
    textured.set_material(capi.MAT_FACTORS, (0.9, 0.9, 0.9), (0.3, 0.2, 0.1), ra.scenes.checker_texture(32, 16), ra.scenes.checker_texture(4, 4))
    spec.meshes = [(textured, spec.meshes[0][1])]
    spec.point_lights = [((0.2, 0.3, 0.5), (1, 1, 1), 2.5)]
    _compare(spec, 192, 192, 2, 6, True, True, flat_oracle=True)


def test_default_material_renders_the_reference_constants():
    spec = scenes.multi_model_scene()
    accs = []
    for opts in (dict(), dict(materials=1)):
        r = spec.build_product(options=dict(maxDepth=6, **opts))
        r.begin_frame(spec.product_camera(), 160, 120, 2)
        r.render_passes(0, 2)
        accs.append(r.accumulator())
        r.close()
    assert np.array_equal(accs[0].view(np.uint32), accs[1].view(np.uint32))


def test_argument_checks():
    spec = scenes.cube_scene()
    r = spec.build_product()
    L = capi.hip_lib()
    shade = np.zeros(12, capi.SHADE_DTYPE)
    mats = np.zeros(1, capi.MATERIAL_DTYPE)
    mats["kind"] = capi.MAT_FACTORS
    ctx = r.context
    assert L.rthip_upload_materials(ctx, shade.ctypes.data, 11, mats.ctypes.data, 1, None, 0) == capi.RTHIP_E_ARG  # one record per triangle
    shade["material"][5] = 1
    assert L.rthip_upload_materials(ctx, shade.ctypes.data, 12, mats.ctypes.data, 1, None, 0) == capi.RTHIP_E_ARG  # material out of range
    shade["material"][5] = 0
    mats["albedoW"], mats["albedoH"] = 4, 4
    texels = np.zeros((15, 3), np.float32)
    assert L.rthip_upload_materials(ctx, shade.ctypes.data, 12, mats.ctypes.data, 1, texels.ctypes.data, 15) == capi.RTHIP_E_ARG  # texture past the pool
    texels = np.zeros((16, 3), np.float32)
    assert L.rthip_upload_materials(ctx, shade.ctypes.data, 12, mats.ctypes.data, 1, texels.ctypes.data, 16) == 0
    r.close()
    # more than 32 shadowed lights do not fit the occlusion mask
    many = scenes.SceneSpec("many_lights", spec.meshes, spec.camera, dir_lights=[(scenes.main_light_dir(), (1, 1, 1))] * 20,
                            point_lights=[((0, 3, 0), (1, 1, 1), 5.0)] * 13)
    many.build_product().close()  # fine while the point lights cast no shadow rays
    with pytest.raises(capi.RtError) as e:
        many.build_product(options=dict(pointShadows=1))
    assert e.value.code == capi.RTHIP_E_LIMIT
    too_many = scenes.SceneSpec("too_many_lights", spec.meshes, spec.camera, dir_lights=[(scenes.main_light_dir(), (1, 1, 1))] * 33, point_lights=[])
    with pytest.raises(capi.RtError) as e:
        too_many.build_product()
    assert e.value.code == capi.RTHIP_E_LIMIT


def _many_lights(n_dir, n_point):
    rng = np.random.default_rng(5)
    dirs = rng.normal(size=(n_dir, 3)).astype(np.float32)
    dirs[:, 1] = -np.abs(dirs[:, 1]) - np.float32(0.2)  # mostly from above
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    dir_lights = [(dirs[k], tuple(rng.random(3).astype(np.float32) * np.float32(0.2))) for k in range(n_dir)]
    point_lights = [(tuple(rng.uniform(-2, 2, 3).astype(np.float32)), tuple(rng.random(3).astype(np.float32) * np.float32(0.3)), float(rng.uniform(2, 5)))
                    for _ in range(n_point)]
    return dir_lights, point_lights


@pytest.mark.parametrize("n_dir,n_point,point_shadows", [(12, 40, False), (9, 14, True), (0, 32, True)])
def test_many_lights_match_oracle_bitexact(n_dir, n_point, point_shadows):
    """More lights than the 8 + 8 of round 1: up to 32 shadow-casting lights (32 occlusion bits per path), any number of
    unshadowed point lights (the reference's vectors are unbounded, Scene.h:53-54); the shadow AOV carries the low 8 bits."""
    base = scenes.multi_model_scene()
    dir_lights, point_lights = _many_lights(n_dir, n_point)
    spec = scenes.SceneSpec("lights_%d_%d" % (n_dir, n_point), base.meshes, base.camera, dir_lights=dir_lights, point_lights=point_lights)
    _compare(spec, 144, 96, 2, 4, False, point_shadows)


def test_a_camera_beyond_the_origin_bound_rebuilds_instead_of_failing():
    """The BVH padding is sized for cameras within 4 scene diagonals of the scene (RTHIP_E_BOUNDS from rthip_begin beyond);
    the reference has no such limit, so the renderer re-flattens for the camera it is given and carries on."""
    base = scenes.multi_model_scene()
    cam = dict(base.camera)
    cam["position"] = tuple(np.float32(60.0) * np.asarray(base.camera["position"], np.float32) / np.float32(6.4))  # ~60 units away: > 4 diagonals
    cam["sensor"] = (0.0036, 0.0024)  # a long lens, so the scene still covers pixels
    spec = scenes.SceneSpec("far_camera", base.meshes, cam, dir_lights=base.dir_lights, point_lights=base.point_lights)
    want = _compare(spec, 160, 120, 2, 5, False, False)
    assert (want["tri_id"][:, 0] != 0xFFFFFFFF).sum() > 2000
