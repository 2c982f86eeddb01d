"""(Depth 0 first; whole paths -- bounces, roulette, per-depth lights, sample-ordered accumulation -- further down.)
The oracle pinned by a second, independently written evaluator (tests/independent_ref.py: numpy fp32, taken straight
from the reference's sources): primary hit ids, shadow flags and the depth-cap-1 accumulator must agree BIT FOR BIT on
the shipped cube, the Stanford bunny and a four-model scene (model tree, DFS tie order, the edge-pool indexing of
src/cpu/CPUScene.cpp:157-158) at 64x64.  A shared misreading of glm's operation order would have to be made twice."""
import numpy as np
import pytest

from tests import independent_ref as ind
from tests import scenes


def _models(spec):
    return [(*mesh.arrays(), t) for mesh, t in spec.meshes]


@pytest.mark.parametrize("name,fix_edges", [("cube", False), ("bunny", False), ("multi", False), ("multi", True)])
def test_oracle_agrees_with_the_independent_numpy_evaluator(name, fix_edges):
    spec = {"cube": scenes.cube_scene, "bunny": scenes.bunny_scene, "multi": scenes.multi_model_scene}[name]()
    W, H, N = 64, 64, 2
    want = spec.build_oracle(fix_edge_offset=fix_edges).render(spec.oracle_camera(), W, H, N, max_depth=1, aov_depths=1, workers=4)
    assert want["counters"]["nonFiniteAccepts"] == 0
    scene = ind.Scene(_models(spec), reference_edge_indexing=not fix_edges)
    acc = np.zeros((H, W, 3), np.float32)
    for s in range(N):
        rad, tri, shadow, t = ind.render_first_hit(scene, spec.camera, W, H, N, s, spec.dir_lights, spec.point_lights)
        assert np.array_equal(tri, want["tri_id"][s, 0]), f"pass {s}: {(tri != want['tri_id'][s, 0]).sum()} hit ids differ"
        assert np.array_equal(shadow, want["shadow"][s, 0]), f"pass {s}: {(shadow != want['shadow'][s, 0]).sum()} shadow flags differ"
        acc = acc + rad
    assert (want["tri_id"][:, 0] != 0xFFFFFFFF).sum() > 200, "the camera should see the scene"
    assert np.array_equal(acc.view(np.uint32), want["accumulator"].view(np.uint32)), \
        f"accumulator differs in {(acc != want['accumulator']).any(axis=2).sum()} pixels, max abs {np.abs(acc - want['accumulator']).max()}"


def test_independent_primary_rays_and_rand01_match_the_oracle_bit_for_bit():
    from oracle import binding as ob
    cam = dict(scenes.MAIN_CAMERA)
    cam.update(f=0.7, A=0.35, S_O=20)  # thin lens active (Camera.h defaults)
    W, H = 96, 64
    for samp in (0, 1, 2, 3, 7):
        o, d, rnd = ind.primary_rays(cam, W, H, samp)
        ocam = ob.make_camera(**cam)
        for (w, h) in ((0, 0), (95, 63), (17, 40), (48, 32), (95, 0)):
            oo, dd, rr = ob.primary_ray(ocam, W, H, w, h, samp)
            assert np.array_equal(np.broadcast_to(o, d.shape)[h, w].astype(np.float32).view(np.uint32), oo.view(np.uint32)), (samp, w, h)
            assert np.array_equal(d[h, w].view(np.uint32), dd.view(np.uint32)), (samp, w, h)
            assert np.array_equal(rnd[h, w].view(np.uint32), rr.view(np.uint32)), (samp, w, h)


# ---- whole paths (round 3): bounce direction, roulette, emissive / throughput order, lights at every depth, the depth cap
# and the samp-ordered accumulation exist in the oracle and -- separately written, numpy fp32 straight from
# src/cpu/CPUScene.cpp:6-65 and src/cpu/CPURenderer.cpp:196-197 -- in tests/independent_ref.render_paths.  The numpy side
# takes its sin / cos from float64 libm rounded once to fp32, the oracle from its own sincos_det: nothing transcendental is shared.
def _room_scene():
    """A closed room (a cube seen from inside) around a small cube: no path ever escapes, so every one of them reaches the depth
    cap, passes the roulette depths and folds lights twelve times."""
    meshes = [(scenes.ra.Mesh.cube(), scenes.scale(4.0)), (scenes.ra.Mesh.cube(), scenes.translate(0.4, -1.0, 0.2) @ scenes.scale(0.8))]
    spec = scenes._ps.SceneSpec("room", meshes, dict(position=(2.5, 1.5, 3.0), target=(0, -0.5, 0), S_O=6, f=0, A=0),
                                point_lights=[((-1.0, 2.0, 1.0), (1.0, 0.9, 0.8), 9.0)])
    spec.__class__ = scenes.SceneSpec
    return spec


def _room_with_materials():
    spec = _room_scene()
    spec.meshes[0][0].set_material(scenes.ra.capi.MAT_FACTORS, (0.7, 0.6, 0.8), (0.02, 0.01, 0.0))  # p = 0.8 < 1: roulette really ends paths
    spec.meshes[1][0].set_material(scenes.ra.capi.MAT_FACTORS, (0.9, 0.3, 0.2), (0.5, 0.4, 0.1))
    return spec


_PATH_CASES = {
    # name: (recipe, fix_edges, materials, point_shadows)
    "cube": (scenes.cube_scene, False, False, False),
    "multi": (scenes.multi_model_scene, False, False, False),
    "multi-fixed-edges": (scenes.multi_model_scene, True, False, False),
    "bunny": (scenes.bunny_scene, False, False, False),
    "heightfield-two-lights": (lambda: scenes.heightfield_scene(24), False, False, False),
    "town-hall": (scenes.town_hall_scene, False, False, False),
    "room": (_room_scene, False, False, False),
    "room-fixed-edges": (_room_scene, True, False, False),
    "room-materials-point-shadows": (_room_with_materials, True, True, True),
    "materials": (scenes.materials_scene, False, True, False),
    "materials-point-shadows": (scenes.materials_scene, False, True, True),
}


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", sorted(_PATH_CASES))
def test_oracle_whole_paths_agree_with_the_independent_numpy_evaluator(case):
    recipe, fix_edges, materials, point_shadows = _PATH_CASES[case]
    spec = recipe()
    W, H, N, D = 64, 64, 4, 12
    want = spec.build_oracle(fix_edge_offset=fix_edges, materials=materials, point_shadows=point_shadows).render(
        spec.oracle_camera(), W, H, N, max_depth=D, aov_depths=D, workers=4)
    assert want["counters"]["nonFiniteAccepts"] == 0
    models = [(*mesh.arrays(), t, mesh.texcoords(), mesh.material_spec) for mesh, t in spec.meshes]
    scene = ind.Scene(models, reference_edge_indexing=not fix_edges)
    acc = np.zeros((H, W, 3), np.float32)
    for s in range(N):
        rad, tri, shadow = ind.render_paths(scene, spec.camera, W, H, N, s, spec.dir_lights, spec.point_lights, D,
                                            materials=materials, point_shadows=point_shadows)
        for depth in range(D):
            bad = tri[depth] != want["tri_id"][s, depth]
            assert not bad.any(), f"pass {s} depth {depth}: hit ids differ at pixels {np.argwhere(bad)[:8].tolist()} ({bad.sum()} in all)"
            bad = shadow[depth] != want["shadow"][s, depth]
            assert not bad.any(), f"pass {s} depth {depth}: shadow flags differ at pixels {np.argwhere(bad)[:8].tolist()} ({bad.sum()} in all)"
        acc = acc + rad  # CPURenderer.cpp:197: passes add in samp order
    bad = (acc.view(np.uint32) != want["accumulator"].view(np.uint32)).any(axis=2)
    assert not bad.any(), f"accumulator differs at pixels {np.argwhere(bad)[:8].tolist()} ({bad.sum()} in all, max abs {np.abs(acc - want['accumulator']).max()})"
    if case.startswith("room"):
        deep = (want["tri_id"][:, D - 1] != 0xFFFFFFFF).sum()
        seen = (want["tri_id"][:, 0] != 0xFFFFFFFF).sum()
        # nearly every path reaches the cap (a few leak through the room's edges: bounce rays start ON the surface, tmin 1e-4);
        # with p = 0.8 the roulette ends the fifth of them whose rnd01.z >= 0.8 (the same z at every depth, CPUScene.cpp:21)
        assert seen == N * W * H and (deep > 0.9 * seen if not materials else 0.6 * seen < deep < 0.85 * seen), (seen, deep)


def test_independent_bounce_directions_match_the_oracle_bit_for_bit():
    """CPUScene.cpp:54-60 on 20000 (normal, rnd) pairs, including rnd.x at the ends of [0, 1] and normals with |w.x| around 0.1."""
    from oracle import binding as ob
    rng = np.random.default_rng(5)
    n = rng.normal(size=(20000, 3)).astype(np.float32)
    n = ind.normalize(n)
    n[:200, 0] = np.float32(0.1) + (rng.integers(-3, 4, 200) * np.float32(2.0 ** -27)).astype(np.float32)
    rnd = rng.random((20000, 3), np.float32)
    rnd[:50, 0], rnd[50:100, 0], rnd[100:150, 1], rnd[150:200, 1] = 0.0, 1.0, 0.0, 1.0
    got = ind.bounce_directions(n, rnd)
    differing = []
    for k in range(len(n)):
        want = ob.bounce_dir(n[k], rnd[k])
        if not np.array_equal(np.asarray(want, np.float32).view(np.uint32), got[k].view(np.uint32)):
            differing.append(k)
    # float64 libm rounded once vs sincos_det: any argument where they part would be listed here (none known)
    assert not differing, f"{len(differing)} bounce directions differ, first at {differing[:5]}"
