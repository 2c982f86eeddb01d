"""The oracle pinned by a second, independently written evaluator (tests/independent_ref.py: numpy fp32, taken straight
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
