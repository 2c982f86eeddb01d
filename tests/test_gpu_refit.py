"""rthip_refit (SURVEY.md 8f-2 "LBVH + refit"): a scene whose vertices moved keeps its triangle-level tree -- topology, leaf
contents, record order -- and gets every box recomputed bottom-up.  Frames through the refitted tree must equal the oracle's
brute-force render of the MOVED scene bit for bit (hit results never depend on the tree), through the host-built and the
device-built tree alike; motion that leaves the box the scene was flattened for is refused."""
import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import capi
from tests import scenes
from tests.bvh_checks import assert_bvh_sound

pytestmark = pytest.mark.gpu


def _moved(spec, fn):
    """the same meshes with fn applied to their vertex positions"""
    meshes = []
    for mesh, t in spec.meshes:
        p, i = mesh.arrays()
        meshes.append((ra.Mesh.from_arrays(fn(p), i), t))
    out = scenes._ps.SceneSpec(spec.name + "-moved", meshes, spec.camera, spec.dir_lights, spec.point_lights)
    out.__class__ = scenes.SceneSpec
    return out


def _squash(p):
    q = p.copy()
    q[:, 1] = np.float32(0.6) * p[:, 1] + np.float32(0.02) * np.sin(np.float32(9.0) * p[:, 0]) * np.cos(np.float32(7.0) * p[:, 2])
    q[:, 0] = p[:, 0] * (np.float32(1.0) - np.float32(0.1) * np.abs(p[:, 2]))
    return q.astype(np.float32)


@pytest.mark.parametrize("device_bvh", [0, 1, 2])
@pytest.mark.parametrize("name", ["hf", "soup", "multi"])
def test_refitted_tree_renders_the_moved_scene_bit_for_bit(name, device_bvh):
    spec = {"hf": lambda: scenes.heightfield_scene(96, 9), "soup": lambda: scenes.soup_scene(20000, 3, 0.05), "multi": scenes.multi_model_scene}[name]()
    moved = _moved(spec, _squash if name != "soup" else (lambda p: (p * np.float32(0.9) + np.float32(0.03) * np.sin(np.float32(5.0) * p[:, [1, 2, 0]])).astype(np.float32)))
    W, H, N, depth = 160, 120, 2, 6
    r = spec.build_product(options=dict(maxDepth=depth, aovDepths=2, deviceBvh=device_bvh))
    before = r.flat()
    nodes_before = int(before["info"].numNodes)
    ms = r.refit(moved.build_product("hip-offline"))
    assert ms > 0
    want = moved.build_oracle().render(moved.oracle_camera(), W, H, N, max_depth=depth, aov_depths=2, workers=8)
    r.begin_frame(moved.product_camera(), W, H, N)
    for s in range(N):
        r.render_passes(s, 1)
        for d in range(2):
            tri, sh = r.aov(s, d)
            assert np.array_equal(tri, want["tri_id"][s, d]), (s, d)
            assert np.array_equal(sh, want["shadow"][s, d]), (s, d)
    assert r.counters()["nonFiniteAccepts"] == 0
    assert np.array_equal(r.accumulator().view(np.uint32), want["accumulator"].view(np.uint32))
    assert (want["tri_id"][:, 0] != 0xFFFFFFFF).sum() > 1000 and (want["tri_id"][:, 1] != 0xFFFFFFFF).sum() > 20
    # the tree on the device after the refit: same size, still sound around the moved triangles
    hl = capi.hip_lib()
    nodes = np.zeros(nodes_before, capi.NODE_DTYPE)
    tris = np.zeros(int(before["info"].numTris), capi.TRI_DTYPE)
    rc = hl.rthip_download_scene(r.context, nodes.ctypes.data, len(nodes), tris.ctypes.data, len(tris))
    assert rc == 0
    assert np.array_equal(tris["triId"], before["tris"]["triId"])  # records kept their places
    pool = np.argsort(tris["triId"], kind="stable")
    mf = moved.build_product("hip-offline").flat()
    mpool = np.argsort(mf["tris"]["triId"], kind="stable")
    assert np.array_equal(tris["p0"][pool], mf["tris"]["p0"][mpool]) and np.array_equal(tris["e1"][pool], mf["tris"]["e1"][mpool])
    assert_bvh_sound(dict(before, nodes=nodes, tris=tris))
    r.close()


def test_refit_refuses_motion_out_of_the_flattened_box():
    spec = scenes.heightfield_scene(32, 4)
    r = spec.build_product(options=dict(maxDepth=2))
    far = _moved(spec, lambda p: (p * np.float32(8.0)).astype(np.float32)).build_product("hip-offline")
    with pytest.raises(ra.RtError) as e:
        r.refit(far)
    assert e.value.code == capi.RTHIP_E_BOUNDS
    # the scene is gone until it is uploaded again: a frame cannot begin
    with pytest.raises(ra.RtError):
        r.begin_frame(spec.product_camera(), 64, 64, 1)
    r.close()
