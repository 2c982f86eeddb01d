"""Host-side tests of the product's scene code (src/scene, src/util, the flattener) against the CPU oracle.
The renderer is created in "hip-offline" mode: no device, nothing rendered by the product here; the oracle walks the
product's flattened arrays (variant B) and must return exactly what its reference-faithful brute-force scan (variant A)
returns.  No GPU needed."""
import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import capi
from tests import scenes

SCENES = {"cube": scenes.cube_scene, "bunny": scenes.bunny_scene, "soup": lambda: scenes.soup_scene(6000, 11, 0.08),
          "soup_lens": lambda: scenes.soup_scene(3000, 12, 0.1, lens=True), "hf": lambda: scenes.heightfield_scene(40, 5),
          "multi": scenes.multi_model_scene, "box_glb": scenes.box_glb_scene, "town_hall": scenes.town_hall_scene}


def offline(spec, **options):
    return spec.build_product("hip-offline", options=options)


@pytest.mark.parametrize("name", ["cube", "soup", "hf", "multi"])
@pytest.mark.parametrize("fix", [False, True])
def test_triangle_records_equal_the_operands_the_oracle_reads(name, fix):
    spec = SCENES[name]()
    r = offline(spec, referenceEdgeIndexing=0 if fix else 1)
    f = r.flat()
    o = spec.build_oracle(fix_edge_offset=fix)
    tris = f["tris"]
    assert f["info"].numTris == o.num_tris == len(tris)
    assert sorted(tris["triId"].tolist()) == list(range(o.num_tris))  # every triangle exactly once
    want = o.all_tri_operands()
    got = np.stack([tris["p0"], tris["e0"], tris["e1"], tris["n"]], 1)
    assert np.array_equal(got.view(np.uint32), want[tris["triId"]].view(np.uint32))


def test_model_tree_topology_matches_the_oracle():
    spec = scenes.multi_model_scene()
    f = offline(spec).flat()
    boxes, topo, model_of = spec.build_oracle().model_tree()
    mn = f["model_nodes"]
    assert len(mn) == len(boxes) == 2 * len(spec.meshes) - 1
    assert np.array_equal(np.concatenate([mn["lo"], mn["hi"]], 1).view(np.uint32), boxes.view(np.uint32))
    assert np.array_equal(np.stack([mn["left"], mn["right"]], 1), topo)  # both are written in pre-order
    leaf = mn["triCount"] > 0
    assert np.array_equal(mn["modelIndex"][leaf], model_of[leaf])
    # order keys: DFS leaf order, ascending
    bases = mn["orderBase"][leaf]
    assert np.array_equal(bases, np.concatenate([[0], np.cumsum(mn["triCount"][leaf])[:-1]]))
    # subtree sizes are consistent
    assert mn["subtreeSize"][0] == len(mn)
    for k in range(len(mn)):
        if not leaf[k]:
            assert mn["subtreeSize"][k] == 1 + mn["subtreeSize"][mn["left"][k]] + mn["subtreeSize"][mn["right"][k]]


@pytest.mark.parametrize("name", ["cube", "soup", "hf", "bunny", "multi"])
def test_bvh_structure_is_sound(name):
    from tests.bvh_checks import assert_bvh_sound
    spec = SCENES[name]()
    assert_bvh_sound(offline(spec).flat())


@pytest.mark.parametrize("name,W,H,N,depth", [("cube", 160, 120, 3, 12), ("bunny", 96, 96, 1, 5), ("soup", 128, 128, 2, 8),
                                             ("soup_lens", 96, 96, 2, 6), ("hf", 128, 96, 2, 6), ("multi", 128, 96, 2, 6), ("box_glb", 128, 96, 2, 6), ("town_hall", 128, 96, 2, 6)])
@pytest.mark.parametrize("fix", [False, True])
def test_flat_walk_equals_reference_scan(name, W, H, N, depth, fix):
    """Variant B (walk of the flattened arrays with the rule the kernels implement) must pick the same triangles, shadow
    flags and radiance bits as variant A (model tree + linear scan, the reference's algorithm)."""
    if fix and name not in ("multi", "box_glb", "town_hall"):
        pytest.skip("the edge-pool indexing only matters with several models")
    spec = SCENES[name]()
    f = offline(spec, referenceEdgeIndexing=0 if fix else 1).flat()
    o = spec.build_oracle(fix_edge_offset=fix)
    o.set_flat(f["model_nodes"], f["nodes"], f["tris"], f["info_bytes"])
    cam = spec.oracle_camera()
    a = o.render(cam, W, H, N, max_depth=depth, aov_depths=min(depth, 4), workers=8)
    b = o.render(cam, W, H, N, max_depth=depth, aov_depths=min(depth, 4), workers=8, flat=True)
    assert a["counters"]["nonFiniteAccepts"] == 0 and b["counters"]["nonFiniteAccepts"] == 0
    assert np.array_equal(a["tri_id"], b["tri_id"]), f"{(a['tri_id'] != b['tri_id']).sum()} hit ids differ"
    assert np.array_equal(a["shadow"], b["shadow"])
    assert np.array_equal(a["accumulator"].view(np.uint32), b["accumulator"].view(np.uint32))
    for k in ("extendPrimary", "extendSecondary", "shadow", "hits"):
        assert a["counters"][k] == b["counters"][k]
    assert b["counters"]["trisTested"] < a["counters"]["trisTested"] or name == "cube"


def test_axis_parallel_rays_do_not_break_the_padded_slab_test():
    """Regression: a direction with an exactly zero component makes 1/d infinite; inf - inf = NaN in the fused slab
    arithmetic used to drop the far plane and cull a box the ray does hit (found on the GPU at cube 256x192, samp 2,
    pixel (116, 44), whose primary direction has d.y == 0)."""
    spec = scenes.cube_scene()
    f = offline(spec).flat()
    o = spec.build_oracle()
    o.set_flat(f["model_nodes"], f["nodes"], f["tris"], f["info_bytes"])
    org = np.array([4.9680495, 0.9936099, 1.9872198], np.float32)
    for d in ([-0.91117615, 0.0, -0.41201708], [-1.0, 0.0, 0.0], [-0.98, -0.19899748, 0.0], [-0.98, 0.0, -0.19899748]):
        d = np.array(d, np.float32)
        assert o.closest_hit(org, d) == o.closest_hit(org, d, flat=True)
        assert o.any_hit(org, d) == o.any_hit(org, d, flat=True)
    rng = np.random.default_rng(9)
    for _ in range(2000):  # random origins on a sphere, directions with one or two zero components
        org = rng.normal(size=3)
        org = (3.0 * org / np.linalg.norm(org)).astype(np.float32)
        d = -org.copy()
        d[rng.integers(0, 3)] = 0.0
        if rng.random() < 0.3:
            d[rng.integers(0, 3)] = 0.0
        if not np.any(d):
            continue
        d = (d / np.linalg.norm(d)).astype(np.float32)
        assert o.closest_hit(org, d) == o.closest_hit(org, d, flat=True)


def test_procedural_meshes_are_deterministic_and_sized_as_documented():
    a, b = ra.Mesh.soup(1000, 0x5EED0001, 0.02), ra.Mesh.soup(1000, 0x5EED0001, 0.02)
    assert np.array_equal(a.arrays()[0], b.arrays()[0]) and a.num_triangles == 1000 and a.num_positions == 3000
    assert not np.array_equal(a.arrays()[0], ra.Mesh.soup(1000, 0x5EED0002, 0.02).arrays()[0])
    p, i = a.arrays()
    e0, e1 = p[i[:, 1]] - p[i[:, 0]], p[i[:, 2]] - p[i[:, 0]]
    assert np.all(np.abs(p[i[:, 0]]) <= 1.0) and np.all(np.linalg.norm(e0, axis=1) <= 0.02 + 1e-6) and np.all(np.linalg.norm(np.cross(e0, e1), axis=1) > 0)
    hf = ra.Mesh.heightfield(10, 3, 0.15)
    assert hf.num_triangles == 200 and hf.num_positions == 121
    assert 2236 * 2236 * 2 == 9999392
    cube = ra.Mesh.cube()
    p, i = cube.arrays()
    assert cube.num_triangles == 12 and cube.num_positions == 24 and np.all(np.abs(p) == 1.0)
    n = np.cross(p[i[:, 1]] - p[i[:, 0]], p[i[:, 2]] - p[i[:, 0]])
    centre = (p[i[:, 0]] + p[i[:, 1]] + p[i[:, 2]]) / 3
    assert np.all((n * centre).sum(1) > 0)  # outward winding


def test_offline_backend_refuses_to_render_and_bad_arguments_are_rejected():
    r = offline(scenes.cube_scene())
    with pytest.raises(ra.RtError) as e:
        r.render(scenes.cube_scene().product_camera(), 32, 32, 1)
    assert e.value.code == capi.RTHIP_E_NODEVICE
    with pytest.raises(ra.RtError):
        r.accumulator()
    with pytest.raises(ra.RtError):
        r.set_option("noSuchOption", 1)
    with pytest.raises(ra.RtError):
        ra.Renderer("cuda")
    with pytest.raises(ra.RtError):
        ra.Mesh.from_arrays([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 3]])  # index out of range
    with pytest.raises(ra.RtError):
        r.generate()  # generate() is one-shot
    assert "no usable HIP device" in capi.error_string(capi.RTHIP_E_NODEVICE)
