"""Known-answer tests that pin the CPU oracle to the reference's formulas (the reference has no tests of
its own, so these hand-derived answers plus the committed golden vectors are the pin).  No GPU needed."""
import os

import numpy as np
import pytest

from oracle import binding as ob
from tests import scenes

GOLDEN = scenes.GOLDEN


def rand01_python(x, y, z):
    """src/cpu/CPURenderer.cpp:111-116 in plain integer arithmetic."""
    M = 0xFFFFFFFF
    for _ in range(3):
        x, y, z = (((x >> 8) ^ y) * 1103515245) & M, (((y >> 8) ^ z) * 1103515245) & M, (((z >> 8) ^ x) * 1103515245) & M
    # vec3(x) * (1.0f / float(0xffffffffU)): float(0xffffffff) rounds to 2^32, so the factor is exactly 2^-32
    return np.array([np.float32(v) * np.float32(2.0 ** -32) for v in (x, y, z)], np.float32)


@pytest.mark.parametrize("xyz", [(0, 0, 0), (1, 2, 3), (1919, 1079, 15), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (123456789, 987654321, 42)])
def test_rand01_matches_integer_restatement(xyz):
    assert np.array_equal(ob.rand01(*xyz).view(np.uint32), rand01_python(*xyz).view(np.uint32))


def test_rand01_zero_and_range():
    assert np.array_equal(ob.rand01(0, 0, 0), np.zeros(3, np.float32))  # all-zero state stays zero
    rng = np.random.default_rng(0)
    v = np.array([ob.rand01(*rng.integers(0, 2 ** 32, 3)) for _ in range(2000)])
    assert v.min() >= 0.0 and v.max() <= 1.0  # 1.0 is reachable (float(x) can round up to 2^32)


def test_golden_rand01_and_primary_rays():
    g = np.load(os.path.join(GOLDEN, "rand01.npz"))
    got = np.array([ob.rand01(*r) for r in g["xyz"]])
    assert np.array_equal(got.view(np.uint32), g["out"].view(np.uint32))
    g = np.load(os.path.join(GOLDEN, "primary_rays_512.npz"))
    cam = ob.make_camera(**scenes.MAIN_CAMERA)
    for k, r in enumerate(g["whs"]):
        o, d, rnd = ob.primary_ray(cam, 512, 512, *[int(v) for v in r])
        assert np.array_equal(o.view(np.uint32), g["origin"][k].view(np.uint32))
        assert np.array_equal(d.view(np.uint32), g["direction"][k].view(np.uint32))
        assert np.array_equal(rnd.view(np.uint32), g["rnd"][k].view(np.uint32))


def test_primary_ray_geometry_float64():
    """Independent float64 derivation of CPURenderer.cpp:158-192 for the main.cpp camera (f = 0, A = 0: pinhole through
    the lens centre lc = pos + 0.035 dir; the ray leaves lc pointing away from the sensor sample)."""
    cam = ob.make_camera(**scenes.MAIN_CAMERA)
    pos = np.array([5.0, 1.0, 2.0])
    dirn = -pos / np.linalg.norm(pos)
    cx = np.cross(dirn, [0, 1, 0])
    cx /= np.linalg.norm(cx)
    cy = np.cross(cx, dirn)
    W = H = 512
    for (w, h, s) in [(256, 256, 0), (0, 0, 1), (511, 300, 2), (17, 450, 3)]:
        o, d, rnd = ob.primary_ray(cam, W, H, w, h, s)
        r2 = 2.0 * rnd[:2].astype(np.float64)
        tent = np.where(r2 < 1, np.sqrt(r2) - 1, 1 - np.sqrt(2 - r2))
        stratum = np.array([(s // 2) % 2, s % 2], float)
        sxy = ((np.array([w, h], float) + 0.5 * (0.5 + stratum + tent)) / np.array([W, H], float) - 0.5) * np.array([0.036, 0.024])
        spos = pos + cx * sxy[0] + cy * sxy[1]
        lc = pos + dirn * 0.035
        want = (lc - spos) / np.linalg.norm(lc - spos)
        np.testing.assert_allclose(o, lc, rtol=0, atol=2e-6)
        np.testing.assert_allclose(d, want, rtol=0, atol=3e-5)  # lc - spos cancels ~2 digits in fp32


def test_centre_pixel_anchor_of_the_reference_scene():
    """the centre ray from (5,1,2) toward the origin hits the cube's x = +1 face at about (1, 0.2, 0.4),
    t ~ 4.35 from the lens centre; the shadow ray escapes; depth-0 radiance = (0,1,0) * n.l + 0.03 with
    n.l = 0.4 / sqrt(0.29); the point light contributes 0 (hit farther than its attenuation 1); the bounce leaves the
    convex cube and misses."""
    spec = scenes.cube_scene()
    o = spec.build_oracle()
    cam = spec.oracle_camera()
    org, d, _ = ob.primary_ray(cam, 512, 512, 256, 256, 0)
    tri, t = o.closest_hit(org, d)
    p0, e0, e1, n = o.tri_operands(tri)
    assert np.allclose(n, [1, 0, 0]) and p0[0] == 1.0
    hit = org + d * np.float32(t)
    assert abs(hit[0] - 1.0) < 1e-5 and abs(hit[1] - 0.2) < 0.02 and abs(hit[2] - 0.4) < 0.02 and abs(t - 4.35) < 0.02
    assert not o.any_hit(hit, -scenes.main_light_dir())
    r = o.render(cam, 512, 512, 1, max_depth=12, window=(256, 256, 257, 257), workers=1, aov_depths=2)
    want = np.array([0.03, 0.4 / np.sqrt(0.29) + 0.03, 0.03])
    np.testing.assert_allclose(r["accumulator"][256, 256], want, rtol=2e-6)
    assert r["tri_id"][0, 0, 256, 256] == tri and r["shadow"][0, 0, 256, 256] == 0
    assert r["tri_id"][0, 1, 256, 256] == 0xFFFFFFFF  # the bounce ray misses
    c = r["counters"]
    assert (c["extendPrimary"], c["extendSecondary"], c["shadow"]) == (1, 1, 1)


def test_self_shadowed_faces_receive_nothing():
    """CPUScene.cpp:34-37: BRDF::evaluate (and with it the +0.03 ambient term of BRDF.cpp:12) is only added when the shadow
    ray escapes.  Seen from the side opposite the light every visible cube face is shadowed by the cube itself, and the
    point light's illuminance max(1 - d/1, 0) is 0 that far from the origin -- so those pixels stay exactly 0."""
    spec = scenes.cube_scene()
    spec.camera = dict(position=(-5, -1, -2), target=(0, 0, 0), S_O=6, f=0, A=0)
    o = spec.build_oracle()
    r = o.render(spec.oracle_camera(), 64, 64, 1, max_depth=12, aov_depths=1, workers=2)
    hit = r["tri_id"][0, 0] != 0xFFFFFFFF
    assert hit.sum() > 500
    assert np.all(r["shadow"][0, 0][hit] == 1)
    assert np.all(r["accumulator"] == 0)  # shadowed hits add nothing; misses add nothing either (no sky, CPUScene.cpp:10)


def test_aabb_intersects_semantics():
    """AABB.h:47-60: strict tmin < tmax, so a box with no thickness is never hit -- and neither are flat models."""
    assert ob.aabb_intersects((-1, -1, -1), (1, 1, 1), (5, 0, 0), (-1, 0.01, 0.01))
    assert not ob.aabb_intersects((-1, -1, -1), (1, 1, 1), (5, 0, 0), (1, 0.01, 0.01))   # behind the origin
    assert not ob.aabb_intersects((-1, 0, -1), (1, 0, 1), (0, 5, 0), (0.01, -1, 0.01))   # flat in y: tmin == tmax
    assert not ob.aabb_intersects((-1, -1, -1), (1, 1, 1), (5, 0, 0), (-1, 0.01, 0.01), tmin=1e-4, tmax=3.9)  # tmax before entry
    # axis-parallel ray inside the slab: 1/0 = inf, (lo-o)*inf = -inf, (hi-o)*inf = +inf -> passes on that axis
    assert ob.aabb_intersects((-1, -1, -1), (1, 1, 1), (5, 0.5, 0.5), (-1, 0, 0))
    # on the slab boundary 0*inf = NaN; the comparison forms of min/max then drop the NaN differently per position
    assert ob.aabb_intersects((-1, -1, -1), (1, 1, 1), (5, 1.0, 0.5), (-1, 0, 0)) in (True, False)


def test_flat_model_is_invisible_like_in_the_reference():
    import raytracer_amd as ra
    quad = ra.Mesh.from_arrays([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], [[0, 2, 1], [0, 3, 2]])
    spec = scenes.SceneSpec("quad", [(quad, ra.capi.IDENTITY)], dict(position=(0, 3, 0.1), target=(0, 0, 0), S_O=6, f=0, A=0))
    r = spec.build_oracle().render(spec.oracle_camera(), 32, 32, 1, aov_depths=1, workers=1)
    assert np.all(r["tri_id"] == 0xFFFFFFFF)


def test_tie_rule_later_triangle_and_right_subtree_win():
    """CPUScene.cpp:231 (`best.t < t` skips, so equal t overwrites) and :136 (right result unless left is strictly nearer)."""
    import raytracer_amd as ra
    tri = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    # (a third small triangle off to the side gives the model box some thickness: a flat box is never hit, AABB.h:59)
    far = np.array([[0.8, 0.8, -0.5], [0.9, 0.8, -0.5], [0.8, 0.9, -0.4]], np.float32)
    two = ra.Mesh.from_arrays(np.concatenate([tri, tri, far]), [[0, 1, 2], [3, 4, 5], [6, 7, 8]])
    spec = scenes.SceneSpec("dup", [(two, ra.capi.IDENTITY)], dict(position=(0, 0, 3), target=(0, 0, 0), S_O=6, f=0, A=0))
    o = spec.build_oracle()
    assert o.closest_hit((0, 0, 3), (0.01, 0.02, -1))[0] == 1
    # two coincident one-triangle models (given thickness by a far vertex so that their boxes can be hit at all)
    one = lambda: ra.Mesh.from_arrays(np.concatenate([tri, far]), [[0, 1, 2], [3, 4, 5]])  # noqa: E731
    spec = scenes.SceneSpec("dup2", [(one(), ra.capi.IDENTITY), (one(), ra.capi.IDENTITY)],
                            dict(position=(0, 0, 3), target=(0, 0, 0), S_O=6, f=0, A=0))
    o = spec.build_oracle(fix_edge_offset=True)
    assert o.closest_hit((0, 0, 3), (0.01, 0.02, -1))[0] == 2  # triangle 0 of the second model, the right child of the root
    # with the reference's own edge-pool indexing (CPUScene.cpp:205-206 reads indicesOffset + 2i, but Scene.cpp:39-40 stored
    # the pair at 2*(indicesOffset + i)) the second model's triangle 0 is tested with the edges of the FIRST model's
    # triangle 1, misses, and the first model's copy wins instead
    assert spec.build_oracle(fix_edge_offset=False).closest_hit((0, 0, 3), (0.01, 0.02, -1))[0] == 0


def test_degenerate_triangle_poisons_the_scan_with_nan_like_the_reference():
    """CPUScene.cpp:214-231: a zero-area triangle gives fraction = 1/0 and 0*inf = NaN; NaN passes every `<`/`>` rejection, is
    stored as t, and `t < FLT_MAX` (:10) then reads as a miss -- unless a later accepted triangle overwrites it.  This is
    the behaviour the parity domain excludes (both sides count it as nonFiniteAccepts)."""
    import raytracer_amd as ra
    tri = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    point = np.array([[0, 0, -0.5]] * 3, np.float32)
    m = ra.Mesh.from_arrays(np.concatenate([tri, point]), [[0, 1, 2], [3, 4, 5]])
    spec = scenes.SceneSpec("nan", [(m, ra.capi.IDENTITY)], dict(position=(0, 0, 3), target=(0, 0, 0), S_O=6, f=0, A=0))
    o = spec.build_oracle()
    assert o.closest_hit((0, 0, 3), (0.01, 0.02, -1))[0] == 0xFFFFFFFF  # the valid hit on triangle 0 is lost
    m2 = ra.Mesh.from_arrays(np.concatenate([point, tri]), [[0, 1, 2], [3, 4, 5]])
    spec2 = scenes.SceneSpec("nan2", [(m2, ra.capi.IDENTITY)], dict(position=(0, 0, 3), target=(0, 0, 0), S_O=6, f=0, A=0))
    assert spec2.build_oracle().closest_hit((0, 0, 3), (0.01, 0.02, -1))[0] == 1   # later triangle overwrites the NaN
    r = o.render(spec.oracle_camera(), 16, 16, 1, workers=1)
    assert r["counters"]["nonFiniteAccepts"] > 0


def test_sincos_det_against_libm_and_float64():
    x = np.linspace(0, 2 * np.pi, 20001, dtype=np.float32)
    got = np.array([ob.sincos(v) for v in x], np.float32)
    ref_s = np.sin(x.astype(np.float64)).astype(np.float32)
    ref_c = np.cos(x.astype(np.float64)).astype(np.float32)
    # essentially the correctly rounded float of the true value
    assert (got[:, 0] != ref_s).mean() < 1e-3 and (got[:, 1] != ref_c).mean() < 1e-3
    libm_s, libm_c = np.sin(x), np.cos(x)  # numpy's float32 routines == platform libm quality
    ulp = np.spacing(np.maximum(np.abs(libm_s), np.float32(1e-30)))
    assert np.all(np.abs(got[:, 0] - libm_s) <= 2 * ulp)
    ulp = np.spacing(np.maximum(np.abs(libm_c), np.float32(1e-30)))
    assert np.all(np.abs(got[:, 1] - libm_c) <= 2 * ulp)
    s0, c0 = ob.sincos(0.0)
    assert s0 == 0.0 and c0 == 1.0


def test_bounce_direction_is_unit_and_in_the_hemisphere():
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        r = rng.random(3).astype(np.float32)
        d = ob.bounce_dir(n, r)
        assert abs(np.linalg.norm(d.astype(np.float64)) - 1) < 1e-6
        assert np.dot(d.astype(np.float64), n.astype(np.float64)) >= -1e-6
    # float64 restatement of CPUScene.cpp:54-60 for one case with |w.x| > 0.1
    n, r = np.array([0.6, 0.0, 0.8], np.float32), np.array([0.25, 0.5, 0.1], np.float32)
    w = n.astype(np.float64)
    u = np.cross([0, 1, 0], w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    r1, r2 = 2 * np.pi * 0.25, 0.5
    want = u * np.cos(r1) * np.sqrt(r2) + v * np.sin(r1) * np.sqrt(r2) + w * np.sqrt(1 - r2)
    np.testing.assert_allclose(ob.bounce_dir(n, r), want / np.linalg.norm(want), atol=1e-6)


def test_depth_cap_and_worker_count_do_not_change_a_convex_scene():
    """The cube is convex: every bounce ray leaves it, so depth caps 1 and 12 accumulate the same radiance; the result
    does not depend on how many ThreadPool workers ran the column tasks."""
    spec = scenes.cube_scene()
    o = spec.build_oracle()
    a = o.render(spec.oracle_camera(), 96, 64, 3, max_depth=12, workers=1)
    b = o.render(spec.oracle_camera(), 96, 64, 3, max_depth=1, workers=7)
    assert np.array_equal(a["accumulator"].view(np.uint32), b["accumulator"].view(np.uint32))
    assert a["counters"]["extendSecondary"] == a["counters"]["hits"][0]  # one bounce ray per primary hit, all miss
    assert b["counters"]["extendSecondary"] == b["counters"]["hits"][0]  # traced and discarded at the cap (CPUScene.cpp:8 before :15)


def test_accumulate_and_tonemap_formula():
    """CPURenderer.cpp:196-198: accumulator += radiance / N in sample order; image = pow(max(acc * N/(samp+1), 0), 0.45)."""
    spec = scenes.cube_scene()
    o = spec.build_oracle()
    N = 4
    full = o.render(spec.oracle_camera(), 48, 32, N, workers=2)
    singles = [o.render(spec.oracle_camera(), 48, 32, N, samples=(s, s + 1), workers=2)["accumulator"] for s in range(N)]
    # orc_render zeroes only when sampBegin == 0, so singles[s>0] continue from an empty numpy buffer: each is radiance_s / N
    acc = np.zeros_like(full["accumulator"])
    for s in range(N):
        acc = acc + singles[s]
    assert np.array_equal(acc.view(np.uint32), full["accumulator"].view(np.uint32))
    np.testing.assert_allclose(full["image"], np.power(np.maximum(full["accumulator"] * np.float32(N / N), 0), np.float32(0.45)), rtol=1e-6)


@pytest.mark.parametrize("name,n", [("cube_64x64_1spp.npz", 1), ("cube_64x64_4spp.npz", 4)])
def test_golden_cube(name, n):
    g = np.load(os.path.join(GOLDEN, name))
    spec = scenes.cube_scene()
    r = spec.build_oracle().render(spec.oracle_camera(), 64, 64, n, max_depth=12, aov_depths=2, workers=3)
    assert np.array_equal(r["tri_id"], g["tri_id"]) and np.array_equal(r["shadow"], g["shadow"])
    assert np.array_equal(r["accumulator"].view(np.uint32), g["accumulator"].view(np.uint32))


def test_golden_bunny_and_multi_model():
    g = np.load(os.path.join(GOLDEN, "bunny_128x128_primary.npz"))
    spec = scenes.bunny_scene()
    r = spec.build_oracle().render(spec.oracle_camera(), 128, 128, 1, max_depth=1, aov_depths=1, workers=6)
    assert np.array_equal(r["tri_id"][0, 0], g["tri_id"]) and np.array_equal(r["shadow"][0, 0], g["shadow"])
    assert np.array_equal(r["accumulator"].view(np.uint32), g["accumulator"].view(np.uint32))
    g = np.load(os.path.join(GOLDEN, "multi_model.npz"))
    spec = scenes.multi_model_scene()
    o = spec.build_oracle()
    boxes, topo, model_of = o.model_tree()
    assert np.array_equal(topo, g["topo"]) and np.array_equal(model_of, g["model_of"]) and np.array_equal(boxes, g["boxes"])
    r = o.render(spec.oracle_camera(), 96, 80, 2, max_depth=6, aov_depths=2, workers=6)
    assert np.array_equal(r["tri_id"], g["tri_id"]) and np.array_equal(r["accumulator"].view(np.uint32), g["accumulator"].view(np.uint32))


def test_config1_cpu_threadpool_render_is_deterministic():
    """BASELINE configs[0]: the reference's scene at 512x512, 1 spp, primary rays + direct shading, on the restated
    ThreadPool path (column tasks, oracle/column_pool.*).  The frame must not depend on the number of workers, and its
    counts are the ones a GPU run of the same configuration has to reproduce (tests/test_gpu_parity.py)."""
    spec = scenes.cube_scene()
    o = spec.build_oracle()
    a = o.render(spec.oracle_camera(), 512, 512, 1, max_depth=1, workers=6, aov_depths=1)
    b = o.render(spec.oracle_camera(), 512, 512, 1, max_depth=1, workers=1, aov_depths=1)
    assert np.array_equal(a["accumulator"].view(np.uint32), b["accumulator"].view(np.uint32))
    assert np.array_equal(a["tri_id"], b["tri_id"]) and np.array_equal(a["shadow"], b["shadow"])
    c = a["counters"]
    assert c["extendPrimary"] == 512 * 512 and c["shadow"] == c["hits"][0] and 0 < c["hits"][0] < 512 * 512
    assert c["nonFiniteAccepts"] == 0
    # direct light only: a lit pixel is albedo (0,1,0) * n.l + ambient 0.03, a shadowed or missed one at most 0.03 (BRDF.cpp:4-14)
    hit = a["tri_id"][0, 0] != 0xFFFFFFFF
    assert np.all(a["accumulator"][~hit] == 0) and np.all(a["accumulator"][hit][:, 0] <= np.float32(0.03))
