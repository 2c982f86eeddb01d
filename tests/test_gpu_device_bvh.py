"""The triangle-level BVH built ON THE DEVICE (rthip_upload_scene_build: Morton codes, radix sort, binary radix tree, refit,
collapse to quantised 4-wide nodes) against the same bar as the host's binned-SAH builder: the downloaded arrays are
structurally sound, the oracle's walk of them (variant B) equals its brute-force scan (variant A), and frames rendered
through the device-built tree equal the oracle bit for bit -- hit ids, shadow flags, accumulator.  Hit results must not
depend on which builder made the tree."""
import numpy as np
import pytest

from tests import scenes
from tests.bvh_checks import assert_bvh_sound

pytestmark = pytest.mark.gpu

SCENES = {"soup": lambda: scenes.soup_scene(20000, 3, 0.05), "hf": lambda: scenes.heightfield_scene(96, 9), "multi": scenes.multi_model_scene,
          "bunny": scenes.bunny_scene, "cube": scenes.cube_scene}


@pytest.mark.parametrize("name,W,H,N,depth", [("soup", 200, 160, 2, 6), ("hf", 192, 128, 2, 5), ("multi", 192, 160, 2, 6), ("bunny", 160, 160, 2, 4),
                                             ("cube", 128, 96, 2, 12)])
def test_frames_through_the_device_built_bvh_match_the_oracle(name, W, H, N, depth):
    spec = SCENES[name]()
    aov = min(depth, 3)
    r = spec.build_product(options=dict(maxDepth=depth, aovDepths=aov, deviceBvh=1))
    f = r.flat()
    host_s, device_ms = r.build_times()
    assert device_ms > 0
    if name != "cube":  # (12 triangles in leaves of 4: three leaves under one node)
        assert f["info"].numNodes > 1
    assert_bvh_sound(f)
    assert sorted(f["tris"]["triId"].tolist()) == list(range(f["info"].numTris))
    oracle = spec.build_oracle()
    want = oracle.render(spec.oracle_camera(), W, H, N, max_depth=depth, aov_depths=aov, workers=14)
    # the oracle's own walk of the device-built arrays finds what its brute-force scan finds
    oracle.set_flat(f["model_nodes"], f["nodes"], f["tris"], f["info_bytes"])
    walk = oracle.render(spec.oracle_camera(), W, H, N, max_depth=depth, aov_depths=aov, workers=14, flat=True)
    assert np.array_equal(walk["tri_id"], want["tri_id"]) and np.array_equal(walk["accumulator"].view(np.uint32), want["accumulator"].view(np.uint32))
    r.begin_frame(spec.product_camera(), W, H, N)
    for s in range(N):
        r.render_passes(s, 1)
        for d in range(aov):
            tri, sh = r.aov(s, d)
            assert np.array_equal(tri, want["tri_id"][s, d]), (s, d)
            assert np.array_equal(sh, want["shadow"][s, d]), (s, d)
    assert r.counters()["nonFiniteAccepts"] == 0
    assert np.array_equal(r.accumulator().view(np.uint32), want["accumulator"].view(np.uint32))
    r.close()


def test_device_build_of_the_10m_triangle_scene_renders_the_same_frame():
    """BASELINE configs[3]'s scene through both builders at 4K, 2 spp, depth 12: identical accumulators and ray counts; the
    device build takes a fraction of the host builder's time."""
    spec = scenes.hf10m_scene()
    W, H, N, depth = 3840, 2160, 2, 12
    frames, counts, times = [], [], []
    for device in (0, 1):
        r = spec.build_product(options=dict(maxDepth=depth, deviceBvh=device))
        host_s, device_ms = r.build_times()
        times.append((host_s, device_ms))
        r.begin_frame(spec.product_camera(), W, H, N)
        ms = r.render_passes(0, N)
        frames.append(r.accumulator())
        c = r.counters()
        counts.append((c["extendPrimary"], c["extendSecondary"], c["shadow"], c["hits"], c["nonFiniteAccepts"]))
        info = r.flat()["info"]
        print("deviceBvh=%d: host %.2f s, device build %.1f ms, %d nodes, depth %d, frame %.1f ms" % (device, host_s, device_ms, info.numNodes, info.maxDepthBlas, ms))
        r.close()
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    assert counts[0] == counts[1] and counts[0][4] == 0
    assert 0 < times[1][1] < 1000.0 * times[0][0]  # device milliseconds against host seconds
